// TEST HARNESS ONLY -- see hip_emu.h
#include "hip_emu.h"
EmuBlock* g_emu_block = nullptr;
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
thread_local int emu_tid = 0;
std::mutex g_emu_atomic_mu;

void emu_run(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const int nt = block.x * block.y * block.z;
    EmuBlock blk;
    blk.bar.reset(new std::barrier<>(nt));
    const int nw = (nt + 63) / 64;
    blk.waves.resize(nw);
    for (int w = 0; w < nw; ++w) {
        blk.waves[w].lanes = std::min(64, nt - 64 * w);
        blk.waves[w].bar.reset(new std::barrier<>(blk.waves[w].lanes));
    }
    blk.dyn.assign(shmem + 64, 0);
    g_emu_block = &blk; blockDim = block; gridDim = grid;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        th.emplace_back([&, t]() {
            emu_tid = t;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        body();
                        blk.bar->arrive_and_wait();   // block boundary: statics (LDS) are reused
                    }
        });
    }
    for (auto& t : th) t.join();
    g_emu_block = nullptr;
}
