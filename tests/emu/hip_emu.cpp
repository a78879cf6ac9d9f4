// TEST HARNESS ONLY -- see hip_emu.h
#include "hip_emu.h"
#include <thread>
#include <ucontext.h>

thread_local EmuBlock* g_emu_block = nullptr;
thread_local dim3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local int emu_tid = 0;
std::mutex g_emu_atomic_mu;

namespace {
constexpr size_t STACK = 192 * 1024;
struct Fiber { ucontext_t ctx; bool done; };
struct Runner {
    ucontext_t sched;
    std::vector<Fiber> fib;
    std::vector<char> stacks;
    const std::function<void()>* body = nullptr;
    int cur = 0;
};
thread_local Runner* g_runner = nullptr;

void trampoline() {
    Runner* r = g_runner;
    (*r->body)();
    r->fib[r->cur].done = true;
    swapcontext(&r->fib[r->cur].ctx, &r->sched);
}
}  // namespace

void emu_yield() {
    Runner* r = g_runner;
    swapcontext(&r->fib[r->cur].ctx, &r->sched);
}

static void run_block(Runner& R, dim3 grid, dim3 block, size_t shmem, unsigned bid) {
    const int nt = block.x * block.y * block.z;
    EmuBlock blk;
    blk.bar.count = nt;
    const int nw = (nt + 63) / 64;
    blk.waves.resize(nw);
    for (int w = 0; w < nw; ++w) { blk.waves[w].lanes = std::min(64, nt - 64 * w); blk.waves[w].bar.count = blk.waves[w].lanes; }
    blk.dyn.assign(shmem + 64, 0);
    g_emu_block = &blk; blockDim = block; gridDim = grid;
    blockIdx = dim3(bid % grid.x, (bid / grid.x) % grid.y, bid / (grid.x * grid.y));
    if ((int)R.fib.size() < nt) { R.fib.resize(nt); R.stacks.resize((size_t)nt * STACK); }
    for (int t = 0; t < nt; ++t) {
        getcontext(&R.fib[t].ctx);
        R.fib[t].ctx.uc_stack.ss_sp = R.stacks.data() + (size_t)t * STACK;
        R.fib[t].ctx.uc_stack.ss_size = STACK;
        R.fib[t].ctx.uc_link = &R.sched;
        R.fib[t].done = false;
        makecontext(&R.fib[t].ctx, trampoline, 0);
    }
    int ndone = 0;
    while (ndone < nt) {
        int progressed = 0;
        for (int t = 0; t < nt; ++t) {
            if (R.fib[t].done) continue;
            R.cur = t; emu_tid = t;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&R.sched, &R.fib[t].ctx);
            if (R.fib[t].done) ++ndone;
            ++progressed;
        }
        if (!progressed) break;
    }
    g_emu_block = nullptr;
}

void emu_run(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const unsigned nblocks = grid.x * grid.y * grid.z;
    std::atomic<unsigned> next(0);
    const unsigned nthreads = std::min<unsigned>(nblocks, std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency())));
    auto work = [&]() {
        Runner R; R.body = &body; g_runner = &R;
        for (;;) { const unsigned b = next.fetch_add(1); if (b >= nblocks) break; run_block(R, grid, block, shmem, b); }
        g_runner = nullptr;
    };
    if (nthreads <= 1) { work(); return; }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nthreads; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}
