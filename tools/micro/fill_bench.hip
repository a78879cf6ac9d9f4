// micro-benchmark: how fast can a block bring a haloed 3-D tile (rows of a pitch-33 tensor) into LDS?
// variants: 0 = LDS-DMA one row per wave-instruction (38 of 64 lanes), 1 = register loads + ds_write (8 rows in flight),
//           2 = LDS-DMA of the contiguous plane block (256 B per instruction), 3 = dwordx4-ish flat register copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ID = 39, IH = 47, IW = 33, CI = 8;
constexpr int LD = 4, LH = 52, LW = 38, LWP = 41;

__device__ __forceinline__ void dma4(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

template <int VAR>
__global__ void __launch_bounds__(256) fill_k(const float* __restrict__ x, float* __restrict__ out, int tilesD, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.y, tdi = blockIdx.x;
    const int id0 = tdi * 2 - 2;
    const size_t plane = IH * IW, vol = plane * ID;
    const float* xb = x + (size_t)n * CI * vol;
    float accv = 0.f;
    for (int c = 0; c < CI; ++c) {
        __syncthreads();
        if (VAR == 0) {
            for (int r = wave; r < LD * LH; r += 4) {
                const int dz = r / LH, hy = r % LH;
                const int id = min(max(id0 + dz, 0), ID - 1), ih = min(max(hy - 2, 0), IH - 1);
                const float* rowp = xb + c * vol + id * plane + ih * IW;
                if (lane < LW) dma4(rowp + min(max(lane - 2, 0), IW - 1), lds + r * LWP);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (VAR == 1) {
            for (int r0 = wave * 8; r0 < LD * LH; r0 += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = min(r0 + u, LD * LH - 1);
                    const int dz = r / LH, hy = r % LH;
                    const int id = min(max(id0 + dz, 0), ID - 1), ih = min(max(hy - 2, 0), IH - 1);
                    v[u] = xb[c * vol + id * plane + ih * IW + min(max(lane - 2, 0), IW - 1)];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (lane < LW && r0 + u < LD * LH) lds[(r0 + u) * LWP + lane] = v[u];
            }
        } else if (VAR == 2) {
            // the tile spans all of H and W: its LD planes are ONE contiguous block of the tensor
            const int idc = min(max(id0, 0), ID - LD);
            const float* src = xb + c * vol + idc * plane;
            const int nfl = LD * (int)plane;
            for (int o = wave * 64; o < nfl; o += 256) { if (o + lane < nfl) dma4(src + o + lane, lds + o); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int idc = min(max(id0, 0), ID - LD);
            const float* src = xb + c * vol + idc * plane;
            const int nfl = LD * (int)plane;
            float v[8];
            for (int o0 = threadIdx.x; o0 < nfl; o0 += 256 * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[min(o0 + u * 256, nfl - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (o0 + u * 256 < nfl) lds[o0 + u * 256] = v[u];
            }
        }
        __syncthreads();
        accv += lds[(threadIdx.x * 7) % lds_floats];
    }
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = accv;
}

int main() {
    const int N = 128, tilesD = 21;
    const size_t nel = (size_t)N * CI * ID * IH * IW;
    float *x, *out;
    CHECK(hipMalloc(&x, nel * 4)); CHECK(hipMalloc(&out, (size_t)N * tilesD * 256 * 4));
    CHECK(hipMemset(x, 0, nel * 4));
    const int lds_floats = LD * LH * LWP;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int extra = 0; extra <= 2; ++extra) {
        const size_t shmem = (size_t)lds_floats * 4 + 256 + extra * 20000;      // extra LDS => fewer blocks per CU
        for (int var = 0; var < 4; ++var) {
            auto launch = [&]() {
                dim3 g(tilesD, N), b(256);
                if (var == 0) hipLaunchKernelGGL(fill_k<0>, g, b, shmem, 0, x, out, tilesD, lds_floats);
                if (var == 1) hipLaunchKernelGGL(fill_k<1>, g, b, shmem, 0, x, out, tilesD, lds_floats);
                if (var == 2) hipLaunchKernelGGL(fill_k<2>, g, b, shmem, 0, x, out, tilesD, lds_floats);
                if (var == 3) hipLaunchKernelGGL(fill_k<3>, g, b, shmem, 0, x, out, tilesD, lds_floats);
            };
            launch(); CHECK(hipDeviceSynchronize());
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); CHECK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)N * tilesD * CI * LD * IH * IW * 4;
            printf("shmem %6zu B  variant %d: %8.1f us  (%.2f TB/s of tile bytes)\n", shmem, var, ms * 100, bytes / (ms * 1e-4) / 1e12);
        }
    }
    return 0;
}
