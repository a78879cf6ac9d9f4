// calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read shapes the kernels use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports
// exactly half of the bytes of a wide coalesced streaming read; other access widths are uncalibrated: calibrate on a known byte count"):
//   variant 0: global_load_lds_dword  -- 4 bytes per lane, 256 B per wave-instruction, straight into LDS (the flat plane copies)
//   variant 1: global_load_dwordx4    -- 16 bytes per lane, 1 KiB per wave-instruction, into registers
//   variant 2: global_load_dword      -- 4 bytes per lane into registers (the batch-norm / GAM streaming kernels)
// Each variant reads a 1 GiB buffer exactly once (4x the Infinity Cache).   build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib     -> FETCH_SIZE (KB) per dispatch vs 1,048,576 KB read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) read_lds_dma4(const float* __restrict__ x, float* __restrict__ out, size_t nfloat) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_block = nfloat / gridDim.x;
    const float* src = x + (size_t)blockIdx.x * per_block;
    for (size_t o = (size_t)wave * 64; o < per_block; o += 256) {
        float* dst = lds + ((o / 64) % 32) * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + o + lane), (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 12345.f) out[0] = 1.f;
}
__global__ void __launch_bounds__(256) read_x4(const float4* __restrict__ x, float* __restrict__ out, size_t nvec) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
__global__ void __launch_bounds__(256) read_x1(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
    if (s == 12345.f) out[0] = s;
}

int main() {
    const size_t bytes = 1ull << 30, nfloat = bytes / 4;
    float *x, *out;
    CHECK(hipMalloc(&x, bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(x, 0, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_lds_dma4, dim3(2048), dim3(256), 32 * 64 * 4, 0, x, out, nfloat);
        hipLaunchKernelGGL(read_x4, dim3(2048), dim3(256), 0, 0, (const float4*)x, out, nfloat / 4);
        hipLaunchKernelGGL(read_x1, dim3(2048), dim3(256), 0, 0, x, out, nfloat);
    }
    CHECK(hipDeviceSynchronize());
    printf("read %zu bytes per dispatch, 3 dispatches per variant\n", bytes);
    return 0;
}
