// Does global_load_lds_dwordx4 accept a global source that is only 4-byte aligned (LDS destination 16-byte aligned)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/dma16_align tools/micro/dma16_align.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* src, float* dst, int shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 4 * lane),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 1024);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(d, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, s, d, shift);
        std::vector<float> o(256);
        hipError_t e = hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) if (o[i] != (float)(i + shift)) ++bad;
        printf("shift %d floats: err=%d mismatches=%d (o[0..4]= %g %g %g %g %g)\n", shift, (int)e, bad, o[0], o[1], o[2], o[3], o[4]);
    }
    return 0;
}
