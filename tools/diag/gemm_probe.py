"""Which of the three GEMMs of the decoder's fc8 at the 82x98x70 geometry (rows = (C+1)*B = 208, K = 200, N = 66,560) is the slow one."""
import torch, time
torch.backends.cuda.preferred_blas_library('cublas')
M, K, N = 208, 200, 66560
x = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda'); gy = torch.randn(M, N, device='cuda')
wg = torch.zeros(N, K, device='cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for lib in ('cublas', 'cublaslt'):
    torch.backends.cuda.preferred_blas_library(lib)
    print(lib, 'fwd addmm(b, x, W^T)      %8.1f us' % t(lambda: torch.addmm(b, x, W.t())))
    print(lib, 'dx  gy @ W               %8.1f us' % t(lambda: gy @ W))
    print(lib, 'dW  wg.addmm_(gy^T, x)   %8.1f us' % t(lambda: wg.addmm_(gy.t(), x)))
    print(lib, 'dW  (x^T @ gy)^T form    %8.1f us' % t(lambda: wg.t().addmm_(x.t(), gy)))
    print(lib, 'db  addmv                %8.1f us' % t(lambda: b.addmv_(gy.t(), torch.ones(M, device="cuda"))))
