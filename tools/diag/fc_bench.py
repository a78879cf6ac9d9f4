"""Time the fully connected products of the 41x49x35 network in isolation (MI355X): back-to-back launches of one product (warm caches)
and the same launches interleaved with a cache-evicting kernel (as inside the step):  python tools/diag/fc_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import vae_gam_amd
from vae_gam_amd import ops, _lib
dev = 'cuda'
F = _lib


def run(name, M, N, K, a_kc, b_kc, flags, n=50):
    A = torch.randn(M * K, device=dev); B = torch.randn(K * (N + 1), device=dev); C = torch.zeros(M * (N + 1), device=dev)
    bias = torch.randn(N + 1, device=dev); cx = torch.zeros(M, device=dev)
    a_str = (K, 1, 0) if a_kc else (1, M, 0); b_str = (1, K, 0) if b_kc else (N, 1, 0)
    big = torch.empty(64 << 20, device=dev)
    f = lambda: ops.fc_gemm(A, B, C, M, N, K, a_str, b_str, (N, 0), flags, bias=bias, amask=A, cx=cx)
    for mode in ('warm', 'evict'):
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            if mode == 'evict':
                big.add_(1.0)                                  # 256 MB through the caches
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); ts.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
        print('%-28s %-6s median %7.1f us   min %7.1f' % (name, mode, t[len(t) // 2], t[0]))


run('fc8 fwd 576x3840x200', 576, 3840, 200, True, True, F.FC_C_BIAS)
run('fc8 dX 576x200x3840', 576, 200, 3840, True, False, 0)
run('fc8 dW 3840x200x576', 3840, 200, 576, False, False, F.FC_B_ONES | F.FC_C_ACCUM)
run('fc7 dW 200x100x576', 200, 100, 576, False, False, F.FC_A_MASK | F.FC_B_ONES | F.FC_C_ACCUM)
run('fc7 fwd 576x200x100', 576, 200, 100, True, True, F.FC_C_BIAS | F.FC_C_RELU)
run('fc1 fwd 64x200x3072', 64, 200, 3072, True, True, F.FC_A_RELU | F.FC_C_BIAS | F.FC_C_RELU)
