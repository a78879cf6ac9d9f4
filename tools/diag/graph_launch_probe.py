"""Does hipGraphLaunch block the host while the previous replay of the SAME executable graph is still running?  (MI355X, ROCm 7)
Prints host-side time of consecutive replay() calls for one graph replayed back to back and for two graphs alternating."""
import time, torch
dev = 'cuda'
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)


def make():
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            c = a @ b
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        c = a
        for _ in range(40):                      # a chain of small kernels + a few large ones
            c = c * 1.0001
        for _ in range(3):
            c = c @ b
    return g


g1, g2 = make(), make()
torch.cuda.synchronize()
for name, seq in (('same graph', [g1] * 6), ('alternating', [g1, g2] * 3)):
    torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for g in seq:
        t0 = time.perf_counter(); g.replay(); ts.append((time.perf_counter() - t0) * 1e3)
    t_issue = (time.perf_counter() - t00) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t00) * 1e3
    print('%-12s host ms per replay() call: %s   issued in %.2f ms, finished in %.2f ms' % (name, ' '.join('%.2f' % t for t in ts), t_issue, t_all))
