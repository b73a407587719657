"""Prompt-pass projection GEMMs (hipBLASLt, M = 960) with COLD weights (a ring of copies larger than the Infinity
Cache), WARM weights (one copy, resident), and cold weights preceded by a streaming touch of the same weights
(what a prefetch during the previous kernel would leave behind).  python tools/prefill_gemm_warm.py [M]"""
import sys
import torch

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 960
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]


def timeit(fn, n=24):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


tot = [0.0, 0.0, 0.0]
for name, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    ncopy = max(2, int(700e6 / (N * K * 2)) + 1)
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(ncopy)]
    cold = timeit(lambda i: torch.nn.functional.linear(x, ws[i % ncopy]))
    warm = timeit(lambda i: torch.nn.functional.linear(x, ws[0]))
    touch = timeit(lambda i: ws[i % ncopy].view(torch.int32).view(-1)[::1].sum())          # streaming read only
    both = timeit(lambda i: (ws[i % ncopy].view(torch.int32).view(-1).sum(), torch.nn.functional.linear(x, ws[i % ncopy])))
    tot[0] += cold
    tot[1] += warm
    tot[2] += both - touch
    print(f"{name:8s} M={M} N={N} K={K} ({N * K * 2 / 1e6:.0f} MB): cold {cold:6.1f} us, warm {warm:6.1f} us, "
          f"touch alone {touch:6.1f} us, touch + gemm {both:6.1f} us -> gemm after touch {both - touch:6.1f} us")
print(f"per layer: cold {tot[0]:.1f} us, warm {tot[1]:.1f} us, after a touch {tot[2]:.1f} us; x 32 layers: "
      f"{tot[0] * 32 / 1e3:.2f} / {tot[1] * 32 / 1e3:.2f} / {tot[2] * 32 / 1e3:.2f} ms")
