"""hipBLASLt vs rocBLAS on the four prompt-pass projection shapes (M = 920 rows)."""
import sys
import torch

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 920
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    tot = 0.0
    for name, N, K in shapes:
        x = torch.randn(M, K, device=dev).bfloat16()
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(4)]
        for i in range(8):
            torch.nn.functional.linear(x, ws[i % 4])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(32):
            torch.nn.functional.linear(x, ws[i % 4])
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 32
        tot += us
        print(f"{lib:9s} {name:8s} M={M} N={N} K={K}: {us:7.1f} us = {2 * M * N * K / us / 1e6:6.0f} TFLOP/s")
    print(f"{lib:9s} per layer {tot:.1f} us -> 32 layers {tot * 32 / 1e3:.2f} ms")
