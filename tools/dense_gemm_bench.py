"""psg_dense_gemm (own 256x256x64 MFMA GEMM with fused bias / GELU epilogue) against the library path
(F.linear [+ psg_bias_gelu]) on the Q-Former's projection shapes at BASELINE C2 (82 500 query rows, 35 000 text rows).
python tools/dense_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


shapes = [("w1q no GELU", 82500, 3072, 768, False), ("w1q + GELU", 82500, 3072, 768, True), ("w1t + GELU", 35000, 3072, 768, True),
          ("wq_x", 82500, 768, 768, False), ("qkv (layer 1)", 117500, 2304, 768, False),
          ("w2q (no LN)", 82500, 768, 3072, False)]
for name, M, N, K, gelu in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N)
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def lib():
        y = torch.nn.functional.linear(x, w)
        if gelu:
            ops.bias_gelu(y, b)
        else:
            ops.bias_gelu  # bias goes through the library epilogue in the product path
        return y
    t_lib_gemm = timeit(lambda: torch.nn.functional.linear(x, w, b.bfloat16() if not gelu else None))
    t_lib = timeit(lib) if gelu else t_lib_gemm
    t_own = timeit(lambda: ops.dense_gemm(x, w, b, gelu=gelu, out=out))
    ref = torch.nn.functional.linear(x[:4096].float(), w.float(), b)
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = (out[:4096].float() - ref).abs().max().item()
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M} N={N} K={K}: library GEMM {t_lib_gemm:7.1f} us ({fl / t_lib_gemm / 1e6:5.0f} TF/s)"
          f"{' + bias_gelu = %7.1f us' % t_lib if gelu else ''}; psg_dense_gemm {t_own:7.1f} us "
          f"({fl / t_own / 1e6:5.0f} TF/s); max err vs fp32 {err:.3e}")
