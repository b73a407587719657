"""psg_dense_gemm_split (three products from one staging of interleaved hi / lo operands) against the K' = 3K form of
psg_dense_gemm_ex, at the fp32s Q-Former's shapes (82 500 rows = 2500 pairs x 33) and the Llama prompt pass's (960 rows):
microseconds of the activation split and of the GEMM, TFLOP/s of algorithmic work (2 M N K).   python tools/dense_split_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    shapes = [(82500, 768, 768, 0), (82500, 2304, 768, 0), (82500, 3072, 768, 1), (82500, 768, 3072, 0), (25168, 3072, 768, 1),
              (2500, 768, 768, 0), (960, 12288, 4096, 0), (960, 4096, 4096, 0), (960, 22016, 4096, 0), (960, 4096, 11008, 0)]
    for M, N, K, gelu in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        tile = "auto" if M < 16384 else "256x256"
        w3, c3 = ops.split_f16x3(w, weights=True)
        w2, c2 = ops.split_f16i2(w)
        t_s3 = timeit(lambda: ops.split_f16x3(x))
        t_s2 = timeit(lambda: ops.split_f16i2(x))
        a3, r3 = ops.split_f16x3(x)
        a2, r2 = ops.split_f16i2(x)
        t_g3 = timeit(lambda: ops.dense_gemm(a3, w3, b, gelu=bool(gelu), out_dtype=torch.float32, row_scale=r3, col_scale=c3, tile=tile))
        t_g2 = timeit(lambda: ops.dense_gemm_split(a2, w2, b, r2, c2, gelu=bool(gelu), tile=tile))
        y3 = ops.dense_gemm(a3, w3, b, gelu=bool(gelu), out_dtype=torch.float32, row_scale=r3, col_scale=c3, tile=tile)
        y2 = ops.dense_gemm_split(a2, w2, b, r2, c2, gelu=bool(gelu), tile=tile)
        d = (y3 - y2).abs().max().item()
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:5d} K={K:5d} gelu={gelu}: split {t_s3:7.1f} -> {t_s2:7.1f} us; GEMM {t_g3:8.1f} -> {t_g2:8.1f} us "
              f"({fl / t_g3 / 1e6:6.1f} -> {fl / t_g2 / 1e6:6.1f} TFLOP/s algorithmic, x3 executed); max |3K form - split form| = {d:.2e}")


if __name__ == "__main__":
    main()
