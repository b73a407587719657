"""Prompt-pass projection shapes (M rows x N x K, fp16): library GEMM vs psg_dense_gemm, also at K' = 3K (the
split-fp16 products of the fp32s mode).  python tools/gemm_shapes_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 980
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
variants = os.environ.get("PSG_GEMM_VARIANTS", "lib,own").split(",")


def bench(fn, n=24):
    for _ in range(4):
        fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for kmul in (1, 3):
    tot = {v: 0.0 for v in variants}
    for name, N, K0 in shapes:
        K = K0 * kmul
        x = (torch.randn(M, K, device=dev)).half()
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(3)]
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        line = f"K'={kmul}K {name:8s} M={M} N={N} K={K}:"
        for v in variants:
            if v == "lib":
                us = bench(lambda i: torch.nn.functional.linear(x, ws[i % 3]))
            elif v == "lib32":
                us = bench(lambda i: torch.mm(x, ws[i % 3].t(), out_dtype=torch.float32))
            elif v == "rb32":                                        # the same through rocBLAS
                torch.backends.cuda.preferred_blas_library("cublas")
                try:
                    us = bench(lambda i: torch.mm(x, ws[i % 3].t(), out_dtype=torch.float32))
                finally:
                    torch.backends.cuda.preferred_blas_library("cublaslt")
            elif v == "rb":
                torch.backends.cuda.preferred_blas_library("cublas")
                try:
                    us = bench(lambda i: torch.nn.functional.linear(x, ws[i % 3]))
                finally:
                    torch.backends.cuda.preferred_blas_library("cublaslt")
            elif v == "own":
                us = bench(lambda i: ops.dense_gemm(x, ws[i % 3], out=out))
            elif v.startswith("own:"):                               # own:<tile>, e.g. own:256x192, own:auto
                us = bench(lambda i: ops.dense_gemm(x, ws[i % 3], out=out, tile=v[4:]))
            elif v.startswith("swiglu:"):                            # gate_up with the SwiGLU epilogue (no interleave needed to time it)
                if name != "gate_up":
                    continue
                o2 = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
                us = bench(lambda i: ops.dense_gemm(x, ws[i % 3], out=o2, tile=v[7:], swiglu=True))
            elif v == "lib+silu":
                if name != "gate_up":
                    continue
                o2 = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
                us = bench(lambda i: ops.silu_mul(torch.nn.functional.linear(x, ws[i % 3]), o2))
            elif v == "sk":
                us = bench(lambda i: ops.streamk_gemm(x, ws[i % 3]))
            elif v == "sk32":
                us = bench(lambda i: ops.streamk_gemm(x, ws[i % 3], out_dtype=torch.float32))
            else:
                continue
            tot[v] += us
            line += f"  {v} {us:7.1f} us = {2 * M * N * K / us / 1e6:5.0f} TF"
        print(line, flush=True)
    print(f"K'={kmul}K per layer: " + ", ".join(f"{v} {t:.0f} us -> 32 layers {t * 32 / 1e3:.2f} ms" for v, t in tot.items()))
