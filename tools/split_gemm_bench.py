"""The fp32s prompt-pass projections ([xh|xh|xl] . [wh|wl|wh]^T: fp16 operands, K' = 3K, fp32 result) at M rows:
the library's pick for the whole product against the same product cut along N, and psg_dense_gemm_tiled tiles.
    python tools/split_gemm_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 960
KMUL = int(os.environ.get("PSG_SPLIT_BENCH_KMUL", "3"))       # 3: the split products (fp32 result); 1: the 16-bit prompt pass (16-bit result)
OD = torch.float32 if KMUL == 3 else torch.float16
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]


def t_us(fn, n=12):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


tot = {}
for name, N, K in shapes:
    K3 = KMUL * K
    x = (torch.randn(M, K3, device=dev) * 0.5).half()
    ws = [(torch.randn(N, K3, device=dev) / K ** 0.5).half() for _ in range(2)]
    fl = 2.0 * M * N * K3
    res = {}
    res["lib"] = t_us(lambda i: torch.mm(x, ws[i & 1].t(), out_dtype=OD))
    for parts in (2, 3, 4):
        if N % (parts * 256):
            continue
        cuts = [w.view(parts, N // parts, K3) for w in ws]
        outs = torch.empty((parts, M, N // parts), device=dev, dtype=OD)

        def f(i, cuts=cuts, parts=parts, outs=outs):
            for p in range(parts):
                torch.mm(x, cuts[i & 1][p].t(), out_dtype=OD, out=outs[p])
        try:
            res[f"lib/{parts}N"] = t_us(f)
        except Exception as ex:                                       # noqa: BLE001
            res[f"lib/{parts}N"] = float("nan")
            print("  (", type(ex).__name__, str(ex)[:80], ")")
    if N % 512 == 0:                                                  # two column halves written in place (ldc = N)
        full = torch.empty((M, N), device=dev, dtype=OD)

        def h(i, full=full):
            for p in range(2):
                torch.mm(x, ws[i & 1][p * (N // 2):(p + 1) * (N // 2)].t(), out_dtype=OD,
                         out=full[:, p * (N // 2):(p + 1) * (N // 2)])
        res["lib/2N in place"] = t_us(h)
        ref = torch.mm(x, ws[0].t(), out_dtype=OD)
        h(0)
        print(f"  in-place halves vs whole: max |diff| {(full.float() - ref.float()).abs().max().item():.3e} (|ref| max {ref.abs().max().item():.1f})")
    if KMUL == 3:                                                     # the three K segments as a batched product + one sum
        xa = x.view(M, 3, K).permute(1, 0, 2)

        def k3(i):
            y3 = torch.bmm(xa, ws[i & 1].view(N, 3, K).permute(1, 2, 0), out_dtype=torch.float32)
            return y3.sum(0)
        try:
            res["bmm3+sum"] = t_us(k3)
            res["bmm3"] = t_us(lambda i: torch.bmm(xa, ws[i & 1].view(N, 3, K).permute(1, 2, 0), out_dtype=torch.float32))
            for P in (6, 12):
                if K3 % (P * 64) == 0:
                    xp = x.view(M, P, K3 // P).permute(1, 0, 2)
                    res[f"bmm{P}"] = t_us(lambda i, P=P, xp=xp: torch.bmm(xp, ws[i & 1].view(N, P, K3 // P).permute(1, 2, 0),
                                                                            out_dtype=torch.float32))
        except Exception as ex:                                       # noqa: BLE001
            print("  ( bmm3", type(ex).__name__, str(ex)[:80], ")")
    for parts in (2, 3, 4):                                           # cut along the rows instead
        if M % (parts * 16):
            continue
        outs = torch.empty((M, N), device=dev, dtype=OD)
        mr = M // parts

        def g(i, parts=parts, outs=outs, mr=mr):
            for p in range(parts):
                torch.mm(x[p * mr:(p + 1) * mr], ws[i & 1].t(), out_dtype=OD, out=outs[p * mr:(p + 1) * mr])
        res[f"lib/{parts}M"] = t_us(g)
    xs = [x[:, j * K:(j + 1) * K].contiguous() for j in range(3)]
    wss = [[w[:, j * K:(j + 1) * K].contiguous() for j in range(3)] for w in ws]
    for tile in ("256x256",):
        try:
            res[f"own {tile}"] = t_us(lambda i, tile=tile: ops.dense_gemm(x, ws[i & 1], None, out_dtype=OD, tile=tile))
        except Exception as ex:                                       # noqa: BLE001
            print("  (", tile, type(ex).__name__, str(ex)[:80], ")")
    line = "  ".join(f"{k} {v:7.1f} ({fl / v / 1e6:5.0f})" for k, v in res.items())
    print(f"{name:8s} M={M} N={N} K'={K3}: {line}   [us (TFLOP/s)]", flush=True)
    for k, v in res.items():
        tot.setdefault(k, 0.0)
        tot[k] += v
    tot.setdefault("best", 0.0)
    tot["best"] += min(v for v in res.values() if v == v)
    del ws, x
    torch.cuda.empty_cache()
print("per layer:", "  ".join(f"{k} {v:.0f}" for k, v in tot.items() if k in ("lib", "best")), "us")
