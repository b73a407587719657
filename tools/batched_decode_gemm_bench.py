"""The decode-step projections at the row counts of `forward_batch` (2 / 4 / 8 images' selected pairs decoded together:
M = 40 / 80 / 160), fp16, COLD weights (> 800 MB rotated): the library's pick, the library on column parts, and the
psg_dense_gemm_tiled geometries, as us and TB/s of weight bytes.   python tools/batched_decode_gemm_bench.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
Ms = [int(a) for a in sys.argv[1:]] or [40, 80, 160]
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]


def bench(fn, n=16):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for M in Ms:
    tot, best_tot, nbytes = {}, 0.0, 0.0
    for name, N, K in shapes:
        nc = int(800e6 / (N * K * 2)) + 2
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(nc)]
        x = torch.randn(M, K, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        res = {"lib": bench(lambda i: torch.nn.functional.linear(x, ws[i % nc]))}
        for parts in ():
            if N % (parts * 256) == 0:
                h = N // parts

                def f(i, parts=parts, h=h):
                    w = ws[i % nc]
                    for p in range(parts):
                        torch.mm(x, w[p * h:(p + 1) * h].t(), out=out[:, p * h:(p + 1) * h])
                res[f"lib/{parts}N"] = bench(f)
        for tile in ():
            try:
                res["own " + tile] = bench(lambda i, tile=tile: ops.dense_gemm(x, ws[i % nc], out=out, tile=tile))
            except Exception as ex:                                   # noqa: BLE001
                pass
        if 32 < M <= 160:
            from openpsg_amd import _lib
            grids = [int(g_) for g_ in os.environ.get("PSG_BENCH_GRIDS", "0").split(",")]
            for grid in grids:
                for bn in (256, 128):
                    for mode in (1, 2):
                        _lib.set_option(0, "batch_gemm_bn", bn)
                        _lib.set_option(0, "batch_gemm_mode", mode)
                        _lib.set_option(0, "batch_gemm_grid", grid)
                        try:
                            pl = ops.batch_gemm(x, ws[0]).splits
                            res[f"g{grid}b{bn}{'al' if mode == 1 else 'sk'}/{pl}"] = bench(lambda i: ops.batch_gemm(x, ws[i % nc]))
                        except Exception:                             # noqa: BLE001
                            pass
            _lib.set_option(0, "batch_gemm_grid", 0)
            if os.environ.get("PSG_BENCH_ABLATE"):
                _lib.set_option(0, "batch_gemm_bn", 256)
                _lib.set_option(0, "batch_gemm_mode", 2)
                for var, nm in ((1, "nostore"), (2, "nomfma"), (3, "nox")):
                    _lib.set_option(0, "batch_gemm_var", var)
                    res[f"b256sk {nm}"] = bench(lambda i: ops.batch_gemm(x, ws[i % nc]))
                _lib.set_option(0, "batch_gemm_var", 0)
            _lib.set_option(0, "batch_gemm_bn", 0)
            _lib.set_option(0, "batch_gemm_mode", 0)
            res[f"batch/{ops.batch_gemm(x, ws[0]).splits}"] = bench(lambda i: ops.batch_gemm(x, ws[i % nc]))
        if M <= 32:
            res["skinny"] = bench(lambda i: ops.skinny_gemm(x, ws[i % nc]))
        gb = N * K * 2 / 1e3
        print(f"M={M:3d} {name:8s} {N:5d}x{K:5d}: " + "  ".join(f"{k} {v:6.1f} ({gb / v / 1e3:4.2f})" for k, v in res.items()),
              "  [us (TB/s)]", flush=True)
        mult = 1 if name == "lm_head" else 32
        for k, v in res.items():
            k = k.split('/')[0]
            tot[k] = tot.get(k, 0.0) + v * mult
        best_tot += min(res.values()) * mult
        nbytes += gb * 1e3 * mult
        del ws
        torch.cuda.empty_cache()
    print(f"M={M:3d} one decode step (32 layers + lm_head, {nbytes / 1e9:.1f} GB): library {tot['lib'] / 1e3:.2f} ms = "
          f"{nbytes / tot['lib'] / 1e6:.2f} TB/s; psg_batch_gemm {tot.get('batch', 0) / 1e3:.2f} ms; best variant per shape "
          f"{best_tot / 1e3:.2f} ms = {nbytes / best_tot / 1e6:.2f} TB/s")
