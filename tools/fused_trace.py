"""Per-wave timeline of one fused decode projection (psg_skinny_gemm_fused): where the in-launch hand-off spends
its time.  python tools/fused_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 20
g = torch.Generator().manual_seed(0)
for name, N, K in [("qkv", 12288, 4096), ("gate_up", 22016, 4096), ("lm_head", 32000, 4096)]:
    resid = torch.randn(M, K, generator=g).to(dev).bfloat16()
    ln = torch.ones(K, device=dev)
    delta = ops.Partials((torch.randn(8, M, K, generator=g) * 0.1).to(dev))
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(max(2, int(600e6 / (N * K * 2)) + 1))]
    n = torch.empty_like(resid)

    def sep(i):
        ops.rmsnorm(resid, delta, ln, 1e-5, n)
        return ops.skinny_gemm(n, ws[i % len(ws)])

    def fused(i):
        sync = syncs[i % 64]
        return ops.skinny_gemm_fused(ops.PSG_PRO_RMSNORM, n, ws[i % len(ws)], sync, inp=delta, resid=resid, norm_w=ln, eps=1e-5)

    for label, fn in (("separate", sep), ("fused", fused)):
        syncs = torch.zeros(64, 2, device=dev, dtype=torch.int32)
        for i in range(len(ws) + 2):
            fn(i)
        torch.cuda.synchronize()
        syncs.zero_()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for i in range(32):
            fn(i)
        e_.record()
        torch.cuda.synchronize()
        print(f"{name:8s} {label:9s}: {s_.elapsed_time(e_) * 1e3 / 32:.1f} us per (row op + projection)")
    syncs = torch.zeros(64, 2, device=dev, dtype=torch.int32)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    _lib.set_trace_buffer(0, _lib.PSG_TRACE_SKINNY_GEMM, buf)
    fused(1)
    torch.cuda.synchronize()
    _lib.set_trace_buffer(0, _lib.PSG_TRACE_NONE)
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    d = lambda a, b: (t[:, a] - t[:, b]) / 100.0   # noqa: E731
    t0 = t[:, 0].min()
    print(f"   waves {len(t)}; start spread {((t[:, 0] - t0) / 100.0).max():.2f} us; per wave from its own start: ring issued {d(1, 0).mean():.2f}  "
          f"row op done {d(6, 0).mean():.2f} (max {d(6, 0).max():.2f})  counter complete {d(7, 0).mean():.2f} (min {d(7, 0).min():.2f} max {d(7, 0).max():.2f})  "
          f"x staged {d(2, 0).mean():.2f}  first batch {d(3, 0).mean():.2f}  done {d(5, 0).mean():.2f} (max {d(5, 0).max():.2f})")
