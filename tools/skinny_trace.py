"""Per-wave timeline of one weight-streaming GEMM launch (debugging aid; psg_set_trace_buffer makes the kernel
write 8 cycle-counter stamps per wave into a caller-provided buffer).  python tools/skinny_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 20
for name, N, K in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(max(2, int(600e6 / (N * K * 2)) + 1))]
    for i in range(len(ws) + 2):
        ops.skinny_gemm(x, ws[i % len(ws)])
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for i in range(16):
        ops.skinny_gemm(x, ws[i % len(ws)])
    e_.record()
    torch.cuda.synchronize()
    kus = s_.elapsed_time(e_) * 1e3 / 16
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)       # caller-provided stamp buffer
    _lib.set_trace_buffer(0, _lib.PSG_TRACE_SKINNY_GEMM, buf)
    ops.skinny_gemm(x, ws[1])
    torch.cuda.synchronize()
    _lib.set_trace_buffer(0, _lib.PSG_TRACE_NONE)
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    life = t[:, 5] - t[:, 0]
    # calibrate ticks/us on the constant 100 MHz assumption check: longest wave ~ kernel time minus launch ramp
    mhz = 100.0
    d = lambda a, b: (t[:, a] - t[:, b]) / mhz
    print(f"{name:8s} N={N} K={K}: launch-to-launch {kus:.1f} us; waves {len(t)}; slabs/wave {t[:, 6].mean():.2f}; "
          f"per wave (us @100MHz ticks): issue {d(1, 0).mean():.2f}  x-staged {d(2, 0).mean():.2f}  first-batch {d(3, 0).mean():.2f}  "
          f"stream-end {d(4, 0).mean():.2f}  done {d(5, 0).mean():.2f} (max {d(5, 0).max():.2f}, min {d(5, 0).min():.2f})")
