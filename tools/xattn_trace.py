"""Per-wave timeline of one cross-attention launch (debugging aid; psg_set_trace_buffer makes the kernel write
32 cycle-counter stamps per wave into a caller-provided buffer).  python tools/xattn_trace.py [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops, _lib  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
P, L = N * N, 256
q = torch.randn(P * 33, 768, device=dev).bfloat16()
k = torch.randn(L, 768, device=dev).bfloat16()
v = torch.randn(L, 768, device=dev).bfloat16()
sc = make_scene((1024, 1024), N, seed=0, device="cuda:0", features=False)
grid = ops.mask_grid(sc["pan_results"], (1024, 1024), (1024, 1024), (16, 16))
bits = ops.object_bitmasks(grid, torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32, device=dev))
pidx = torch.arange(P, device=dev, dtype=torch.int32)
out = torch.empty_like(q)
for _ in range(3):
    ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out, variant=_lib.PSG_XATTN_MFMA)
torch.cuda.synchronize()
buf = torch.zeros(1 << 21, dtype=torch.int64, device=dev)           # caller-provided stamp buffer
_lib.set_trace_buffer(0, _lib.PSG_TRACE_CROSS_ATTN, buf)
ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out, variant=_lib.PSG_XATTN_MFMA)
torch.cuda.synchronize()
_lib.set_trace_buffer(0, _lib.PSG_TRACE_NONE)
t = buf.cpu().numpy().reshape(-1, 32)
t = t[t[:, 0] > 0]
busy = t[t[:, 31] > 0]
t0 = t[:, 0].min()
nun = busy[:, 31]
last = np.array([busy[i, 2 + min(int(nun[i]), 28)] for i in range(len(busy))])
life = last - busy[:, 0]
s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_.record()
for _ in range(10):
    ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out, variant=_lib.PSG_XATTN_MFMA)
e_.record()
torch.cuda.synchronize()
kus = s_.elapsed_time(e_) * 100.0
MHZ = life.max() / kus                        # the longest-lived wave spans (almost) the whole kernel
print(f"kernel ~{kus:.1f} us; longest wave {life.max()} ticks -> {MHZ:.1f} ticks/us (counters of different XCDs have different bases)")
stage = (busy[:, 1] - busy[:, 0]) / MHZ
pre = (busy[:, 2] - busy[:, 1]) / MHZ
print(f"waves with work {len(busy)} of {len(t)}; staging us: mean {stage.mean():.1f} p90 {np.percentile(stage, 90):.1f} max {stage.max():.1f}; "
      f"queue+first fetch us: mean {pre.mean():.1f} max {pre.max():.1f}; wave lifetime us: mean {(life / MHZ).mean():.1f} "
      f"p10 {np.percentile(life / MHZ, 10):.1f} p90 {np.percentile(life / MHZ, 90):.1f}")
print("units per wave: min %d mean %.1f max %d" % (nun.min(), nun.mean(), nun.max()))
d = []
for i in range(len(busy)):
    n = min(int(nun[i]), 28)
    d += list(np.diff(busy[i, 2:3 + n]) / MHZ)
d = np.array(d)
print("unit duration us: mean %.2f  p10 %.2f p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
for w in (0, 777 % len(busy), len(busy) - 1):
    print(f"units of busy wave {w}:", np.round(np.diff(busy[w, 2:3 + min(int(nun[w]), 28)]) / MHZ, 2))
