"""Can the prompt pass's matrix-core GEMMs (MFMA / LDS bound) run beside another image's decode projections (HBM
bound) on the same CUs?  Stream A: the four weight-streaming launches of a decode layer (M = 20, cold weights, LAYERS
distinct layers), stream B: the four prompt-pass GEMMs at M = 980 on the repo's own kernel with a given tile or on the
library.  Prints each stream alone and both together; sum / together = 1 means serial, 2 means perfect overlap.

    PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8 python tools/corun_probe.py 256x128
    python tools/corun_probe.py lib
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
tile = sys.argv[1] if len(sys.argv) > 1 else "256x128"
LAYERS = int(os.environ.get("LAYERS", "6"))
REPS_A = int(os.environ.get("REPS_A", "40"))
shapes = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]
wl = [[(torch.randn(n, k, device=dev) / k ** 0.5).half() for n, k in shapes] for _ in range(LAYERS)]
xa = [torch.randn(20, k, device=dev).half() for _, k in shapes]
xb = [torch.randn(980, k, device=dev).half() for _, k in shapes]
outb = [torch.empty(980, n, device=dev, dtype=torch.float16) for n, _ in shapes]


def run_a():
    for _ in range(REPS_A):
        for L in wl:
            for x, w in zip(xa, L):
                ops.skinny_gemm(x, w)


def run_b(reps):
    for _ in range(reps):
        for L in wl:
            for x, w, o in zip(xb, L, outb):
                if tile == "lib":
                    torch.mm(x, w.t(), out=o)
                else:
                    ops.dense_gemm(x, w, out=o, tile=tile)


def timed(fn_list):
    torch.cuda.synchronize()
    evs = []
    s0 = torch.cuda.Event(enable_timing=True)
    s0.record()
    for st, fn in fn_list:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
    torch.cuda.synchronize()
    return [s0.elapsed_time(e) for e in evs]


sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with torch.cuda.stream(sa):
    run_a()
with torch.cuda.stream(sb):
    run_b(1)
torch.cuda.synchronize()
ta = timed([(sa, run_a)])[0]
tb1 = timed([(sb, lambda: run_b(4))])[0] / 4
reps_b = max(1, int(round(ta / tb1)))                      # about as much B work as A work
tb = timed([(sb, lambda: run_b(reps_b))])[0]
both = timed([(sa, run_a), (sb, lambda: run_b(reps_b))])
print(f"tile {tile} skinny_dma={os.environ.get('PSG_SKINNY_DMA', 'default')}: A (decode GEMMs) alone {ta:.2f} ms "
      f"({ta * 1e3 / REPS_A / LAYERS:.1f} us/layer), B (prompt GEMMs x{reps_b}) alone {tb:.2f} ms "
      f"({tb * 1e3 / reps_b / LAYERS:.1f} us/layer); together A ends {both[0]:.2f}, B ends {both[1]:.2f} ms -> "
      f"(A + B) / max = {(ta + tb) / max(both):.3f}", flush=True)
