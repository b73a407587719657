"""Stress of `head.submit` (two images in flight) on the benchmark model, with a watchdog: a hang dumps the host
traceback and exits instead of blocking the box.  python tools/inflight_stress.py [iterations]

Round 4 also tried two BATCHES in flight (forward_batch on two streams: 40 / 80 decode rows, so the decode projections
are library GEMMs) with this tool: the GPU hung at the first overlap - a library GEMM inside one stream's graph beside an
eager one on another stream -, while one image per batch (20 rows, own kernels) and PSG_NO_SKINNY=1 at 20 rows ran.  That
path was removed; `head.submit` orders the library-GEMM phases of consecutive images by an event."""
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

mode = "single"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
faulthandler.dump_traceback_later(int(os.environ.get("PSG_WATCHDOG_S", "240")), exit=True)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
# PSG_STRESS_DTYPE=fp32s (the headline mode: library GEMMs of the prompt pass inside both slots' graphs), PSG_STRESS_VALUES=fp16
a = bench.argparse.Namespace(objects=50, size=1024, llm_layers=32, workload="full", dtype=os.environ.get("PSG_STRESS_DTYPE", "mixed"),
                             one_phase=False, pair_chunk=0, categories=133, llm_values=os.environ.get("PSG_STRESS_VALUES", "fp32"))
head = bench.setup_head(a, dev)
if os.environ.get("PSG_NO_SKINNY") == "1":                      # decode projections on the library GEMM (inside the graphs)
    head.llm_engine.use_skinny = False
scenes = [bench.scene_inputs(make_scene((1024, 1024), 50, seed=m, device=str(dev))) for m in range(4)]
t0 = time.perf_counter()
pend = []
for j in range(iters):
    pend.append(head.submit(scenes[j % 4], slot=j % 2))
    if len(pend) > 1:
        pend.pop(0).result()
    if j % 10 == 0:
        print(mode, "iteration", j, f"{time.perf_counter() - t0:.1f}s", flush=True)
while pend:
    pend.pop(0).result()
torch.cuda.synchronize()
print(mode, "done:", iters, "iterations in", f"{time.perf_counter() - t0:.1f}s", flush=True)
