"""`head.forward_batch` (B images per step, their selected pairs decoded together; mixed mode, C3 scenes, 32 layers) with the
decode-step projections on the library GEMM (decode_batch_gemm = 0) against the measured plan that may use
psg_batch_gemm (1): ms per step, pairs/s, and how many token sequences agree.   python tools/batched_decode_ab.py [B ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd import _lib  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("batches", nargs="*", type=int, default=[2, 4, 8])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--only", type=int, default=-1, help="run only decode_batch_gemm = 0 or 1 (profiling)")
ns = ap.parse_args()
dev = torch.device("cuda:0")
a = argparse.Namespace(llm_layers=ns.layers, objects=50, dtype="mixed", workload="full", one_phase=False, slot_priorities="0,-1",
                       pair_chunk=0)
N = 50
res = {}
for flag in ((0, 1) if ns.only < 0 else (ns.only,)):
    _lib.set_option(0, "decode_batch_gemm", flag)
    head = bench.setup_head(a, dev)
    assert head.llm_engine.batch_gemm == bool(flag)
    for B in ns.batches:
        batch = [bench.scene_inputs(make_scene((1024, 1024), N, seed=m, device=str(dev))) for m in range(B)]
        outs = head.forward_batch(batch)
        el = bench.time_steps(lambda: head.forward_batch(batch), 1, ns.steps) / ns.steps
        toks = [repr(o.get("rel_pred")) for o in outs]
        res[(flag, B)] = (el, toks)
        print(f"decode_batch_gemm={flag} B={B}: {el * 1e3:8.2f} ms per step = {B * N * (N - 1) / el:9.0f} pairs/s", flush=True)
    del head
    torch.cuda.empty_cache()
_lib.set_option(0, "decode_batch_gemm", 1)
for B in (ns.batches if ns.only < 0 else []):
    t0, t1 = res[(0, B)][1], res[(1, B)][1]
    same = sum(int(x == y) for x, y in zip(t0, t1))
    print(f"B={B}: rel_pred of {same} of {len(t0)} images identical between the two; speed-up {res[(0, B)][0] / res[(1, B)][0]:.3f}x")
