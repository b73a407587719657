"""Host-side cost of NEW images: the benchmark repeats one synthetic scene, so the per-names caches of the head
(prompt tables, per-chunk prompt gathers, decode graphs) are hot.  This runs S different scenes (different class
lists) once each - every cache misses - and then again (hits): python tools/cold_scenes.py [S] [--workload rq]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd.categories import INSTANCE_OFFSET, object_categories  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
rq_only = "rq" in sys.argv
prewarm = "--warm-prompts" in sys.argv
dev = torch.device("cuda:0")
sys.argv = [sys.argv[0]] + (["--workload", "rq"] if rq_only else [])
a = bench.parse()
head = bench.setup_head(a, dev)
if prewarm:
    t0 = time.perf_counter()
    head.warm_prompts()
    print(f"warm_prompts(): {time.perf_counter() - t0:.2f} s (all 133 x 133 class pairs, both tokenizers)", flush=True)
scenes = [make_scene((a.size, a.size), a.objects, seed=100 + m, device="cuda:0") for m in range(S)]


def run(sc):
    if rq_only:
        ids = [int(i) for i in sc["object_id_list"]]
        names = [object_categories[i % INSTANCE_OFFSET] for i in ids]
        rq = head.run_relation_query(sc["mask_features"], sc["img_meta"], ids, names, sc["pan_results"])
        head.selected_pair_features(rq)
        return rq["selected"].cpu()
    return head(bench.scene_inputs(sc))


run(make_scene((a.size, a.size), a.objects, seed=99, device="cuda:0"))      # library / graph warm-up on another scene
torch.cuda.synchronize()
for label in ("cold (every scene new)", "warm (same scenes again)"):
    ts = []
    for sc in scenes:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(sc)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{label}: mean {sum(ts) / len(ts):.2f} ms per image, each {[round(t, 1) for t in ts]}", flush=True)
