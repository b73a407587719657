"""Within-process interleaved A/B of a psg_ctx option on the relation-query workload (BASELINE C2):
python tools/ab_option.py qformer_own_gemm 0 1 [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd import _lib  # noqa: E402
from openpsg_amd.categories import INSTANCE_OFFSET, object_categories  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

name, vals = sys.argv[1], [int(v) for v in sys.argv[2:4]]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
sys.argv = [sys.argv[0], "--workload", "rq"]
a = bench.parse()
head = bench.setup_head(a, dev)
scene = make_scene((1024, 1024), 50, seed=0, device="cuda:0")
ids = [int(i) for i in scene["object_id_list"]]
names = [object_categories[i % INSTANCE_OFFSET] for i in ids]


def step():
    return head.run_relation_query(scene["mask_features"], scene["img_meta"], ids, names, scene["pan_results"])["selected"].cpu()


res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        _lib.set_option(0, name, v)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 20 * 1e3)
for v in vals:
    xs = sorted(res[v])
    print(f"{name}={v}: median {xs[len(xs) // 2]:.3f} ms, min {xs[0]:.3f}, all {[round(x, 3) for x in res[v]]}")
