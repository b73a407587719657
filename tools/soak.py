"""Soak run (development aid): many images with varying object counts, sizes and class lists through one head; prints
time per image and the allocator's high-water marks so that growth of the per-names / per-shape caches (prompt stores,
index tables, captured decode graphs) shows up.   python tools/soak.py [images]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402

n_img = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
sys.argv = sys.argv[:1] + ["--no-cpu-baseline", "--no-parity"] + [x for x in sys.argv[2:] if x.startswith("--") or x in ("fp16", "fp32")]
a = bench.parse()
dev = torch.device("cuda:0")
head = bench.setup_head(a, dev)
head.suppress_eos = False                                   # natural EOS: the chunked graphs and the early exit
head.warm_prompts()
g = torch.Generator().manual_seed(0)
sizes = [(1024, 1024), (768, 1024), (1024, 1344), (512, 512), (640, 960)]
marks = []
t_all = time.time()
for i in range(n_img):
    n = int(torch.randint(2, 51, (1,), generator=g))
    hw = sizes[int(torch.randint(0, len(sizes), (1,), generator=g))]
    sc = make_scene(hw, n, seed=1000 + i, device="cuda:0", num_categories=int(torch.randint(5, 134, (1,), generator=g)))
    t0 = time.time()
    out = head(bench.scene_inputs(sc))
    torch.cuda.synchronize()
    dt = time.time() - t0
    if i % 10 == 9 or i == n_img - 1:
        marks.append((i + 1, torch.cuda.memory_allocated() / 2**30, torch.cuda.max_memory_allocated() / 2**30,
                      torch.cuda.memory_reserved() / 2**30, len(head.llm_engine._graphs), dt * 1e3))
        print(f"image {i + 1:4d}: N={n:2d} {hw}  {dt * 1e3:7.1f} ms; allocated {marks[-1][1]:.2f} GiB (peak {marks[-1][2]:.2f}, "
              f"reserved {marks[-1][3]:.2f}); decode graph shapes cached {marks[-1][4]}", flush=True)
print(f"{n_img} images in {time.time() - t_all:.1f} s")
grow = marks[-1][1] - marks[len(marks) // 2][1]
print(f"allocated memory growth over the second half of the run: {grow:+.3f} GiB")
