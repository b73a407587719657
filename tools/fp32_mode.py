"""The reference-precision (fp32) mode of the full path on the bench scene (BASELINE C3): ms per image.

    python tools/fp32_mode.py [steps] [warmup]            # prints one JSON line
    rocprofv3 --kernel-trace --stats -d /tmp/p -- python tools/fp32_mode.py 3 1

Same scene, head construction and step as bench.py's `parity_grade` leg (V4:99-100 loads the LLM without a dtype:
fp32 is the reference's own arithmetic).  PSG_F32_SKINNY=0 sends the decode projections back to the library SGEMM; PSG_MODE=fp32s runs the prompt pass's
projections as split-fp16 products (psg_split.hip).
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import bench
    from openpsg_amd.synthetic import make_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    a = bench.parse.__globals__["argparse"].Namespace(objects=50, size=1024, llm_layers=int(os.environ.get("PSG_LAYERS", "32")),
                                                       workload="full", dtype=os.environ.get("PSG_MODE", "fp32"), one_phase=False, pair_chunk=0,
                                                       categories=133)
    head = bench.setup_head(a, dev)
    if os.environ.get("PSG_F32_SKINNY") == "0":
        head.llm_engine.use_skinny = False
    scene = make_scene((a.size, a.size), a.objects, seed=0, device=str(dev), num_categories=a.categories)
    inputs = bench.scene_inputs(scene)
    el = bench.time_steps(lambda: head(inputs), warmup, steps) / steps
    print(json.dumps({"mode": a.dtype, "ms_per_step": round(el * 1e3, 3), "pairs_per_s": round(50 * 49 / el, 1),
                      "steps": steps, "skinny": bool(head.llm_engine.use_skinny)}), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
