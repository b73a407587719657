"""Does image k+1's front half (relation query + prompt pass: matrix-core bound) overlap image k's decode (HBM bound)?
Two heads with their own weights, two host threads, two HIP streams, the same scene: images per second against one head
alone.  python tools/overlap_probe.py [steps]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openpsg_amd.synthetic import make_scene  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    a = bench.argparse.Namespace(objects=50, size=1024, llm_layers=32, workload="full", dtype="mixed", one_phase=False,
                                 pair_chunk=0, categories=133)
    heads = [bench.setup_head(a, dev) for _ in range(2)]
    scene = make_scene((1024, 1024), 50, seed=0, device=str(dev))
    inputs = bench.scene_inputs(scene)
    for h in heads:
        for _ in range(3):
            h(inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        heads[0](inputs)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / steps

    def worker(h, stream, n, off):
        with torch.cuda.stream(stream):
            if off:
                time.sleep(off)
            for _ in range(n):
                h(inputs)
            stream.synchronize()
    for stagger in (0.0, one * 0.4):
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        for h, s in zip(heads, streams):                   # graphs were captured on the default stream; warm on these
            with torch.cuda.stream(s):
                h(inputs)
        torch.cuda.synchronize()
        th = [threading.Thread(target=worker, args=(heads[i], streams[i], steps, stagger * i)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / (2 * steps)
        print(f"one head: {one * 1e3:.2f} ms per image; two heads on two streams (stagger {stagger * 1e3:.0f} ms): "
              f"{two * 1e3:.2f} ms per image = {one / two:.3f}x", flush=True)


if __name__ == "__main__":
    main()
