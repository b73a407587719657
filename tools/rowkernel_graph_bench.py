"""What a decode step's row kernels cost as INDEPENDENT launches inside a HIP graph (replays of 64 launches each on cold
slices).  Caveat learnt in round 6 (profiles/r06_norm_commute_ab.txt): behind a weight-streaming launch a DEPENDENT kernel
lasts ~5 us whatever these figures say - its first loads wait for the producer's slices to become visible across the eight
L2s - so a kernel that is 3x cheaper here can buy nothing in the chain.   python tools/rowkernel_graph_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402


def graph_time(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    K, D, I, n = 20, 4096, 11008, 64
    w = torch.ones(D, device=dev)
    resid = torch.randn(K, D, device=dev)
    out = torch.empty(K, D, device=dev)
    act = torch.empty(K, I, device=dev)
    for S in (8, 16):
        # one partial buffer per launch: as cold in L2 as the GEMM's slices are for the kernel behind it
        parts = [ops.Partials(torch.randn(S, K, D, device=dev)) for _ in range(n)]

        def norm():
            for i in range(n):
                ops.rmsnorm(resid, parts[i], w, 1e-5, out)
        print(f"rmsnorm, {S:2d} slices of [20, 4096]: {graph_time(norm) / n:5.2f} us per launch in a graph")
    gus = [ops.Partials(torch.randn(8, K, 2 * I, device=dev)) for _ in range(n)]

    def silu():
        for i in range(n):
            ops.silu_mul(gus[i], act)
    print(f"silu_mul, 8 slices of [20, 22016]: {graph_time(silu) / n:5.2f} us per launch in a graph")
    x = torch.randn(K, D, device=dev)

    def split():
        for i in range(n):
            ops.split_f16x2(x)
    print(f"split_f16x2 [20, 4096]: {graph_time(split) / n:5.2f} us per launch in a graph (a kernel with one round trip)")


if __name__ == "__main__":
    main()
