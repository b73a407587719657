"""Why is the cross-attention launch 5 us slower inside the relation-query pass than alone?  Times the kernel (events
around ONE launch) behind different predecessors."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops
from openpsg_amd.synthetic import make_scene

dev = torch.device("cuda:0")
N, L, P = 50, 256, 2500
g = torch.Generator(device=dev).manual_seed(3)
h = torch.randn(P * 33, 768, device=dev, generator=g).bfloat16()
Wq = (torch.randn(768, 768, device=dev, generator=g) / 27.7).bfloat16()
q = torch.nn.functional.linear(h, Wq)
q2 = torch.empty_like(q)
k = torch.randn(L, 768, device=dev, generator=g).bfloat16()
v = torch.randn(L, 768, device=dev, generator=g).bfloat16()
sc = make_scene((1024, 1024), N, seed=0, device="cuda:0", features=False)
grid = ops.mask_grid(sc["pan_results"], (1024, 1024), (1024, 1024), (16, 16))
bits = ops.object_bitmasks(grid, torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32, device=dev))
pidx = torch.arange(P, device=dev, dtype=torch.int32)
out = torch.empty_like(q)
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
A = torch.randn(8192, 8192, device=dev).bfloat16()
B = torch.randn(8192, 8192, device=dev).bfloat16()
C = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)


def xa():
    ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out)


def timed(pre, n=40):
    ts = []
    for i in range(n + 5):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        xa()
        e1.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


cases = [
    ("nothing (sync before)", lambda: None),
    ("another cross-attention", xa),
    ("library GEMM writing q (h @ Wq^T -> q)", lambda: torch.nn.functional.linear(h, Wq, out=None) if False else torch.mm(h, Wq.t(), out=q)),
    ("library GEMM writing another buffer", lambda: torch.mm(h, Wq.t(), out=q2)),
    ("own dense GEMM writing q", lambda: ops.dense_gemm(h, Wq, None, out=q) if "out" in ops.dense_gemm.__code__.co_varnames else q.copy_(ops.dense_gemm(h, Wq, None))),
    ("8192^3 bf16 GEMM (1.1 TFLOP, ~1 ms of matrix work)", lambda: torch.mm(A, B, out=C)),
    ("512 MB fill (HBM write, evicts L2 / MALL)", lambda: big.fill_(1)),
    ("q.clone() (q re-written by a copy kernel)", lambda: q.copy_(q2)),
]
q2.copy_(q)
for name, pre in cases:
    try:
        med, mn = timed(pre)
        print(f"{name:55s}: {med:6.1f} us (min {mn:6.1f})", flush=True)
    except Exception as e:
        print(f"{name:55s}: failed {type(e).__name__}: {e}"[:200], flush=True)
