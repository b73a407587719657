"""psg_decode_layer against the launch chain it replaces: one decoder layer of the decode step at Llama-2-7B width,
fp32 weights, M rows, cold weights (NL layers walked in turn: 809 MB each).   python tools/decode_layer_bench.py [M] [NL]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops  # noqa: E402

DEV = "cuda:0"
D, I, HEADS, CTX = 4096, 11008, 32, 64
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s, std=0.02: torch.randn(*s, generator=g, device=DEV) * std   # noqa: E731
    layers = [dict(wqkv=r(3 * D, D), wo=r(D, D), wgu=r(2 * I, D), wdown=r(D, I, std=0.015), ln1=torch.ones(D, device=DEV),
                   ln2=torch.ones(D, device=DEV)) for _ in range(NL)]
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    ang = torch.arange(CTX, dtype=torch.float32)[:, None] * inv[None, :]
    rope = (ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV))
    pos = torch.full((M,), 50, dtype=torch.int32, device=DEV)
    pair = torch.arange(M, dtype=torch.int32, device=DEV)
    resid = torch.randn(M, D, generator=g, device=DEV)
    kc = [torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV) for _ in range(NL)]
    vc = [torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV) for _ in range(NL)]
    n, att, act = torch.empty_like(resid), torch.empty_like(resid), torch.empty((M, I), device=DEV)

    def chain():
        delta = None
        for l, L in enumerate(layers):
            ops.rmsnorm(resid, delta, L["ln1"], 1e-5, n)
            qkv = ops.skinny_gemm(n, L["wqkv"])
            ops.decode_attn(qkv, pair, pos, rope, HEADS, 128, CTX, kc[l], vc[l], att)
            o = ops.skinny_gemm(att, L["wo"])
            ops.rmsnorm(resid, o, L["ln2"], 1e-5, n)
            gu = ops.skinny_gemm(n, L["wgu"])
            ops.silu_mul(gu, act)
            delta = ops.skinny_gemm(act, L["wdown"])

    ws, ncnt = ops.decode_layer_workspace(M, D, I, DEV)
    dparts = [torch.empty((16, M, D), device=DEV) for _ in range(2)]
    counters = torch.zeros(NL * ncnt, device=DEV, dtype=torch.int32)

    def persistent():
        counters.zero_()
        delta = None
        for l, L in enumerate(layers):
            delta = ops.decode_layer(resid, delta, L["ln1"], L["ln2"], L["wqkv"], L["wo"], L["wgu"], L["wdown"], pair, pos,
                                     rope, HEADS, CTX, 1e-5, kc[l], vc[l], ws, counters[l * ncnt:(l + 1) * ncnt], dparts[l & 1])

    dparts2 = torch.empty((2, 16, M, D), device=DEV)
    table = ops.decode_layer_table(layers, kc, vc)

    def stack():
        counters.zero_()
        ops.decode_layers(resid, None, table, NL, pair, pos, rope, HEADS, CTX, 1e-5, I, ws, counters, dparts2)

    def timed(fn, graph):
        fn()
        torch.cuda.synchronize()
        if graph:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fn()
            run = gr.replay
        else:
            run = fn
        run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                run()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 5 / NL * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    nbytes = sum(w.numel() * 4 for k, w in layers[0].items() if k.startswith("w"))
    for name, fn in (("launch chain", chain), ("persistent layer", persistent), ("persistent stack", stack)):
        for graph in (False, True):
            us = timed(fn, graph)
            print(f"M={M} {name:17s} {'graph' if graph else 'eager'}: {us:7.1f} us per layer = {nbytes / us / 1e6:5.2f} TB/s of weights")
    resid.copy_(torch.randn(M, D, generator=g, device=DEV))
    persistent()
    torch.cuda.synchronize()
    print("timeouts:", counters.view(NL, ncnt)[:, 255 * 64].tolist())


if __name__ == "__main__":
    main()
