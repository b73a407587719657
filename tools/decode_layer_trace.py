"""Phase timeline of psg_decode_layer (100 MHz wall-clock stamps per workgroup at its phase boundaries).
   python tools/decode_layer_trace.py [M] [stack]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
D, I, HEADS, CTX = 4096, 11008, 32, 64
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NAMES = ["start", "owner1+arrive", "wait X1", "norm stage", "qkv gemm", "publish HEAD", "attention", "wait ATT", "stage att",
         "o gemm", "publish+wait OSLAB", "owner2+arrive", "wait X2", "norm stage2", "gu gemm", "publish GU + silu", "wait H",
         "stage h", "down gemm"]


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    r = lambda *s, std=0.02: torch.randn(*s, generator=g, device=DEV) * std   # noqa: E731
    NL = 3
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
    ws, ncnt = ops.decode_layer_workspace(M, D, I, DEV)
    dparts = [torch.empty((16, M, D), device=DEV) for _ in range(2)]
    counters = torch.zeros(NL * ncnt, device=DEV, dtype=torch.int32)
    trace = torch.zeros(256 * 24, device=DEV, dtype=torch.int64)

    def run(last_traced):
        counters.zero_()
        delta = None
        for l, L in enumerate(layers):
            if last_traced and l == NL - 1:
                _lib.set_trace_buffer(0, _lib.PSG_TRACE_DECODE_LAYER, trace)
            delta = ops.decode_layer(resid, delta, L["ln1"], L["ln2"], L["wqkv"], L["wo"], L["wgu"], L["wdown"], pair, pos,
                                     rope, HEADS, CTX, 1e-5, kc[l], vc[l], ws, counters[l * ncnt:(l + 1) * ncnt], dparts[l & 1])
        _lib.set_trace_buffer(0, _lib.PSG_TRACE_NONE)
    stack = len(sys.argv) > 2 and sys.argv[2] == "stack"
    if stack:                                                         # the layers chained inside one launch: the LAST one is stamped
        table = ops.decode_layer_table(layers, kc, vc)
        dp2 = torch.empty((2, 16, M, D), device=DEV)

        def run(traced):                                              # noqa: F811
            counters.zero_()
            if traced:
                _lib.set_trace_buffer(0, _lib.PSG_TRACE_DECODE_LAYER, trace)
            ops.decode_layers(resid, None, table, NL, pair, pos, rope, HEADS, CTX, 1e-5, I, ws, counters, dp2)
            _lib.set_trace_buffer(0, _lib.PSG_TRACE_NONE)
    run(False)
    run(False)
    torch.cuda.synchronize()
    run(True)
    torch.cuda.synchronize()
    t = trace.view(256, 24)[:, :19].double().cpu() / 100.0            # us
    t0 = t[:, 0].min()
    t = t - t0
    print(f"M={M}; stamps in us from the first workgroup's start; mean / min / max over 256 workgroups of the END of each phase, and the phase's mean duration")
    for i in range(19):
        dur = (t[:, i] - t[:, i - 1]).mean().item() if i else 0.0
        print(f"{i:2d} {NAMES[i]:22s} end {t[:, i].mean():8.2f} [{t[:, i].min():8.2f} .. {t[:, i].max():8.2f}]   phase {dur:7.2f}")
    print("timeouts:", counters.view(NL, ncnt)[:, 255 * 64].tolist())


if __name__ == "__main__":
    main()
