"""Kernel micro-benchmarks on one MI355X (development aid): achieved GB/s / TFLOP/s per kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpsg_amd import ops, _lib  # noqa: E402


def timeit(fn, iters=30, warm=5, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)                      # evict L2/MALL (512 MiB touch)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def skinny(M, dtype=torch.bfloat16, w16=False):
    """w16: fp32 rows against weights stored as fp16 (psg_skinny_gemm_w16)"""
    dev = torch.device("cuda:0")
    esz = 2 if w16 else torch.empty(0, dtype=dtype).element_size()
    shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008),
              ("lm_head", 32000, 4096)]
    tot_b, tot_s, tot_l = 0, 0.0, 0.0
    only = os.environ.get("PSG_BENCH_SHAPES")               # e.g. "gate_up" or "qkv,o": PMC passes of one shape
    if only:
        shapes = [s_ for s_ in shapes if s_[0] in only.split(",")]
    nolib = os.environ.get("PSG_BENCH_NOLIB") == "1"
    for name, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(dtype)
        ncopy = max(2, int(800e6 / (N * K * esz)) + 1)     # rotate > MALL-size of weights: always cold
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.float16 if w16 else dtype) for _ in range(ncopy)]
        it = [0]
        xs2 = ops.split_f16x2(x) if w16 == "split" else None
        split_in_loop = os.environ.get("PSG_BENCH_SPLIT_IN_LOOP") == "1"

        REP = 16                                            # back-to-back launches per timing sample

        def run_s():
            for _ in range(REP):
                it[0] += 1
                if w16 == "split":
                    x2, inv = ops.split_f16x2(x) if split_in_loop else xs2
                    ops.split_gemm_w16(x2, inv, ws[it[0] % ncopy], int(os.environ.get("PSG_BENCH_PAIR_MODE", "0")))
                elif w16:
                    ops.skinny_gemm_w16(x, ws[it[0] % ncopy])
                else:
                    ops.skinny_gemm(x, ws[it[0] % ncopy])

        def run_l():
            for _ in range(REP):
                it[0] += 1
                torch.nn.functional.linear(x, ws[it[0] % ncopy])
        ts, _ = timeit(run_s, iters=12, warm=2)
        tl = None if (nolib or w16) else timeit(run_l, iters=12, warm=2)[0] / REP   # PSG_BENCH_NOLIB=1: no library column at all
        ts = ts / REP
        gb = N * K * esz / 1e9
        lib = "" if tl is None else f" | library (F.linear) {tl:8.1f} us = {gb / tl * 1e6:7.0f} GB/s"
        print(f"skinny {str(dtype)[6:]} M={M} {name:8s} N={N:6d} K={K:6d}: {ts:8.1f} us = {gb / ts * 1e6:7.0f} GB/s{lib}",
              flush=True)
        mult = 1 if name == "lm_head" else 32
        tot_b += gb * mult
        tot_s += ts * mult
        tot_l += (tl or 0.0) * mult
    lib = "" if (nolib or w16) else f", library {tot_l / 1e3:.2f} ms ({tot_b / tot_l * 1e6:.0f} GB/s)"
    print(f"  one decode step (32 layers + lm_head): {tot_b:.2f} GB, skinny {tot_s / 1e3:.2f} ms "
          f"({tot_b / tot_s * 1e6:.0f} GB/s){lib}")


def xattn(N=50, L=256):
    dev = torch.device("cuda:0")
    P = N * N
    q = torch.randn(P * 33, 768, device=dev).bfloat16()
    k = torch.randn(L, 768, device=dev).bfloat16()
    v = torch.randn(L, 768, device=dev).bfloat16()
    # realistic pair masks: the synthetic scene's rectangles through the mask kernels
    from openpsg_amd.synthetic import make_scene
    sc = make_scene((1024, 1024), N, seed=0, device="cuda:0", features=False)
    grid = ops.mask_grid(sc["pan_results"], (1024, 1024), (1024, 1024), (16, 16))
    bits = ops.object_bitmasks(grid, torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32,
                                                  device=dev))
    dens = sum(bin(int(b) & (2**64 - 1)).count("1") for b in bits.cpu().flatten().tolist()) / (N * L)
    pidx = torch.arange(P, device=dev, dtype=torch.int32)
    out = torch.empty_like(q)
    flops = 4.0 * P * 33 * L * 768
    nbytes = 2.0 * P * 33 * 768 * 2                                 # Q in + context out
    for name, var in (("LDS-DMA kernel", _lib.PSG_XATTN_MFMA), ("first-generation kernel", _lib.PSG_XATTN_MFMA_V1)):
        t, tmin = timeit(lambda: ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out, variant=var))
        print(f"cross_attn [{name}] N={N} L={L} (object mask density {dens:.3f}): {t:.1f} us (min {tmin:.1f}) = "
              f"{flops / t / 1e6:.1f} dense-equivalent TFLOP/s ({flops / t / 1e6 / 2500 * 100:.1f}% of 2.5 PF dense "
              f"bf16); Q + context traffic {nbytes / 1e6:.0f} MB -> {nbytes / t / 1e6:.2f} TB/s "
              f"({nbytes / t / 1e6 / 8 * 100:.0f}% of 8 TB/s)")


def mall():
    """Does the skinny GEMM run faster when part of its weights was pulled into L2 / Infinity Cache just
    before?  (decides whether the small kernels between the GEMMs should carry prefetch blocks)"""
    dev = torch.device("cuda:0")
    M = 20
    for name, N, K in [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
        x = torch.randn(M, K, device=dev).bfloat16()
        ncopy = max(2, int(800e6 / (N * K * 2)) + 1)
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(ncopy)]
        for frac in (0.0, 0.1, 0.25, 0.5, 1.0):
            ts = []
            for it in range(40):
                w = ws[it % ncopy]
                flat = w.view(-1).view(torch.int32)
                n = int(flat.numel() * frac)
                if n:
                    flat[:n].sum()                                  # touch: reads the first `frac` of the weights
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.skinny_gemm(x, w)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) * 1e3)
            ts.sort()
            print(f"mall {name:8s} prefetched {frac:4.2f} ({N * K * 2 * frac / 1e6:6.1f} MB): "
                  f"gemm {ts[len(ts) // 2]:6.1f} us (min {ts[0]:6.1f})", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["skinny", "xattn"])
    a = ap.parse_args()
    if "skinny" in a.what:
        skinny(20)
    if "skinny32" in a.what:                  # the fp32 instantiation (psg_gemm_f32.hip); skinny32_m<M> for other row counts
        skinny(20, torch.float32)
    for wname in a.what:
        if wname.startswith("skinny32_m"):
            skinny(int(wname[10:]), torch.float32)
    for wname in a.what:
        if wname.startswith("skinnysplit"):   # fp32 rows as two fp16 planes x fp16-valued weights (psg_split_gemm_w16)
            skinny(int(wname[13:]) if wname.startswith("skinnysplit_m") else 20, torch.float32, w16="split")
    for wname in a.what:
        if wname.startswith("skinnyw16"):     # skinnyw16 or skinnyw16_m<M>
            skinny(int(wname[11:]) if wname.startswith("skinnyw16_m") else 20, torch.float32, w16=True)
    if "mall" in a.what:
        mall()
    if "xattn" in a.what:
        if "only100" not in a.what:
            xattn(50, 256)
        if "only50" not in a.what:
            xattn(100, 256)


def decode_attn():
    dev = torch.device("cuda:0")
    K, heads, ctx, D, layers = 20, 32, 64, 4096, 32
    pair = torch.arange(K, device=dev, dtype=torch.int32)
    ang = torch.arange(128, dtype=torch.float32)[:, None] / (10000.0 ** (torch.arange(0, 128, 2) / 128))[None, :]
    inv = (ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous())
    for dt in (torch.float32, torch.float16):
        # one cache per layer, rotated: as cold as in the decode step (32 x 2 x 21 MB of fp32 >> L2; the Infinity Cache holds
        # part of it, as it does in situ)
        kcs = [torch.randn(K, heads, ctx, 128, device=dev).to(dt) for _ in range(layers)]
        vcs = [torch.randn(K, heads, ctx, 128, device=dev).to(dt) for _ in range(layers)]
        out = torch.empty(K, D, device=dev, dtype=dt)
        for pos_v in (1, 50, 60):
            pos = torch.full((K,), pos_v, device=dev, dtype=torch.int32)
            for splits in (0, 8):
                if splits:
                    qkv = ops.Partials(torch.randn(splits, K, 3 * D, device=dev))
                else:
                    qkv = torch.randn(K, 3 * D, device=dev).to(dt)

                def run():
                    for l in range(layers):
                        ops.decode_attn(qkv, pair, pos, inv, heads, 128, ctx, kcs[l], vcs[l], out)
                t, _ = timeit(run, iters=8, warm=2)
                nbytes = 2 * K * heads * pos_v * 128 * kcs[0].element_size()
                print(f"decode_attn {str(dt)[6:]} pos={pos_v} splits={splits}: {t / layers:.2f} us per launch, {nbytes / 1e6:.1f} MB of "
                      f"K/V = {nbytes / (t / layers) / 1e6:.2f} TB/s")


if __name__ == "__main__" and "decode_attn" in sys.argv:
    decode_attn()


def pool():
    from openpsg_amd.synthetic import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene((1024, 1024), 50, seed=0, device="cuda:0", void_id=0, force_id0=True)
    ids = torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32, device=dev)
    feat, pan = sc["mask_features"], sc["pan_results"]
    t, tmin = timeit(lambda: ops.masked_mean_pool(feat, pan, (1024, 1024), (1024, 1024), ids), iters=20)
    nbytes = feat.numel() * 4
    print(f"masked_mean_pool 1024^2 N=50: {t:.1f} us (min {tmin:.1f}) incl. mask_grid + index passes; feature map "
          f"{nbytes / 1e6:.0f} MB -> {nbytes / tmin / 1e3:.0f} GB/s (read-once upper bound on algorithmic bytes)")


if __name__ == "__main__" and "pool" in sys.argv:
    pool()


if __name__ == "__main__" and "xattn_scale" in sys.argv:
    for n in (4, 12, 25, 50, 71, 100):
        xattn(n, 256)
