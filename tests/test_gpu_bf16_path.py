"""Parity of the BENCHMARKED configuration (`-m gpu`): the bf16 path - matrix-core attention kernels, the
weight-streaming decode GEMM with split-K partials summed by the consumer kernels, HIP-graph replay - against
(a) the goldens captured from the real reference head and (b) the fp32 CPU oracle fed the SAME bf16-rounded
weights, which separates the rounding of the weights from the error of the kernels.

bf16 cannot meet the fp32 bar (1e-3 on logits, identical tokens): activations carry 8 mantissa bits.  What is
asserted instead, per golden case and with the REFERENCE's selection injected (SURVEY 7 "top-K boundary"):
  * existence logits: deviation from the reference and from the bf16-weight oracle, top-20 overlap;
  * first decode step: logits against the reference's top-8 values;
  * greedy tokens: where the bf16 decode first leaves the oracle's token sequence, the oracle's own logit margin
    between the two candidates at that step is small - argmax flips happen only at near-ties;
  * the split-K consumer kernels (decode attention, RMSNorm, rotary + cache write) against fp32 torch at the
    benchmark's geometry (32 heads, context 64).
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = ["G1_c1_512_n10", "G2_768x1024_n12", "G4_llm_wide_n6", "G5_c5geo_1024x1344_n8", "G6_llm_7b_width_n6"]
# Bounds.  The yardstick is PyTorch's own bf16 CPU path: the oracle run with bf16 weights AND bf16 activations is what
# the reference's modules compute when the model is cast to bf16 (HF semantics: finfo(bf16).min masks, fp32 softmax
# and RMSNorm statistics cast back).  The HIP bf16 path keeps fp32 accumulators through more of the chain, so it must
# stay at least as close to the fp32 reference as that - with a floor for cases where both are tiny.
VS_TORCH_BF16 = 1.25
LOGIT_FLOOR = 0.12
FIRST_LOGIT_FLOOR = 0.45
FLIP_MARGIN = 1.0          # oracle logit margin at a greedy-token flip (logit std ~3); measured <= 0.64


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def _head(cfg, w, dtype, **kw):
    from openpsg_amd.head import RelationTransformerHeadV4
    h = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", qformer_vocab_size=cfg.qformer.vocab,
                                  llm_config=cfg.llm, llm_feature_size=cfg.llm.hidden, tokenizers="word",
                                  max_object_num=cfg.max_object_num, on_parse_error="skip", **kw)
    h.load_weights(w)
    return h


def _oracle_run(O, w, cfg, scene, sel, suppress):
    ids = [int(i) for i in scene["object_id_list"]]
    qids, qmask = H.qformer_prompts(scene)
    with torch.no_grad():
        orq = O.relation_query(w, cfg, scene["mask_features"], scene["img_meta"], ids, scene["pan_results"], qids, qmask)
        pids, pmask = H.llm_prompts(scene, sel)
        gens = []
        for i, si in enumerate(sel):
            x, mask = O.llm_inputs(w, orq["pair_feature"][si], pids[i], pmask[i])
            gens.append(O.llm_generate(w, cfg, x, mask, suppress_eos=suppress))
    return orq["exist_logit"].float().numpy(), gens


@pytest.fixture(scope="module", params=CASES)
def bf16_run(request):
    from oracle import psg_oracle as O
    g, cfg, w, scene = H.load_case(request.param)
    suppress = bool(g["suppress_eos"])
    head = _head(cfg, w, "bf16", suppress_eos=suppress)
    dev = _dev()
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names,
                                 scene["pan_results"].to(dev))
    ref_sel = torch.from_numpy(g["selected"].astype(np.int32)).to(dev)
    dec = head.decode_selected(rq, names, selected=ref_sel)              # the reference's selection, injected
    torch.cuda.synchronize()
    sel = g["selected"].tolist()
    # fp32 oracle on bf16-ROUNDED WEIGHTS: isolates weight rounding; torch bf16: weights and activations in bf16
    w_rounded = {k: (v.bfloat16().float() if v.dim() >= 2 else v.clone()) for k, v in w.items()}
    lw, gens_w = _oracle_run(O, w_rounded, cfg, scene, sel, suppress)
    lt, gens_t = _oracle_run(O, {k: v.bfloat16() for k, v in w.items()}, cfg, scene, sel, suppress)
    return dict(g=g, cfg=cfg, head=head, rq=rq, dec=dec, logit_w=lw, gens_w=gens_w, logit_t=lt, gens_t=gens_t,
                name=request.param)


def test_bf16_existence_logits_vs_reference(bf16_run):
    r = bf16_run
    g = r["g"]
    logit = r["rq"]["exist_logit"].cpu().numpy()
    e_ref = np.abs(logit - g["exist_logit"]).max()
    e_w = np.abs(r["logit_w"] - g["exist_logit"]).max()
    e_t = np.abs(r["logit_t"] - g["exist_logit"]).max()
    e_kernel = np.abs(logit - r["logit_w"]).max()
    k = min(20, logit.size)
    overlap = len(set(r["rq"]["selected"].cpu().tolist()) & set(g["selected"].tolist()))
    t_overlap = len(set(np.argsort(-r["logit_t"], kind="stable")[:k].tolist()) & set(g["selected"].tolist()))
    print(f"{r['name']}: max |existence logit - reference|: HIP bf16 {e_ref:.3e}; torch-CPU bf16 {e_t:.3e}; fp32 oracle "
          f"on bf16-rounded weights {e_w:.3e} (weight rounding alone); HIP bf16 vs that oracle {e_kernel:.3e} "
          f"(activation rounding + kernels); top-{k} overlap HIP {overlap}/{k}, torch-CPU bf16 {t_overlap}/{k}")
    assert e_ref < max(VS_TORCH_BF16 * e_t, LOGIT_FLOOR)
    assert overlap >= min(t_overlap, int(0.9 * k)) - 1


def _no_eos(v, eos):
    v = np.array(v, dtype=np.float32)
    v[eos] = 0.0                                                  # suppress_eos writes -inf there on one side only
    return v


def test_bf16_first_step_logits_vs_reference(bf16_run):
    r = bf16_run
    g = r["g"]
    eos = r["cfg"].llm.eos
    fl = r["dec"]["first_logits"].float().cpu().numpy()
    K = fl.shape[0]
    hip = max(float(np.abs(fl[i][g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max()) for i in range(K))
    tor = max(float(np.abs(r["gens_t"][i][1][0].numpy()[g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max())
              for i in range(K))
    kern = max(float(np.abs(_no_eos(fl[i], eos) - _no_eos(r["gens_w"][i][1][0].numpy(), eos)).max()) for i in range(K))
    print(f"{r['name']}: first-step logits, max |x - reference top-8|: HIP bf16 {hip:.3e}, torch-CPU bf16 {tor:.3e}; "
          f"HIP bf16 vs fp32 oracle on bf16-rounded weights over the whole vocabulary {kern:.3e}")
    assert hip < max(VS_TORCH_BF16 * tor, FIRST_LOGIT_FLOOR)


def _first_diff(a, b):
    return next((s for s in range(min(len(a), len(b))) if a[s] != b[s]), None if len(a) == len(b) else min(len(a), len(b)))


def test_bf16_greedy_tokens_leave_the_reference_only_at_near_ties(bf16_run):
    r = bf16_run
    g = r["g"]
    toks = r["dec"]["tokens_host"]
    K = toks.shape[0]
    n_tok = hip_match = tor_match = hip_equal = tor_equal = 0
    worst_margin = 0.0
    for i in range(K):
        want = g["gen_tokens"][i]
        want = want[want >= 0].tolist()
        got = [int(t) for t in toks[i] if t >= 0]
        tor = r["gens_t"][i][0]
        n_tok += len(want)
        for seq, is_hip in ((got, True), (tor, False)):           # tokens matched before the first divergence
            fd = _first_diff(seq, want)
            m = len(want) if fd is None else fd
            if is_hip:
                hip_match += m
                hip_equal += fd is None
            else:
                tor_match += m
                tor_equal += fd is None
        # where the HIP decode leaves the sequence of the fp32 oracle on the same (bf16-rounded) weights, that
        # oracle's own margin between the two candidates must be small: flips happen at near-ties only
        otoks, ologits = r["gens_w"][i]
        fo = _first_diff(got, otoks)
        if fo is not None and fo < min(len(got), len(otoks)):
            lg = ologits[fo]
            worst_margin = max(worst_margin, float(lg[otoks[fo]] - lg[got[fo]]))
    print(f"{r['name']}: pairs decoding the reference's exact sequence: HIP bf16 {hip_equal}/{K}, torch-CPU bf16 "
          f"{tor_equal}/{K}; tokens matched before the first divergence: HIP {hip_match}/{n_tok}, torch-CPU bf16 "
          f"{tor_match}/{n_tok}; largest fp32 margin at a HIP flip {worst_margin:.3e} (logit std ~3)")
    assert worst_margin < FLIP_MARGIN
    assert hip_match >= 0.8 * tor_match - 8


# ---- split-K consumer kernels at the benchmark's geometry -----------------------------------------------------------
def _rope_ref(x, cos, sin):
    """x [..., 128] fp32, cos/sin [..., 64]: HF half-split rotary (HF-LL:130-160)."""
    a, b = x[..., :64], x[..., 64:]
    return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1)


def _split(t, S, gen):
    """fp32 [rows, cols] -> split-K partials [S, rows, cols] that sum to it (random decomposition)."""
    if S == 0:
        return None
    parts = torch.randn((S,) + tuple(t.shape), generator=gen).to(t.device) * 0.5
    parts[0] += t - parts.sum(0)
    return parts.contiguous()


@pytest.mark.parametrize("splits", [0, 4, 8])
@pytest.mark.parametrize("rows,heads,ctx", [(20, 32, 64), (3, 2, 40)])
def test_decode_attn_split_inputs_vs_fp32_torch(splits, rows, heads, ctx):
    """psg_decode_attn<bf16> (HF-LL:130-160, 191-214 for the newest token of every pair): rotary, cache append and
    attention over the cached prefix, with the q/k/v projection arriving as bf16 or as `splits` fp32 partials."""
    from openpsg_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(100 * splits + rows)
    D = heads * 128
    pos = torch.randint(1, ctx - 1, (rows,), generator=gen)
    pos[0] = ctx - 1                                               # a full context
    if rows > 1:
        pos[1] = 0                                                 # the very first token: attends to itself only
    qkv = torch.randn(rows, 3 * D, generator=gen)
    kc0 = torch.randn(rows, heads, ctx, 128, generator=gen).bfloat16()
    vc0 = torch.randn(rows, heads, ctx, 128, generator=gen).bfloat16()
    ang = torch.arange(ctx, dtype=torch.float32)[:, None] / (10000 ** (torch.arange(0, 128, 2).float() / 128))[None]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    kc, vc = kc0.clone().to(dev), vc0.clone().to(dev)
    for r in range(rows):                                          # unwritten cache rows must never be read
        kc[r, :, int(pos[r]):] = float("nan")
        vc[r, :, int(pos[r]):] = float("nan")
    pair = torch.arange(rows, dtype=torch.int32, device=dev)
    pos_d = pos.to(torch.int32).to(dev)
    if splits == 0:
        x_in = qkv.bfloat16().to(dev)
        qkv_eff = x_in.float().cpu()
    else:
        x_in = ops.Partials(_split(qkv.to(dev), splits, gen))
        qkv_eff = x_in.t.sum(0).cpu()
    out = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    ops.decode_attn(x_in, pair, pos_d, (cos, sin), heads, 128, ctx, kc, vc, out)
    torch.cuda.synchronize()
    # fp32 reference
    q, k, v = [t.view(rows, heads, 128) for t in qkv_eff.split(D, dim=1)]
    c, s = cos.cpu()[pos][:, None, :], sin.cpu()[pos][:, None, :]
    qr, kr = _rope_ref(q, c, s), _rope_ref(k, c, s)
    ref = torch.empty(rows, heads, 128)
    for r in range(rows):
        p = int(pos[r])
        K_ = torch.cat([kc0[r, :, :p].float(), kr[r][:, None]], dim=1)      # [heads, p+1, 128]
        V_ = torch.cat([vc0[r, :, :p].float(), v[r][:, None]], dim=1)
        sc = torch.einsum("hd,hjd->hj", qr[r], K_) / 128 ** 0.5
        ref[r] = torch.einsum("hj,hjd->hd", torch.softmax(sc, -1), V_)
    err = (out.float().cpu().view(rows, heads, 128) - ref).abs().max().item()
    kerr = max((kc[r, :, int(pos[r])].float().cpu() - kr[r]).abs().max().item() for r in range(rows))
    verr = max((vc[r, :, int(pos[r])].float().cpu() - v[r]).abs().max().item() for r in range(rows))
    print(f"decode_attn splits={splits} rows={rows} heads={heads}: out err {err:.3e}, K append err {kerr:.3e}, "
          f"V append err {verr:.3e}")
    assert err < 4e-2 and kerr < 4e-2 and verr < 4e-2
    for r in range(rows):                                          # nothing but the new row was written
        p = int(pos[r])
        assert torch.equal(kc[r, :, :p].cpu(), kc0[r, :, :p]) and torch.isnan(kc[r, :, p + 1:].float()).all()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 6e-3), (torch.bfloat16, 4e-2)])
def test_decode_attn_every_dtype_long_contexts_and_poisoned_rows(dtype, tol):
    """psg_decode_attn in the three cache types (fp32 = the headline's KV cache) over contexts of 0 .. 199 cached
    positions - the 64-key passes of the round-6 layout (two whole cache rows per load instruction, a 32-lane
    transposing sum per key) with full, partial and empty last passes - with every unwritten cache row set to NaN:
    against fp32 torch (HF-LL:130-160, 191-214)."""
    from openpsg_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(7)
    heads, ctx = 4, 200
    D = heads * 128
    pos = torch.tensor([0, 1, 2, 15, 16, 17, 31, 33, 47, 48, 63, 64, 65, 79, 80, 127, 128, 129, 150, 199])
    rows = pos.numel()
    qkv = torch.randn(rows, 3 * D, generator=gen)
    kc0 = torch.randn(rows, heads, ctx, 128, generator=gen).to(dtype)
    vc0 = torch.randn(rows, heads, ctx, 128, generator=gen).to(dtype)
    ang = torch.arange(ctx, dtype=torch.float32)[:, None] / (10000 ** (torch.arange(0, 128, 2).float() / 128))[None]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    kc, vc = kc0.clone().to(dev), vc0.clone().to(dev)
    for r in range(rows):
        kc[r, :, int(pos[r]):] = float("nan")
        vc[r, :, int(pos[r]):] = float("nan")
    pair = torch.arange(rows, dtype=torch.int32, device=dev)
    x_in = ops.Partials(_split(qkv.to(dev), 4, gen))
    qkv_eff = x_in.t.sum(0).cpu()
    out = torch.empty(rows, D, device=dev, dtype=dtype)
    ops.decode_attn(x_in, pair, pos.to(torch.int32).to(dev), (cos, sin), heads, 128, ctx, kc, vc, out)
    torch.cuda.synchronize()
    q, k, v = [t.view(rows, heads, 128) for t in qkv_eff.split(D, dim=1)]
    c, s_ = cos.cpu()[pos][:, None, :], sin.cpu()[pos][:, None, :]
    qr, kr = _rope_ref(q, c, s_), _rope_ref(k, c, s_)
    worst = 0.0
    for r in range(rows):
        p = int(pos[r])
        K_ = torch.cat([kc0[r, :, :p].float(), kr[r][:, None].to(dtype).float()], dim=1)
        V_ = torch.cat([vc0[r, :, :p].float(), v[r][:, None].to(dtype).float()], dim=1)
        sc = torch.einsum("hd,hjd->hj", qr[r].to(dtype).float(), K_) / 128 ** 0.5
        ref = torch.einsum("hj,hjd->hd", torch.softmax(sc, -1), V_)
        got = out[r].float().cpu().view(heads, 128)
        assert torch.isfinite(got).all(), f"row {r} (pos {p}) read an unwritten cache row"
        worst = max(worst, (got - ref).abs().max().item())
        assert torch.equal(kc[r, :, :p].cpu(), kc0[r, :, :p]) and torch.isnan(kc[r, :, p + 1:].float()).all()
        assert torch.isfinite(kc[r, :, p].float()).all() and torch.isfinite(vc[r, :, p].float()).all()
    print(f"decode_attn {dtype}: max |out - fp32 torch| = {worst:.3e} over contexts 0..199")
    assert worst < tol


@pytest.mark.parametrize("splits", [0, 4, 8])
@pytest.mark.parametrize("rows,D", [(20, 4096), (900, 4096), (5, 512)])
def test_rmsnorm_split_delta_vs_fp32_torch(splits, rows, D):
    """psg_rmsnorm (HF-LL:53-67 behind the residual add of HF-LL decoder layer): resid += delta, out = norm(resid),
    with delta as bf16 or as fp32 split-K partials."""
    from openpsg_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(7 * splits + rows)
    resid = torch.randn(rows, D, generator=gen).bfloat16().to(dev)
    delta = torch.randn(rows, D, generator=gen)
    w = (1 + 0.1 * torch.randn(D, generator=gen)).to(dev)
    if splits == 0:
        d_in = delta.bfloat16().to(dev)
        d_eff = d_in.float()
    else:
        d_in = ops.Partials(_split(delta.to(dev), splits, gen))
        d_eff = d_in.t.sum(0)
    res = resid.clone()
    out = torch.empty_like(resid)
    ops.rmsnorm(res, d_in, w, 1e-5, out)
    torch.cuda.synchronize()
    # HF bf16 semantics: the projection output is bf16, the residual add is bf16 (llm.py keeps both roundings)
    want_res = (resid.float() + d_eff.bfloat16().float()).bfloat16().float()
    # HF normalises the residual stream as stored (bf16); a fused kernel may normalise the fp32 sum: either reading
    # must hold to bf16 rounding
    xr = res.float()
    want = w * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))
    rel = lambda a, b: ((a - b).abs() / (1 + b.abs())).max().item()            # noqa: E731  bf16: 2^-8 relative
    e_res, e_out = rel(res.float(), want_res), rel(out.float(), want)
    e_out32 = rel(out.float(), w * (want_res * torch.rsqrt(want_res.pow(2).mean(-1, keepdim=True) + 1e-5)))
    print(f"rmsnorm splits={splits} rows={rows} D={D}: resid rel err {e_res:.3e}, out rel err {e_out:.3e} "
          f"(vs the fp32 residual: {e_out32:.3e})")
    assert e_res < 8e-3 and min(e_out, e_out32) < 8e-3          # one bf16 ulp (the fp32 split sum may round the other way)


@pytest.mark.parametrize("splits", [0, 4, 8])
def test_rope_kvwrite_split_inputs_vs_fp32_torch(splits):
    """psg_rope_kvwrite (HF-LL:130-160 + cache update) on a prompt batch with padding rows (pos = -1)."""
    from openpsg_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(31 + splits)
    K, S, heads, ctx = 5, 13, 32, 64
    D = heads * 128
    rows = K * S
    lens = torch.tensor([13, 9, 13, 1, 7])
    t = torch.arange(S)[None, :].expand(K, -1)
    pos = torch.where(t < lens[:, None], t, torch.full_like(t, -1)).reshape(-1)
    pair = torch.arange(K)[:, None].expand(-1, S).reshape(-1)
    qkv = torch.randn(rows, 3 * D, generator=gen)
    ang = torch.arange(ctx, dtype=torch.float32)[:, None] / (10000 ** (torch.arange(0, 128, 2).float() / 128))[None]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    if splits == 0:
        x_in = qkv.bfloat16().to(dev)
        eff = x_in.float().cpu()
    else:
        x_in = ops.Partials(_split(qkv.to(dev), splits, gen))
        eff = x_in.t.sum(0).cpu()
    q_out = torch.zeros(rows, D, device=dev, dtype=torch.bfloat16)
    kc = torch.full((K, heads, ctx, 128), float("nan"), device=dev, dtype=torch.bfloat16)
    vc = torch.full_like(kc, float("nan"))
    ops.rope_kvwrite(x_in, pair.to(torch.int32).to(dev), pos.to(torch.int32).to(dev), (cos, sin), heads, 128, ctx,
                     q_out, kc, vc)
    torch.cuda.synchronize()
    q, k, v = [x.view(rows, heads, 128) for x in eff.split(D, dim=1)]
    ok = pos >= 0
    pc = pos.clamp(min=0)
    c, s = cos.cpu()[pc][:, None, :], sin.cpu()[pc][:, None, :]
    qr, kr = _rope_ref(q, c, s), _rope_ref(k, c, s)
    eq = (q_out.float().cpu().view(rows, heads, 128) - qr)[ok].abs().max().item()
    kcc, vcc = kc.float().cpu(), vc.float().cpu()
    ek = max((kcc[int(pair[r]), :, int(pos[r])] - kr[r]).abs().max().item() for r in range(rows) if ok[r])
    ev = max((vcc[int(pair[r]), :, int(pos[r])] - v[r]).abs().max().item() for r in range(rows) if ok[r])
    print(f"rope_kvwrite splits={splits}: q err {eq:.3e}, k err {ek:.3e}, v err {ev:.3e}")
    assert eq < 4e-2 and ek < 4e-2 and ev < 4e-2
    for kk in range(K):                                            # rows past a pair's length stay untouched
        assert torch.isnan(kcc[kk, :, int(lens[kk]):]).all()


# ---- fp16 / mixed modes (BASELINE config 5 names fp16; mixed = the bench headline) -----------------------------------------
_MODE_TABLE = {}


@pytest.mark.parametrize("case", CASES)
def test_fp16_and_mixed_modes_vs_reference(case):
    """The same matrix-core kernels with the fp16 MFMA opcodes (11 mantissa bits instead of bf16's 8), and the mixed mode
    (fp16 operands, fp32 Llama residual stream) that bench.py reports as its headline.  With the reference's selection
    injected, on every golden: existence logits within 0.03 of the reference (bf16: 0.06-0.2), the reference's top-20
    reproduced, most selected pairs decoding the reference's exact token sequence - next to what rounding the WEIGHTS to
    fp16 costs alone (fp32 oracle on fp16-rounded weights: the floor of any implementation with 16-bit weights)."""
    from oracle import psg_oracle as O
    g, cfg, w, scene = H.load_case(case)
    dev = _dev()
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    suppress = bool(g["suppress_eos"])
    sel = g["selected"].tolist()
    res = {}
    modes = ("fp16", "mixed", "mixed_q32", "bf16") if case in ("G1_c1_512_n10", "G5_c5geo_1024x1344_n8") else \
        ("fp16", "mixed", "bf16")                                   # the fp32 Q-Former residual chain on two cases only
    for dt in modes:
        head = _head(cfg, w, dt, suppress_eos=suppress)
        rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names,
                                     scene["pan_results"].to(dev))
        dec = head.decode_selected(rq, names, selected=torch.from_numpy(g["selected"].astype(np.int32)).to(dev))
        e_logit = np.abs(rq["exist_logit"].cpu().numpy() - g["exist_logit"]).max()
        fl = dec["first_logits"].float().cpu().numpy()
        e_first = max(float(np.abs(fl[i][g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max()) for i in range(fl.shape[0]))
        toks = dec["tokens_host"]
        exact = 0
        for i in range(toks.shape[0]):
            want = g["gen_tokens"][i]
            exact += [int(t) for t in toks[i] if t >= 0] == want[want >= 0].tolist()
        overlap = len(set(rq["selected"].cpu().tolist()) & set(sel))
        res[dt] = (float(e_logit), e_first, exact, overlap)
        assert rq["hidden"].dtype == (torch.bfloat16 if dt == "bf16" else torch.float16)
        assert (head.rq_engine.res32, head.llm_engine.resid_dtype == torch.float32) == \
            {"fp16": (False, False), "bf16": (False, False), "mixed": (False, True), "mixed_q32": (True, True)}[dt]
        del head, rq, dec
        torch.cuda.empty_cache()
    # the floor: fp32 arithmetic on fp16-rounded weights
    wr = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in w.items()}
    lw, gens_w = _oracle_run(O, wr, cfg, scene, sel, suppress)
    f_logit = float(np.abs(lw - g["exist_logit"]).max())
    f_exact = 0
    for i in range(len(sel)):
        want = g["gen_tokens"][i]
        f_exact += gens_w[i][0] == want[want >= 0].tolist()
    _MODE_TABLE[case] = (res, f_logit, f_exact)
    res.setdefault("mixed_q32", res["mixed"])
    print(f"{case}: max |existence logit - reference|: fp16 {res['fp16'][0]:.3e} / mixed {res['mixed'][0]:.3e} / mixed + fp32 "
          f"Q-Former residual {res['mixed_q32'][0]:.3e} / bf16 {res['bf16'][0]:.3e} (fp16 weight rounding alone {f_logit:.3e}); "
          f"first-step logits fp16 {res['fp16'][1]:.3e} / mixed {res['mixed'][1]:.3e} / mixed_q32 {res['mixed_q32'][1]:.3e} / "
          f"bf16 {res['bf16'][1]:.3e}; pairs with the reference's exact tokens fp16 {res['fp16'][2]}/20, mixed "
          f"{res['mixed'][2]}/20, mixed_q32 {res['mixed_q32'][2]}/20, bf16 {res['bf16'][2]}/20 (fp16 weight rounding alone "
          f"{f_exact}/20); top-20 overlap fp16 {res['fp16'][3]}, mixed {res['mixed'][3]}, mixed_q32 {res['mixed_q32'][3]}, "
          f"bf16 {res['bf16'][3]}")
    for dt in ("fp16", "mixed", "mixed_q32"):
        assert res[dt][0] < 0.03, f"{dt}: existence logits {res[dt][0]} from the reference"
        assert res[dt][0] < max(0.5 * res["bf16"][0], 0.03) and res[dt][1] < max(0.5 * res["bf16"][1], 0.1)
        # the reference's top-20 (one near-tie at the cut of G5 is the only difference measured on the five goldens)
        assert res[dt][3] >= min(20, len(g["exist_logit"])) - 1
        assert res[dt][2] >= res["bf16"][2]
        # measured 14-19 of 20 against 17-19 for weight rounding alone (G5, the worst case, moves between 14 and 16 from
        # run to run: the library GEMM's kernel choice is not pinned)
        assert res[dt][2] >= min(14, f_exact - 4), f"{dt}: {res[dt][2]}/20 exact sequences, weight-rounding floor {f_exact}/20"


@pytest.mark.parametrize("M,N,K", [(20, 4096, 4096), (7, 32000, 4096), (32, 768, 2752)])
def test_skinny_gemm_fp16_vs_fp32_reference(M, N, K):
    from openpsg_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=gen).to(dev).half()
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev).half()
    y = ops.skinny_gemm(x, w).reduce(torch.float16)
    ref = x.float() @ w.float().t()
    err = (y.float() - ref).abs().max().item()
    print(f"fp16 skinny GEMM M={M} N={N} K={K}: err {err:.3e}")
    assert err < 4e-3


def test_indexed_cross_attention_equals_the_expanded_queries_bit_for_bit():
    """psg_qformer_cross_attn_indexed (queries stored once per prompt, looked up through the pair -> prompt index by the
    LDS-DMA kernel) against psg_qformer_cross_attn on the expanded [P x 33] query rows (HF-IB:464-496)."""
    from openpsg_amd import ops
    from openpsg_amd.synthetic import make_scene
    dev = "cuda:0"
    N, L, H, heads, nq = 23, 256, 768, 12, 33
    sc = make_scene((1024, 1024), N, seed=2, device=dev, features=False)
    grid = ops.mask_grid(sc["pan_results"], (1024, 1024), (1024, 1024), (16, 16))
    bits = ops.object_bitmasks(grid, torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32, device=dev))
    P, U = N * N, 37
    g = torch.Generator(device=dev).manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        q_u = torch.randn(U * nq, H, generator=g, device=dev).to(dt)
        k = torch.randn(L, H, generator=g, device=dev).to(dt)
        v = torch.randn(L, H, generator=g, device=dev).to(dt)
        inv = torch.randint(0, U, (P,), generator=g, device=dev).to(torch.int32)
        pidx = torch.arange(P, device=dev, dtype=torch.int32)
        rows = (inv.long()[:, None] * nq + torch.arange(nq, device=dev)[None, :]).reshape(-1)
        q_full = q_u[rows].contiguous()
        want = ops.qformer_cross_attn(q_full, k, v, bits, pidx, N, nq, heads)
        q_cls = q_u[inv.long() * nq].contiguous()
        got = ops.qformer_cross_attn_indexed(q_u, inv, q_cls, k, v, bits, pidx, N, heads)
        assert got is not None and torch.equal(got, want)
    assert ops.qformer_cross_attn_indexed(q_u.float(), inv, q_cls.float(), k.float(), v.float(), bits, pidx, N, heads) is None
