"""The decode leg at the shape it is benchmarked at (`-m gpu`): Llama-2-7B width - hidden 4096, 32 heads x 128, MLP 11008,
vocabulary 32000 (what the reference instantiates, V4:99-100).

  * G6 (tests/golden/G6_llm_7b_width_n6.npz) was captured from the REAL reference head with a 2-layer model of that
    width (the reference's own llm_truncate_num knob, V4:101-103): the fp32 head must reproduce every greedy token and
    the first-step top-8 logits within 1e-3 (tests/test_gpu_parity.py runs G6 through its fp32 cases too); the 16-bit
    heads (bf16, fp16, mixed = fp16 operands + fp32 residual stream) are bounded with the reference's selection
    injected, next to the floor that rounding the WEIGHTS alone sets (fp32 oracle on rounded weights);
  * the 32-layer engine bench.py times (random bf16 weights generated in HBM) has no CPU oracle that finishes in
    seconds; it is checked through size-independent properties: HIP-graph replay == eager launch sequence, a pair
    decoded in a batch of 20 == the same pair decoded alone or in a batch of 4 (batch invariance of the split-K
    sums: every row's reduction order is fixed by the kernel, not by the batch), chunked natural-EOS graphs ==
    the single worst-case graph.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

CASE = "G6_llm_7b_width_n6"


def _head(cfg, w, dtype, **kw):
    from openpsg_amd.head import RelationTransformerHeadV4
    h = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", qformer_vocab_size=cfg.qformer.vocab,
                                  llm_config=cfg.llm, llm_feature_size=cfg.llm.hidden, tokenizers="word",
                                  max_object_num=cfg.max_object_num, on_parse_error="skip", **kw)
    h.load_weights(w)
    return h


@pytest.fixture(scope="module")
def g6():
    g, cfg, w, scene = H.load_case(CASE)
    assert (cfg.llm.hidden, cfg.llm.heads, cfg.llm.inter, cfg.llm.vocab) == (4096, 32, 11008, 32000)
    return g, cfg, w, scene


def _decode_with_reference_selection(head, g, scene):
    dev = torch.device("cuda:0")
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names,
                                 scene["pan_results"].to(dev))
    dec = head.decode_selected(rq, names, selected=torch.from_numpy(g["selected"].astype(np.int32)).to(dev))
    torch.cuda.synchronize()
    return rq, dec


def _score(g, rq, dec):
    e_logit = float(np.abs(rq["exist_logit"].cpu().numpy() - g["exist_logit"]).max())
    fl = dec["first_logits"].float().cpu().numpy()
    e_first = max(float(np.abs(fl[i][g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max()) for i in range(fl.shape[0]))
    toks = dec["tokens_host"]
    exact = matched = total = 0
    for i in range(toks.shape[0]):
        want = g["gen_tokens"][i]
        want = want[want >= 0].tolist()
        got = [int(t) for t in toks[i] if t >= 0]
        exact += got == want
        fd = next((s for s in range(min(len(got), len(want))) if got[s] != want[s]), min(len(got), len(want)))
        matched += fd
        total += len(want)
    overlap = len(set(rq["selected"].cpu().tolist()) & set(g["selected"].tolist()))
    return dict(logit=e_logit, first=e_first, exact=exact, matched=matched, total=total, overlap=overlap)


def test_fp32_head_reproduces_the_reference_at_7b_width(g6):
    """Every greedy token of all 20 selected pairs and the first-step top-8 logits (1e-3) of the real reference head."""
    g, cfg, w, scene = g6
    head = _head(cfg, w, "fp32", suppress_eos=bool(g["suppress_eos"]))
    rq, dec = _decode_with_reference_selection(head, g, scene)
    s = _score(g, rq, dec)
    print(f"G6 fp32: {s}")
    assert s["logit"] < 1e-3 and s["first"] < 1e-3
    assert s["exact"] == 20 and s["overlap"] == 20
    assert rq["selected"].cpu().tolist() == g["selected"].tolist()


def test_16bit_heads_at_7b_width_are_bounded_by_weight_rounding(g6):
    """bf16 / fp16 / mixed heads against the reference, with the reference's selection injected.  The yardstick is what
    rounding the WEIGHTS to the 16-bit type costs on its own (fp32 oracle on rounded weights: no 16-bit-weight
    implementation can do better): the HIP path must stay within a small factor of that floor."""
    from oracle import psg_oracle as O
    g, cfg, w, scene = g6
    sel = g["selected"].tolist()
    suppress = bool(g["suppress_eos"])
    ids = [int(i) for i in scene["object_id_list"]]
    qids, qmask = H.qformer_prompts(scene)
    pids, pmask = H.llm_prompts(scene, sel)
    floors = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        wr = {k: (v.to(dt).float() if v.dim() >= 2 else v) for k, v in w.items()}
        with torch.no_grad():
            orq = O.relation_query(wr, cfg, scene["mask_features"], scene["img_meta"], ids, scene["pan_results"], qids, qmask)
            e1, exact = 0.0, 0
            for i, si in enumerate(sel):
                x, mask = O.llm_inputs(wr, orq["pair_feature"][si], pids[i], pmask[i])
                toks, lg = O.llm_generate(wr, cfg, x, mask, suppress_eos=suppress)
                want = g["gen_tokens"][i]
                exact += toks == want[want >= 0].tolist()
                e1 = max(e1, float(np.abs(lg[0].numpy()[g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max()))
        floors[name] = dict(logit=float(np.abs(orq["exist_logit"].numpy() - g["exist_logit"]).max()), first=e1, exact=exact)
        del wr
    res = {}
    for mode in ("bf16", "fp16", "mixed"):
        head = _head(cfg, w, mode, suppress_eos=suppress)
        rq, dec = _decode_with_reference_selection(head, g, scene)
        res[mode] = _score(g, rq, dec)
        del head, rq, dec
        torch.cuda.empty_cache()
    for mode in res:
        fl = floors["bf16" if mode == "bf16" else "fp16"]
        print(f"G6 {mode}: logits {res[mode]['logit']:.3e} (weight-rounding floor {fl['logit']:.3e}); first-step logits "
              f"{res[mode]['first']:.3e} (floor {fl['first']:.3e}); exact sequences {res[mode]['exact']}/20 (floor "
              f"{fl['exact']}/20); tokens before the first divergence {res[mode]['matched']}/{res[mode]['total']}; "
              f"top-20 overlap {res[mode]['overlap']}/20")
    for mode in res:
        fl = floors["bf16" if mode == "bf16" else "fp16"]
        assert res[mode]["logit"] < max(3.0 * fl["logit"], 0.02)
        assert res[mode]["first"] < max(3.0 * fl["first"], 0.06)
    # 11 mantissa bits against 8: the fp16 modes sit well inside the bf16 error, and inside the bars the headline
    # mode is held to (existence logits 0.03, the reference's top-20 reproduced, most sequences token-exact)
    for mode in ("fp16", "mixed"):
        assert res[mode]["logit"] < 0.03 and res[mode]["overlap"] == 20
        assert res[mode]["first"] < 0.5 * res["bf16"]["first"] + 0.02
        assert res[mode]["exact"] >= min(16, floors["fp16"]["exact"] - 2)


# ---- the 32-layer engine of bench.py ----------------------------------------------------------------------------------
MODES = dict(bf16=(torch.bfloat16, None), bf16_res32=(torch.bfloat16, torch.float32), fp16=(torch.float16, None),
             mixed=(torch.float16, torch.float32))


@pytest.fixture(scope="module")
def bench_engine():
    """Llama-2-7B shape, 32 layers, random weights generated in HBM at bf16 precision (so that every engine - fp32,
    bf16, fp16 - holds the SAME weight values: what is compared is the arithmetic, not the weight rounding)."""
    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.llm import LlamaDecodeEngine
    from openpsg_amd.weights import make_weights_device
    dev = torch.device("cuda:0")
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=LlamaConfig(layers=32), max_object_num=50)
    w = make_weights_device(cfg, 0, dev, llm_dtype=torch.bfloat16)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    K, Tp = 20, 16
    X = torch.randn((K, 32 + Tp, 4096), device=dev, generator=g).to(torch.bfloat16)
    plen = torch.randint(9, Tp + 1, (K,), device=dev, generator=g, dtype=torch.int32)
    # the truth: the same engine code in fp32 (library GEMMs, fp32 kernels; 27 GB of weights)
    e32 = LlamaDecodeEngine(w, cfg, dev, torch.float32)
    t32, f32 = e32.generate(X.float(), plen, suppress_eos=True, return_first_logits=True)
    torch.cuda.synchronize()
    del e32
    torch.cuda.empty_cache()
    return dict(cfg=cfg, w=w, X=X, plen=plen, t32=t32, f32=f32, dev=dev)


def _engine(be, mode):
    from openpsg_amd.llm import LlamaDecodeEngine
    dt, rdt = MODES[mode]
    return LlamaDecodeEngine(be["w"], be["cfg"], be["dev"], dt, resid_dtype=rdt)


def _vs_truth(be, tokens, first):
    eos = be["cfg"].llm.eos
    a, b = first.float().clone(), be["f32"].float().clone()
    a[:, eos] = 0
    b[:, eos] = 0
    err = (a - b).abs().max(dim=1).values                                         # per pair
    exact = int((tokens == be["t32"]).all(dim=1).sum())
    same = (tokens == be["t32"]).int().cumprod(dim=1).sum().item()               # tokens before the first divergence
    return err, exact, int(same)


_RESULTS = {}


@pytest.mark.parametrize("mode", list(MODES))
def test_32_layer_engine_graph_replay_and_accuracy_against_the_fp32_engine(bench_engine, mode):
    be = bench_engine
    cfg, X, plen = be["cfg"], be["X"].to(MODES[mode][0]), be["plen"]
    eng = _engine(be, mode)
    tg, fg = eng.generate(X, plen, suppress_eos=True, return_first_logits=True)
    tg2 = eng.generate(X, plen, suppress_eos=True)                       # second replay of the same graph
    eng.use_graph = False
    te, fe = eng.generate(X, plen, suppress_eos=True, return_first_logits=True)
    eng.use_graph = True
    torch.cuda.synchronize()
    assert tg.shape == (20, 16) and int(tg.min()) >= 0 and int(tg.max()) < cfg.llm.vocab
    assert torch.equal(tg, te) and torch.equal(tg, tg2), "HIP-graph replay differs from the eager launch sequence"
    assert torch.equal(fg, fe)
    # natural-EOS graphs (chunks of 4 steps) == the single worst-case graph for every pair that never emits EOS
    tn = eng.generate(X, plen, suppress_eos=False)
    torch.cuda.synchronize()
    for i in range(20):
        row = tn[i].tolist()
        if cfg.llm.eos not in row:
            assert row == tg[i].tolist()
    err20, exact20, same20 = _vs_truth(be, tg, fg)
    # the same pairs decoded alone / in a batch of 4: the prompt pass goes through the library GEMM, whose kernel choice
    # depends on the row count, and 32 random layers amplify every rounding difference - so the check is not equality
    # but that a pair decoded in ANOTHER batch is as close to the fp32 engine as it is in the batch of 20
    worst = 0.0
    for idx in ([0], [7], [19], [3, 4, 5, 6]):
        ii = torch.tensor(idx, device=X.device)
        ts, fs = eng.generate(X[ii].contiguous(), plen[ii].contiguous(), suppress_eos=True, return_first_logits=True)
        torch.cuda.synchronize()
        a, b = fs.float().clone(), be["f32"][ii].float().clone()
        a[:, cfg.llm.eos] = 0
        b[:, cfg.llm.eos] = 0
        worst = max(worst, float((a - b).abs().max()))
    _RESULTS[mode] = (float(err20.max()), float(err20.mean()), exact20, same20, worst)
    print(f"32 layers, {mode}: first-step logits vs the fp32 engine: max {err20.max():.3f} mean {err20.mean():.3f} (batch of "
          f"20), max {worst:.3f} (same pairs in batches of 1 and 4); pairs with the fp32 engine's 16 tokens {exact20}/20; "
          f"tokens before the first divergence {same20}/320")
    assert worst < 1.6 * float(err20.max()) + 0.05, "a pair is less accurate alone than in the batch of 20"
    del eng
    torch.cuda.empty_cache()


def test_32_layer_accuracy_ordering_of_the_modes(bench_engine):
    """fp16 operands beat bf16 operands, and keeping the residual stream in fp32 does not hurt: the mixed mode is the
    most accurate 16-bit mode at the benchmarked depth."""
    if len(_RESULTS) < len(MODES):
        pytest.skip("needs the per-mode results of the previous test")
    r = _RESULTS
    for m in r:
        print(m, r[m])
    assert r["fp16"][1] < 0.6 * r["bf16"][1] and r["mixed"][1] < 0.6 * r["bf16_res32"][1]
    assert r["mixed"][1] <= 1.1 * r["fp16"][1] and r["bf16_res32"][1] <= 1.1 * r["bf16"][1]
    assert r["mixed"][3] >= r["bf16"][3]


def test_decode_steps_are_batch_invariant_bit_for_bit(bench_engine):
    """The hand-written decode step (weight-streaming GEMM + row kernels) for a row of a 20-row batch == that row in a
    batch of 1 and of 4, bit for bit: same KV cache contents, same residual in, same logits out."""
    from openpsg_amd import ops
    be = bench_engine
    cfg, X = be["cfg"], be["X"]
    eng = _engine(be, "bf16")
    m = cfg.llm
    dev = X.device
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    x = torch.randn((20, m.hidden), device=dev, generator=g).to(torch.bfloat16)
    outs = {}
    for rows in (list(range(20)), [5], [2, 5, 9, 17]):
        ii = torch.tensor(rows, device=dev)
        xs = x[ii].contiguous()
        n = torch.empty_like(xs)
        ops.rmsnorm(xs.clone(), None, eng.layers[0]["ln1"], m.rms_eps, n)
        qkv = ops.skinny_gemm(n, eng.layers[0]["wqkv"]).reduce(torch.bfloat16)
        gu = ops.skinny_gemm(n, eng.layers[0]["wgu"])
        act = torch.empty((len(rows), m.inter), device=dev, dtype=torch.bfloat16)
        ops.silu_mul(gu, act)
        d = ops.skinny_gemm(act, eng.layers[0]["wdown"]).reduce(torch.bfloat16)
        lg = ops.skinny_gemm(n, eng.lm_head).reduce(torch.bfloat16)
        outs[len(rows)] = (rows, qkv, d, lg)
    torch.cuda.synchronize()
    rows20, q20, d20, l20 = outs[20]
    for nrows in (1, 4):
        rows, q, d, lg = outs[nrows]
        for r, i in enumerate(rows):
            assert torch.equal(q[r], q20[i]) and torch.equal(d[r], d20[i]) and torch.equal(lg[r], l20[i]), \
                f"row {i}: batch of {nrows} differs from batch of 20"


def test_fp32_decode_steps_are_batch_invariant_bit_for_bit(bench_engine):
    """The same for the fp32 kernel (psg_gemm_f32.hip) at the benchmarked width: a row of a 20-row batch (rows 0..15 on
    v_mfma_f32_16x16x1_4b, rows 16..19 on v_mfma_f32_4x4x1_16b) == that row in a batch of 1, 4, 16 or 17, bit for bit -
    whichever instruction and whichever position it gets.  One 7B-width layer of fp32 weights (0.8 GB)."""
    from openpsg_amd import ops
    be = bench_engine
    cfg = be["cfg"]
    m = cfg.llm
    dev = be["dev"]
    pre = "language_model.model.layers.0."
    w = {k: be["w"][pre + k].float() for k in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                                "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")}
    wqkv = torch.cat([w["self_attn.q_proj.weight"], w["self_attn.k_proj.weight"], w["self_attn.v_proj.weight"]], 0).contiguous()
    wgu = torch.cat([w["mlp.gate_proj.weight"], w["mlp.up_proj.weight"]], 0).contiguous()
    wdown = w["mlp.down_proj.weight"].contiguous()
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    x = torch.randn((20, m.hidden), device=dev, generator=g)
    ln = torch.ones(m.hidden, device=dev)
    outs = {}
    for rows in (list(range(20)), [17], [2, 5, 9, 17], list(range(16)), list(range(3, 20))):
        ii = torch.tensor(rows, device=dev)
        xs = x[ii].contiguous()
        n = torch.empty_like(xs)
        ops.rmsnorm(xs.clone(), None, ln, m.rms_eps, n)
        qkv = ops.skinny_gemm(n, wqkv).reduce(torch.float32)
        act = torch.empty((len(rows), m.inter), device=dev, dtype=torch.float32)
        ops.silu_mul(ops.skinny_gemm(n, wgu), act)
        d = ops.skinny_gemm(act, wdown).reduce(torch.float32)
        outs[tuple(rows)] = (qkv, d)
    torch.cuda.synchronize()
    q20, d20 = outs[tuple(range(20))]
    for rows, (q, d) in outs.items():
        for r, i in enumerate(rows):
            assert torch.equal(q[r], q20[i]) and torch.equal(d[r], d20[i]), f"row {i} in a batch of {len(rows)} differs"
