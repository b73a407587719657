"""psg_decode_layer (`-m gpu`): one persistent launch per decoder layer of the decode step against the chain of eight
launches it replaces (psg_rmsnorm / psg_skinny_gemm / psg_decode_attn / psg_silu_mul), BIT FOR BIT: residual stream, KV
cache rows, down-projection partials - at Llama-2-7B width (the only width it is built for), for every row-group variant
(13..24 rows), with and without an incoming delta, two layers chained, positions from 0 (no cached key) upwards, and
through the engine on the reference golden G6 (HF-LL:53-281 via V4:293-312)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
D, I, HEADS, CTX = 4096, 11008, 32, 64


def _layer(g):
    r = lambda *s, std=0.02: (torch.randn(*s, generator=g, device=DEV) * std)   # noqa: E731
    return dict(wqkv=r(3 * D, D), wo=r(D, D), wgu=r(2 * I, D), wdown=r(D, I, std=0.015),
                ln1=1.0 + 0.1 * torch.randn(D, generator=g, device=DEV), ln2=1.0 + 0.1 * torch.randn(D, generator=g, device=DEV))


def _chain(ops, L, resid, delta, pair, pos, rope, kc, vc):
    M = resid.shape[0]
    n = torch.empty_like(resid)
    ops.rmsnorm(resid, delta, L["ln1"], 1e-5, n)
    qkv = ops.skinny_gemm(n, L["wqkv"])
    att = torch.empty_like(resid)
    ops.decode_attn(qkv, pair, pos, rope, HEADS, 128, CTX, kc, vc, att)
    o = ops.skinny_gemm(att, L["wo"])
    ops.rmsnorm(resid, o, L["ln2"], 1e-5, n)
    gu = ops.skinny_gemm(n, L["wgu"])
    act = torch.empty((M, I), device=DEV)
    ops.silu_mul(gu, act)
    return ops.skinny_gemm(act, L["wdown"])


@pytest.mark.parametrize("M,with_delta", [(20, False), (20, True), (16, True), (13, False), (24, True), (17, False), (21, True)])
def test_decode_layer_equals_the_launch_chain_bit_for_bit(M, with_delta):
    from openpsg_amd import ops
    if not ops.decode_layer_supported(M, D, I, HEADS, torch.float32, DEV):
        pytest.skip("psg_decode_layer needs a 256-CU device")
    g = torch.Generator(device=DEV).manual_seed(100 * M + with_delta)
    layers = [_layer(g), _layer(g)]
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    ang = torch.arange(CTX, dtype=torch.float32)[:, None] * inv[None, :]
    rope = (ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV))
    pos = torch.randint(0, CTX - 1, (M,), generator=torch.Generator().manual_seed(M)).to(torch.int32)
    pos[0], pos[1] = 0, CTX - 1                                       # no cached key / a full context
    pos = pos.to(DEV)
    pair = torch.randperm(M, generator=torch.Generator().manual_seed(M + 1)).to(torch.int32).to(DEV)   # rows in any pair order
    resid0 = torch.randn(M, D, generator=g, device=DEV)
    delta0 = ops.Partials((torch.randn(16, M, D, generator=g, device=DEV) * 0.1).contiguous()) if with_delta else None
    caches = [(torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV), torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV))
              for _ in layers]
    # chain
    resid_c = resid0.clone()
    cc = [(k.clone(), v.clone()) for k, v in caches]
    delta = delta0
    for L, (k, v) in zip(layers, cc):
        delta = _chain(ops, L, resid_c, delta, pair, pos, rope, k, v)
    torch.cuda.synchronize()
    # persistent launches
    resid_p = resid0.clone()
    cp = [(k.clone(), v.clone()) for k, v in caches]
    ws, ncnt = ops.decode_layer_workspace(M, D, I, DEV)
    ws.fill_(float("nan"))
    counters = torch.zeros(len(layers) * ncnt, device=DEV, dtype=torch.int32)
    dparts = [torch.full((16, M, D), float("nan"), device=DEV) for _ in range(2)]
    dl = delta0
    for l, (L, (k, v)) in enumerate(zip(layers, cp)):
        dl = ops.decode_layer(resid_p, dl, L["ln1"], L["ln2"], L["wqkv"], L["wo"], L["wgu"], L["wdown"], pair, pos, rope,
                              HEADS, CTX, 1e-5, k, v, ws, counters[l * ncnt:(l + 1) * ncnt], dparts[l & 1])
    torch.cuda.synchronize()
    assert int(counters.view(len(layers), ncnt)[:, 255 * 64].abs().sum()) == 0, "a bounded poll of psg_decode_layer gave up"
    assert torch.isfinite(resid_p).all()
    for l, ((kc_, vc_), (kp, vp)) in enumerate(zip(cc, cp)):
        assert torch.equal(kc_, kp) and torch.equal(vc_, vp), f"layer {l}: KV cache differs from the chain"
    assert torch.equal(resid_c, resid_p), f"residual stream differs: max {(resid_c - resid_p).abs().max().item():.3e}"
    assert delta.t.shape == dl.t.shape and torch.equal(delta.t, dl.t), \
        f"down partials differ: max {(delta.t - dl.t).abs().max().item():.3e}"


@pytest.mark.parametrize("M,NL", [(20, 3), (16, 2), (23, 4)])
def test_decode_layers_chained_in_one_launch_equal_the_launch_chain_bit_for_bit(M, NL):
    """psg_decode_layers: NL decoder layers chained INSIDE one launch (the owners of layer l + 1 wait for the column groups
    of layer l's down projection; workspace buffers reused layer after layer) against the chain of 8 NL launches."""
    from openpsg_amd import ops
    if not ops.decode_layer_supported(M, D, I, HEADS, torch.float32, DEV):
        pytest.skip("psg_decode_layer needs a 256-CU device")
    g = torch.Generator(device=DEV).manual_seed(7 * M + NL)
    layers = [_layer(g) for _ in range(NL)]
    inv = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    ang = torch.arange(CTX, dtype=torch.float32)[:, None] * inv[None, :]
    rope = (ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV))
    pos = torch.randint(0, CTX - 1, (M,), generator=torch.Generator().manual_seed(M)).to(torch.int32).to(DEV)
    pair = torch.randperm(M, generator=torch.Generator().manual_seed(M + 1)).to(torch.int32).to(DEV)
    resid0 = torch.randn(M, D, generator=g, device=DEV)
    caches = [(torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV), torch.randn(M, HEADS, CTX, 128, generator=g, device=DEV))
              for _ in layers]
    resid_c = resid0.clone()
    cc = [(k.clone(), v.clone()) for k, v in caches]
    delta = None
    for L, (k, v) in zip(layers, cc):
        delta = _chain(ops, L, resid_c, delta, pair, pos, rope, k, v)
    torch.cuda.synchronize()
    resid_p = resid0.clone()
    cp = [(k.clone(), v.clone()) for k, v in caches]
    ws, ncnt = ops.decode_layer_workspace(M, D, I, DEV)
    ws.fill_(float("nan"))
    dparts = torch.full((2, 16, M, D), float("nan"), device=DEV)
    table = ops.decode_layer_table(layers, [k for k, _ in cp], [v for _, v in cp])
    for rep in range(2):                                            # the second run reuses workspace and (re-zeroed) counters
        resid_p.copy_(resid0)
        for (k, v), (k0, v0) in zip(cp, caches):
            k.copy_(k0)
            v.copy_(v0)
        counters = torch.zeros(NL * ncnt, device=DEV, dtype=torch.int32)
        dl = ops.decode_layers(resid_p, None, table, NL, pair, pos, rope, HEADS, CTX, 1e-5, I, ws, counters, dparts)
        torch.cuda.synchronize()
        assert int(counters.view(NL, ncnt)[:, 255 * 64].abs().sum()) == 0, "a bounded poll of psg_decode_layers gave up"
        for l, ((kc_, vc_), (kp, vp)) in enumerate(zip(cc, cp)):
            assert torch.equal(kc_, kp) and torch.equal(vc_, vp), f"layer {l}: KV cache differs from the chain"
        assert torch.equal(resid_c, resid_p) and torch.equal(delta.t, dl.t)


def test_engine_with_persistent_layers_decodes_the_reference_golden_and_equals_the_chain():
    """G6 (the LLM at the width the reference instantiates, 2 layers, 20 selected pairs) through the fp32 head with and
    without psg_decode_layer: identical tokens and first-step logits, both equal to the real reference's greedy tokens;
    graph replay included (the counters are zeroed by a memset node replayed first)."""
    from openpsg_amd import _lib, ops
    from openpsg_amd.head import RelationTransformerHeadV4
    from tests import helpers as H
    if not ops.decode_layer_supported(20, D, I, HEADS, torch.float32, DEV):
        pytest.skip("psg_decode_layer needs a 256-CU device")
    g, cfg, w, scene = H.load_case("G6_llm_7b_width_n6")
    dev = torch.device(DEV)
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    sel = torch.from_numpy(g["selected"].astype(np.int32)).to(dev)
    outs = {}
    for flag in (0, 1):
        _lib.set_option(0, "decode_persistent", flag)
        try:
            head = RelationTransformerHeadV4(dtype="fp32", device=DEV, qformer_vocab_size=cfg.qformer.vocab,
                                             llm_config=cfg.llm, llm_feature_size=cfg.llm.hidden, tokenizers="word",
                                             max_object_num=cfg.max_object_num, on_parse_error="skip",
                                             suppress_eos=bool(g["suppress_eos"]))
            head.load_weights(w)
            assert head.llm_engine.persistent_layer == bool(flag)
            rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names,
                                         scene["pan_results"].to(dev))
            runs = []
            for _ in range(3):                                      # capture, then two replays
                dec = head.decode_selected(rq, names, selected=sel)
                runs.append((dec["tokens_host"].copy(), dec["first_logits"].float().cpu()))
            torch.cuda.synchronize()
        finally:
            _lib.set_option(0, "decode_persistent", 0)
        for t, f in runs[1:]:
            assert np.array_equal(t, runs[0][0]) and torch.equal(f, runs[0][1])
        outs[flag] = runs[0]
        del head
        torch.cuda.empty_cache()
    assert np.array_equal(outs[0][0], outs[1][0]), "tokens of the persistent layers differ from the launch chain"
    assert torch.equal(outs[0][1], outs[1][1])
    for i in range(outs[1][0].shape[0]):
        want = g["gen_tokens"][i]
        want = want[want >= 0].tolist()
        assert [int(t) for t in outs[1][0][i] if t >= 0] == want
