"""Weights stored as fp16 under fp32 arithmetic (`-m gpu`): the reference's LLM is the frozen Llama-2-7b-hf checkpoint -
fp16 on disk, upcast by `from_pretrained` (V4:99-100; configs/psg/baseline_v4_ov.py:61-65 freezes
`relation_head.language_model`) - so every LLM weight is an fp16 value.  psg_skinny_gemm_w16 streams such a weight as
2 bytes and widens it in the register: bit-identical to psg_skinny_gemm(PSG_F32) on the widened tensor."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fuzz_nk():
    """PSG_FUZZ_GEMM=count (a one-off sweep, profiles/r06_fuzz_gemm.txt): random (N, K) next to the fixed list - N any
    multiple of 16 up to 8192, K any multiple of 32 (64 for the two-plane product) up to 12288."""
    import os
    import random
    n = int(os.environ.get("PSG_FUZZ_GEMM", "0")) // 4
    r = random.Random(77)
    return tuple((16 * r.randint(1, 512), 64 * r.randint(1, 192)) for _ in range(n))


@pytest.mark.parametrize("M", [1, 4, 12, 13, 16, 20, 24, 29, 32])
def test_w16_stream_equals_the_fp32_weight_stream_bit_for_bit(M):
    from openpsg_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(50 + M)
    for N, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096), (272, 320), (16, 32), (1040, 96)) + _fuzz_nk():
        x = torch.randn(M, K, generator=g, device=DEV)
        w16 = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).half()
        try:
            a = ops.skinny_gemm(x, w16.float().contiguous())
        except _lib.PsgHipError as e:                          # M rows of this K beside the weight rings: refused at plan time
            assert "do not fit the LDS" in str(e) and M * K > 20 * 20480 * 0.9, (M, N, K, str(e))
            with pytest.raises(_lib.PsgHipError, match="do not fit the LDS"):
                ops.skinny_gemm_w16(x, w16)
            continue
        b = ops.skinny_gemm_w16(x, w16)
        assert a.t.shape == b.t.shape
        assert torch.equal(a.t, b.t), f"M={M} N={N} K={K}: max diff {(a.t - b.t).abs().max().item():.3e}"


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M", [1, 5, 16, 20, 32])
def test_split_gemm_w16_is_fp32_grade(M, mode):
    """psg_split_f16x2 + psg_split_gemm_w16 (fp32 rows = high + low fp16 part, weight an fp16 value, two products on the
    16-bit matrix cores) against the exact product: the error class of an fp32 GEMM (<= 2e-6 of |x|.|w|), rows of very
    different magnitudes included (per-row power-of-two scale); rows do not depend on their neighbours."""
    from openpsg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(90 + M)
    for N, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096), (272, 320), (16, 64)) + _fuzz_nk():
        x = torch.randn(M, K, generator=g, device=DEV) * torch.logspace(-6, 3, M, device=DEV)[:, None]
        w16 = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).half()
        x2, inv = ops.split_f16x2(x)
        assert torch.equal((x2[0].double() + x2[1].double()) * inv.double()[:, None], x.double()) or \
            ((x2[0].double() + x2[1].double()) * inv.double()[:, None] - x.double()).abs().max() <= 2.0 ** -21 * x.abs().max(1)[0].max()
        part = ops.split_gemm_w16(x2, inv, w16, mode)
        assert part.t.shape[1:] == (M, N) and part.splits <= 16
        got = part.t.sum(0).double()
        ref = x.double() @ w16.double().t()
        bound = (x.abs().double() @ w16.abs().double().t())
        assert ((got - ref).abs() <= 2e-6 * bound + 1e-30).all(), \
            f"M={M} N={N} K={K}: worst {((got - ref).abs() / bound.clamp_min(1e-30)).max().item():.3e}"
        if M >= 5:
            x_ = x.clone()
            x_[2:] = torch.randn(M - 2, K, generator=g, device=DEV)
            x2b, invb = ops.split_f16x2(x_)
            assert torch.equal(ops.split_gemm_w16(x2b, invb, w16, mode).t[:, :2], part.t[:, :2])


def _g6_head(dtype, w, cfg, g, flag):
    from openpsg_amd import _lib
    from openpsg_amd.head import RelationTransformerHeadV4
    _lib.set_option(0, "llm_w16", flag)
    try:
        head = RelationTransformerHeadV4(dtype=dtype, device=DEV, qformer_vocab_size=cfg.qformer.vocab, llm_config=cfg.llm,
                                         llm_feature_size=cfg.llm.hidden, tokenizers="word", max_object_num=cfg.max_object_num,
                                         on_parse_error="skip", suppress_eos=bool(g["suppress_eos"]))
        head.load_weights(w)
    finally:
        _lib.set_option(0, "llm_w16", 1)
    return head


@pytest.mark.parametrize("dtype", ["fp32", "fp32s"])
def test_engine_streams_fp16_valued_weights_as_fp16(dtype):
    """G6 (Llama-2-7B width, 2 layers, 20 selected pairs) with the LLM's matrices rounded through fp16 - the values a frozen
    fp16 checkpoint has after `from_pretrained` upcast it (V4:99-100).  The engine finds that every projection round-trips
    and streams fp16 copies in the decode steps: exact mode bit-identical to the fp32 stream (tokens AND first-step
    logits), fp32s mode token-identical with logits within 1e-4; weights that are NOT fp16 values keep the fp32 stream."""
    import numpy as np
    from tests import helpers as H
    g, cfg, w, scene = H.load_case("G6_llm_7b_width_n6")
    w16 = {k: (torch.as_tensor(v).half().float() if k.startswith("language_model.") and torch.as_tensor(v).dim() >= 2 else v)
           for k, v in w.items()}
    dev = torch.device(DEV)
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    sel = torch.from_numpy(g["selected"].astype(np.int32)).to(dev)
    outs = {}
    for flag in (0, 1):
        head = _g6_head(dtype, w16, cfg, g, flag)
        assert bool(head.llm_engine._w16) == bool(flag)
        rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names, scene["pan_results"].to(dev))
        dec = head.decode_selected(rq, names, selected=sel)
        dec2 = head.decode_selected(rq, names, selected=sel)                 # graph replay
        assert np.array_equal(dec["tokens_host"], dec2["tokens_host"])
        outs[flag] = (dec["tokens_host"].copy(), dec["first_logits"].float().cpu())
        del head
        torch.cuda.empty_cache()
    assert np.array_equal(outs[0][0], outs[1][0]), "tokens differ between the fp32 and the fp16 weight stream"
    if dtype == "fp32":
        assert torch.equal(outs[0][1], outs[1][1])
    else:
        assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-4
    head = _g6_head(dtype, w, cfg, g, 1)                                       # generic fp32 weights: nothing to narrow
    assert not head.llm_engine._w16
    # one matrix that is NOT an fp16 value (a fine-tuned lm_head): it alone keeps the fp32 stream, tokens unchanged
    del head
    torch.cuda.empty_cache()
    wm = dict(w16)
    wm["language_model.lm_head.weight"] = torch.as_tensor(w["language_model.lm_head.weight"]).clone()
    outs_m = {}
    for flag in (0, 1):
        head = _g6_head(dtype, wm, cfg, g, flag)
        eng = head.llm_engine
        assert (not eng._w16_all) and (len(eng._w16) == 4 * len(eng.layers) if flag else not eng._w16)
        rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names, scene["pan_results"].to(dev))
        outs_m[flag] = head.decode_selected(rq, names, selected=sel)["tokens_host"].copy()
        del head
        torch.cuda.empty_cache()
    assert np.array_equal(outs_m[0], outs_m[1])


def test_two_plane_rmsnorm_equals_the_row_kernel_followed_by_the_split_bit_for_bit():
    """psg_rmsnorm_split2 (decode steps: fp32 rows, fp32 split-K slices) against psg_rmsnorm + psg_split_f16x2."""
    from openpsg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    for rows, D, I in ((20, 4096, 11008), (32, 4096, 11008), (3, 512, 1024), (7, 1024, 2816)):
        w = 1.0 + 0.1 * torch.randn(D, generator=g, device=DEV)
        for S in (0, 1, 4, 16):
            resid = torch.randn(rows, D, generator=g, device=DEV) * 3
            delta = ops.Partials(torch.randn(S, rows, D, generator=g, device=DEV).contiguous()) if S else None
            ra, rb = resid.clone(), resid.clone()
            n = torch.empty_like(ra)
            ops.rmsnorm(ra, delta, w, 1e-5, n)
            a2, ainv = ops.split_f16x2(n)
            b2, binv = ops.rmsnorm_split2(rb, delta, w, 1e-5)
            assert torch.equal(ra, rb) and torch.equal(a2, b2) and torch.equal(ainv, binv), (rows, D, S)


def test_plans_never_exceed_the_slices_a_consumer_can_sum():
    """Other widths than Llama-2-7B's (13B: 5120 / 13824; 70B-like: 8192 / 28672; narrow N) must still plan <= 16 slices."""
    from openpsg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    for N, K in ((5120, 5120), (5120, 13824), (8192, 8192), (1024, 28672), (4096, 16384), (256, 8192)):
        x = torch.randn(20, K, generator=g, device=DEV)
        w16 = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).half()
        x2, inv = ops.split_f16x2(x)
        for mode in (0, 1, 2):
            try:
                part = ops.split_gemm_w16(x2, inv, w16, mode)
            except Exception as exc:                                   # a forced mode may have no plan; the estimate must
                assert mode != 0, exc
                continue
            assert part.splits <= 16
            ref = x.double() @ w16.double().t()
            bound = x.abs().double() @ w16.abs().double().t()
            assert ((part.t.sum(0).double() - ref).abs() <= 2e-6 * bound + 1e-30).all()
        xb = torch.randn(96, K, generator=g, device=DEV).half()
        for bn in (0, 128, 256):
            for mode in (0, 1, 2):
                try:
                    pb = ops.batch_gemm(xb, w16, bn, mode)
                except Exception as exc:
                    assert mode != 0 or bn != 0, exc
                    continue
                assert pb.splits <= 16


def test_small_llm_with_fp16_valued_weights_decodes_the_same_tokens_with_and_without_the_fp16_stream():
    """A small LLM shape (hidden 512, inter 1024, vocab 512, 2 layers: other kernel instantiations than the 7B width) through
    the whole head in fp32s: llm_w16 = 1 (fp16 stream: psg_split_gemm_w16 decode steps, two-plane prompt pass) against
    llm_w16 = 0 (fp32 stream, three-segment prompt pass) on the same fp16-valued weights - identical tokens, existence
    logits identical (the Q-Former does not change), first-step logits within 1e-4."""
    import numpy as np
    from openpsg_amd import _lib
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    dev = torch.device(DEV)
    cfg = PSGConfig(qformer=QFormerConfig(vocab=30522), llm=tiny_llm(512, 2, 1024, 512), max_object_num=50)
    w = make_weights_device(cfg, 7, dev, llm_dtype=torch.float32, llm_values=torch.float16)
    sc = make_scene((512, 768), 12, seed=4, device=DEV)
    inp = dict(mask_features=sc["mask_features"], img_metas=[sc["img_meta"]],
               object_info=[dict(object_id_list=sc["object_id_list"], pan_results=sc["pan_results"])])
    res = {}
    for flag in (0, 1):
        _lib.set_option(0, "llm_w16", flag)
        try:
            head = RelationTransformerHeadV4(dtype="fp32s", device=DEV, llm_config=cfg.llm, llm_feature_size=512,
                                             tokenizers="word", max_object_num=50, on_parse_error="skip", suppress_eos=True)
            head.load_weights(w)
        finally:
            _lib.set_option(0, "llm_w16", 1)
        assert head.llm_engine._w16_all == bool(flag)
        out = head(inp)
        out2 = head(inp)                                               # graph replay
        assert out == out2
        res[flag] = (head.last["tokens_host"].copy(), head.last["exist_logit"].float().cpu().clone(), out)
        del head
        torch.cuda.empty_cache()
    assert np.array_equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
