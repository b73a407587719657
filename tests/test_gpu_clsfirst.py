"""Last Q-Former layer in two phases (qformer.forward_pairs_cls / pair_hidden): the cls row of every pair for the
existence head, then rows 1..32 of the selected pairs only.  Must be the same function as the one-phase layer."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_parity import _dev, _head, _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["G1_c1_512_n10", "G2_768x1024_n12", "G5_c5geo_1024x1344_n8"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16", "mixed_q32"])
def test_cls_first_equals_one_phase(case, dtype):
    g, cfg, w, scene = H.load_case(case)
    runs = {}
    for cf in (False, True):
        head = _head(cfg, w, dtype, suppress_eos=bool(g["suppress_eos"]), cls_first=cf)
        head(_inputs(scene))
        torch.cuda.synchronize()
        last = head.last
        assert ("pending" in last) == cf
        sel = last["selected"]
        runs[cf] = dict(logit=last["exist_logit"].float().cpu().numpy(), sel=sel.cpu().tolist(),
                        pf=head.selected_pair_features(last, sel).float().cpu().numpy(),
                        hidden=last["hidden"].float().cpu().numpy(), tokens=last["tokens_host"].copy())
    a, b = runs[False], runs[True]
    tol = 2e-5 if dtype == "fp32" else 0.06                       # 16-bit: other kernels / GEMM shapes round differently
    err = np.abs(a["logit"] - b["logit"]).max()
    print(f"{case} {dtype}: max |logit(one phase) - logit(cls first)| = {err:.3e}")
    assert err < tol
    if dtype == "fp32":
        assert a["sel"] == b["sel"]
        np.testing.assert_allclose(a["pf"], b["pf"], atol=2e-5)
        np.testing.assert_allclose(a["hidden"], b["hidden"], atol=2e-5)   # lazy `hidden` = last layer on every pair
        assert np.array_equal(a["tokens"], b["tokens"])
        assert np.abs(b["logit"] - g["exist_logit"]).max() < 1e-3 and b["sel"] == g["selected"].tolist()
    else:
        common = [s for s in a["sel"] if s in b["sel"]]
        assert len(common) >= len(a["sel"]) - 3                   # near-ties at the cut may swap
        ia = [a["sel"].index(s) for s in common]
        ib = [b["sel"].index(s) for s in common]
        pa = a["pf"].reshape(len(a["sel"]), 32, -1)[ia]
        pb = b["pf"].reshape(len(b["sel"]), 32, -1)[ib]
        assert np.abs(pa - pb).max() < 0.15


def test_cls_first_over_several_pair_chunks():
    """More pairs than one chunk holds (N = 100 at BASELINE C4): the selection phase runs per chunk, the selected pairs'
    rows come from whichever chunk owns them - same logits, selection, features and tokens as a single chunk."""
    g, cfg, w, scene = H.load_case("G2_768x1024_n12")               # 144 pairs
    runs = []
    for chunk in (4096, 37):
        head = _head(cfg, w, "fp32", suppress_eos=bool(g["suppress_eos"]), cls_first=True, pair_chunk=chunk)
        head(_inputs(scene))
        torch.cuda.synchronize()
        last = head.last
        assert len(last["pending"]) == (1 if chunk == 4096 else 4)
        runs.append(dict(logit=last["exist_logit"].cpu().numpy(), sel=last["selected"].cpu().tolist(),
                         pf=head.selected_pair_features(last).float().cpu().numpy(),
                         hidden=last["hidden"].float().cpu().numpy(), tokens=last["tokens_host"].copy()))
    a, b = runs
    np.testing.assert_allclose(a["logit"], b["logit"], atol=2e-5)
    assert a["sel"] == b["sel"] == g["selected"].tolist()
    np.testing.assert_allclose(a["pf"], b["pf"], atol=2e-5)
    np.testing.assert_allclose(a["hidden"], b["hidden"], atol=2e-5)
    assert np.array_equal(a["tokens"], b["tokens"])


@pytest.mark.parametrize("size,dtype", [((1280, 1536), "bf16"), ((1536, 1536), "bf16"), ((1280, 1536), "fp32")])
def test_cls_first_on_large_images(size, dtype):
    """Patch grids beyond the LDS-DMA cross-attention kernel (L = 480: first-generation MFMA kernel; L = 576: the row
    kernel): the selection phase calls the same kernels with one query row per pair and must agree with the one-phase
    layer there too."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    dev = _dev()
    cfg = PSGConfig(qformer=QFormerConfig(), llm=tiny_llm(256, 2, 512, 512), max_object_num=12)
    w = make_weights_device(cfg, 3, dev, llm_dtype=torch.float32 if dtype == "fp32" else torch.bfloat16)
    scene = make_scene(size, 12, seed=7, device="cuda:0")
    logits = {}
    for cf in (False, True):
        head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", tokenizers="word", max_object_num=12,
                                         llm_config=cfg.llm, llm_feature_size=256, on_parse_error="skip", cls_first=cf)
        head.load_weights(w)
        head(_inputs(scene))
        torch.cuda.synchronize()
        logits[cf] = head.last["exist_logit"].float().cpu().numpy()
        feats = head.selected_pair_features(head.last).float()
        assert torch.isfinite(feats).all()
    err = np.abs(logits[False] - logits[True]).max()
    print(f"{size} {dtype}: L = {size[0] // 64 * (size[1] // 64)}, max |logit diff| = {err:.3e}")
    assert err < (2e-5 if dtype == "fp32" else 0.06)


@pytest.mark.parametrize("dtype,T", [("fp32", 14), ("fp32", 0), ("bf16", 14), ("bf16", 31), ("fp16", 9)])
def test_cls_attention_in_input_space_matches_kv_form(dtype, T):
    """psg_qformer_cls_attn_input (queries projected back through W_k, weighted row means projected through W_v) against
    psg_qformer_self_attn_cls on the materialised K | V and against the fp64 formula (HF-IB:471-515, row 0 only)."""
    from openpsg_amd import ops
    dev = _dev()
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    P, nq, heads, Hd = 37, 33, 12, 768
    gen = torch.Generator(device="cpu").manual_seed(5)
    X = (torch.randn(P * (nq + T), Hd, generator=gen) * 0.7).to(dev, tdt)
    wq, wk, wv = [(torch.randn(Hd, Hd, generator=gen) * 0.06).to(dev, tdt) for _ in range(3)]
    bq, bk, bv = [(torch.randn(Hd, generator=gen) * 0.1).to(dev, tdt) for _ in range(3)]
    mask = (torch.rand(P, max(T, 1), generator=gen) < 0.7).to(torch.uint8)[:, :T].contiguous().to(dev)
    if T:
        mask[3] = 0                                                 # a pair whose text rows are all padding
    x_cls = X[:P * nq].view(P, nq, Hd)[:, 0].contiguous()
    q_cls = torch.nn.functional.linear(x_cls, wq, bq)
    kvs = torch.nn.functional.linear(X, torch.cat([wk, wv]), torch.cat([bk, bv]))
    ref_kv = ops.qformer_self_attn_cls(q_cls, kvs, mask, P, T, nq, heads).float().cpu()
    g = torch.bmm(q_cls.float().view(P, heads, 64).transpose(0, 1), wk.float().view(heads, 64, Hd))
    xbar = ops.qformer_cls_attn_input(X, g, mask, P, T, nq, heads)
    got = (torch.bmm(xbar, wv.float().view(heads, 64, Hd).transpose(1, 2)).permute(1, 0, 2).reshape(P, Hd)
           + bv.float()).to(tdt).float().cpu()
    # fp64 formula on the same (rounded) inputs
    Xd, qd = X.double().cpu(), q_cls.double().cpu()
    rows = torch.cat([Xd[:P * nq].view(P, nq, Hd), Xd[P * nq:].view(P, T, Hd)], dim=1)           # [P, S, Hd]
    K = rows @ wk.double().cpu().T + bk.double().cpu()
    V = rows @ wv.double().cpu().T + bv.double().cpu()
    valid = torch.cat([torch.ones(P, nq, dtype=torch.bool), mask.cpu().bool()], dim=1)
    sc = torch.einsum("phd,pshd->phs", qd.view(P, heads, 64), K.view(P, -1, heads, 64)) / 8.0
    sc = sc.masked_fill(~valid[:, None, :], float("-inf"))
    ref = torch.einsum("phs,pshd->phd", sc.softmax(-1), V.view(P, -1, heads, 64)).reshape(P, Hd).float()
    e_ref, e_kv = (got - ref).abs().max().item(), (ref_kv - ref).abs().max().item()
    print(f"{dtype} T={T}: input-space vs fp64 {e_ref:.3e}, K|V form vs fp64 {e_kv:.3e}")
    if dtype == "fp32":
        assert e_ref < 2e-5 and (got - ref_kv).abs().max().item() < 2e-5
    else:
        assert e_ref < max(1.2 * e_kv, 0.02)                        # no K / V rounding: at least as close


@pytest.mark.parametrize("dtype,chunk", [("fp32", 4096), ("fp32", 100), ("bf16", 4096), ("mixed_q32", 4096)])
def test_prompt_dedup_equals_per_pair_path(dtype, chunk, monkeypatch):
    """Scenes with repeated classes (many pairs share a prompt): the selection phase with the prompt-only work done
    once per distinct prompt (qformer._forward_pairs_cls_dedup) against the per-pair path - same logits, selection,
    selected pairs' features and tokens (fp32: 2e-5), also over several pair chunks."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    dev = _dev()
    N = 16
    cfg = PSGConfig(qformer=QFormerConfig(), llm=tiny_llm(256, 2, 512, 512), max_object_num=N)
    w = make_weights_device(cfg, 5, dev, llm_dtype=torch.float32 if dtype == "fp32" else torch.bfloat16)
    scene = make_scene((1024, 1024), N, seed=11, device="cuda:0", num_categories=4)
    runs = {}
    for dd in (False, True):
        head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", tokenizers="word", max_object_num=N,
                                         llm_config=cfg.llm, llm_feature_size=256, on_parse_error="skip",
                                         suppress_eos=True, cls_first=True, pair_chunk=chunk)
        head.load_weights(w)
        head.rq_engine.dedup_prompts = dd
        seen = []
        orig = head.rq_engine._forward_pairs_cls_dedup
        monkeypatch.setattr(head.rq_engine, "_forward_pairs_cls_dedup", lambda *a, **k: (seen.append(1), orig(*a, **k))[1])
        head(_inputs(scene))
        torch.cuda.synchronize()
        assert bool(seen) == dd                                     # the path under test really ran (or did not)
        last = head.last
        runs[dd] = dict(logit=last["exist_logit"].float().cpu().numpy(), sel=last["selected"].cpu().tolist(),
                        pf=head.selected_pair_features(last).float().cpu().numpy(),
                        hidden=last["hidden"].float().cpu().numpy(), tokens=last["tokens_host"].copy())
    a, b = runs[False], runs[True]
    err = np.abs(a["logit"] - b["logit"]).max()
    print(f"{dtype} chunk {chunk}: max |logit(per pair) - logit(per prompt)| = {err:.3e}")
    # pairs with the same prompt and the same mask union are exact ties in exact arithmetic (repeated classes make
    # them common): which of them makes the cut may differ, the selected SCORES may not
    la, lb = np.sort(a["logit"][a["sel"]]), np.sort(b["logit"][b["sel"]])
    common = [s for s in a["sel"] if s in b["sel"]]
    ia, ib = [a["sel"].index(s) for s in common], [b["sel"].index(s) for s in common]
    pa = a["pf"].reshape(len(a["sel"]), 32, -1)[ia]
    pb = b["pf"].reshape(len(b["sel"]), 32, -1)[ib]
    if dtype == "fp32":
        # (two fp32 paths whose library GEMMs see other row counts - per pair / per prompt, 33 shared query rows - and
        # whose LayerNorm chains amplify one rounding: both are ~1e-5 from the fp64 truth)
        assert err < 5e-5 and np.abs(la - lb).max() < 5e-5 and len(common) >= len(a["sel"]) - 4
        np.testing.assert_allclose(pa, pb, atol=5e-5)
        np.testing.assert_allclose(a["hidden"], b["hidden"], atol=5e-5)       # last layer of EVERY pair, both ways
        assert np.array_equal(a["tokens"][ia], b["tokens"][ib])
    else:
        assert err < 0.06 and len(common) >= len(a["sel"]) - 6
        assert np.abs(pa - pb).max() < 0.15


@pytest.mark.parametrize("N,cats", [(2, 1), (3, 1), (5, 2)])
def test_prompt_dedup_tiny_scenes(N, cats):
    """Two or three objects of ONE class (a single distinct prompt for all pairs) and five of two: the per-prompt path
    against the per-pair path in fp32."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    dev = _dev()
    cfg = PSGConfig(qformer=QFormerConfig(), llm=tiny_llm(256, 1, 256, 256), max_object_num=8)
    w = make_weights_device(cfg, 9, dev, llm_dtype=torch.float32)
    scene = make_scene((512, 512), N, seed=3, device="cuda:0", num_categories=cats)
    logits = {}
    for dd in (False, True):
        head = RelationTransformerHeadV4(dtype="fp32", device="cuda:0", tokenizers="word", max_object_num=8,
                                         llm_config=cfg.llm, llm_feature_size=256, on_parse_error="skip", cls_first=True)
        head.load_weights(w)
        head.rq_engine.dedup_prompts = dd
        out = head(_inputs(scene))
        torch.cuda.synchronize()
        assert set(out) >= {"rel_pred", "rel_score"}
        logits[dd] = head.last["exist_logit"].float().cpu().numpy()
        assert logits[dd].shape == (N * N,)
    assert np.abs(logits[False] - logits[True]).max() < 2e-5
