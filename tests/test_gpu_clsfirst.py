"""Last Q-Former layer in two phases (qformer.forward_pairs_cls / pair_hidden): the cls row of every pair for the
existence head, then rows 1..32 of the selected pairs only.  Must be the same function as the one-phase layer."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_parity import _dev, _head, _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["G1_c1_512_n10", "G2_768x1024_n12", "G5_c5geo_1024x1344_n8"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
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
