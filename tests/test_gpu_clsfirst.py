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
