"""Rows f2 / f4 of SURVEY 8 on the HIP path, pinned on captures of the reference's own lines
(oracle/capture_reference.py: F2_threshold_selector.npz, F4_pooling_scorer.npz) and on the oracle restatements
that tests/test_oracle_golden.py pins to the same captures.

f2: `head.select_pairs` in 'threshold' mode == V4:230-234 (commented out in the reference) as a SET: the pairs over
    the threshold united with the top-max_llm_forward_num; the build returns them ordered by score (ties: lower pair
    index first), capped at max_selected, optionally without the diagonal - extensions checked separately.
f4: `psg_masked_split_mean_pool` == `_mask_pooling` (openseed_relation.py:175-200), `psg_masked_mean_pool` == the
    masked-mean block (:453-468), `psg_bilinear_scores` == the v2 scorer (relation_transformer_head_v2.py:208-213).
"""
import numpy as np
import pytest
import torch

from oracle import psg_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def thr_head():
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_numpy
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 2, 512, 512), max_object_num=30)
    h = RelationTransformerHeadV4(dtype="fp32", device="cuda:0", qformer_vocab_size=512, llm_config=cfg.llm,
                                  llm_feature_size=cfg.llm.hidden, tokenizers="word", pair_selector="threshold",
                                  max_selected=4096)
    h.load_weights(make_weights_numpy(cfg, seed=2))
    return h


def test_threshold_selector_vs_reference_lines(thr_head):
    g = dict(np.load(H.GOLDEN + "/F2_threshold_selector.npz"))
    dev = _dev()
    for k in range(int(g["num_cases"])):
        prob = torch.from_numpy(g[f"c{k}_prob"])
        n = int(round(prob.numel() ** 0.5))
        thr_head.pair_selector_threshold = float(g[f"c{k}_threshold"])
        thr_head.max_llm_forward_num = int(g[f"c{k}_max_llm_forward_num"])
        got = thr_head.select_pairs(prob.to(dev), n).cpu().tolist()
        want = O.select_threshold(prob, thr_head.pair_selector_threshold, thr_head.max_llm_forward_num)
        if not str(g[f"c{k}_error"]):                              # the reference's own lines ran: their set, exactly
            assert want == g[f"c{k}_selected_sorted"].tolist()
        assert sorted(got) == want, str(g[f"c{k}_name"])
        # the build's order: by score, ties -> lower pair index first
        assert got == sorted(got, key=lambda i: (-float(prob[i]), i))


@pytest.mark.parametrize("n_hit", [0, 3, 40])
def test_threshold_selector_cap_ties_and_diagonal(thr_head, n_hit):
    """What the build adds to V4:230-234: the cap (`max_selected`, one decode batch) keeps the best by score of the
    reference's set; `exclude_diagonal` removes the i == j pairs first.  Scores with many exact ties, n_hit in
    {0, < max_llm_forward_num, > cap}."""
    dev = _dev()
    N, mlf, cap, thr = 12, 8, 16, 0.5
    gen = torch.Generator().manual_seed(n_hit)
    prob = (torch.randint(0, 5, (N * N,), generator=gen).float() / 10.0)          # 0 .. 0.4 in steps of 0.1: ties
    hit = torch.randperm(N * N, generator=gen)[:n_hit]
    prob[hit] = 0.6 + (torch.randint(0, 3, (n_hit,), generator=gen).float() / 10.0)
    thr_head.pair_selector_threshold, thr_head.max_llm_forward_num = thr, mlf
    old_cap, old_diag = thr_head.max_selected, thr_head.exclude_diagonal
    try:
        for diag in (False, True):
            thr_head.max_selected, thr_head.exclude_diagonal = cap, diag
            p = prob.clone()
            if diag:
                p[torch.arange(N) * (N + 1)] = -1.0
            full = O.select_threshold(p, thr, mlf)                                # the reference's set
            want = sorted(full, key=lambda i: (-float(p[i]), i))[:cap]            # best of it, the build's order
            got = thr_head.select_pairs(prob.to(dev), N).cpu().tolist()
            assert got == want, (n_hit, diag)
            if diag:
                assert not any(i // N == i % N for i in got)
    finally:
        thr_head.max_selected, thr_head.exclude_diagonal = old_cap, old_diag


def test_split_mean_pool_vs_reference_mask_pooling():
    """`_mask_pooling(None, feature, mask, k)` of the reference on its captured inputs: the kernel takes the id map and
    the object ids, so the six captured masks are painted into six disjoint id maps (one object each)."""
    from openpsg_amd import ops
    g = dict(np.load(H.GOLDEN + "/F4_pooling_scorer.npz"))
    dev = _dev()
    feature = torch.from_numpy(g["pool_feature"])
    masks = torch.from_numpy(g["pool_masks"])
    C, h, w = feature.shape
    feat = feature[None].to(dev).contiguous()
    for k in (1, 4, 7):
        for i in range(masks.shape[0]):
            pan = torch.where(masks[i, 0] >= 0.5, 7, 133).to(torch.int32).to(dev)    # id map at feature resolution
            got = ops.masked_split_mean_pool(feat, pan, (h, w), (h, w), torch.tensor([7], dtype=torch.int32, device=dev), k)
            np.testing.assert_allclose(got[0].cpu().numpy(), g[f"pool_k{k}"][i], atol=1e-5, err_msg=f"mask {i}, k={k}")
            want = O.mask_pooling(feature, masks[i], k)
            assert (got[0].cpu() - want).abs().max().item() < 1e-5


def test_masked_mean_pool_vs_reference_block():
    from openpsg_amd import ops
    g = dict(np.load(H.GOLDEN + "/F4_pooling_scorer.npz"))
    dev = _dev()
    ori, img, pad = g["mean_shapes"]
    feat = torch.from_numpy(g["mean_feature_map"]).to(dev)
    pan = torch.from_numpy(g["mean_pan"]).to(dev)
    ids = torch.from_numpy(g["mean_ids"]).to(dev)
    got = ops.masked_mean_pool(feat, pan, tuple(int(v) for v in img), tuple(int(v) for v in pad), ids)
    np.testing.assert_allclose(got.cpu().numpy(), g["mean_object_embedding"], atol=1e-5)
    want = O.masked_mean_objects(feat.cpu(), pan.cpu(), g["mean_ids"], img, pad)
    assert (got.cpu() - want).abs().max().item() < 1e-5


def test_bilinear_scores_vs_reference_scorer():
    from openpsg_amd import ops
    g = dict(np.load(H.GOLDEN + "/F4_pooling_scorer.npz"))
    dev = _dev()
    got = ops.bilinear_scores(torch.from_numpy(g["score_sub"]).to(dev), torch.from_numpy(g["score_obj"]).to(dev), int(g["score_R"]))
    np.testing.assert_allclose(got.cpu().numpy(), g["score_pred"], atol=1e-4)
    want = O.bilinear_scores(torch.from_numpy(g["score_sub"]), torch.from_numpy(g["score_obj"]), int(g["score_R"]))
    assert (got.cpu() - want).abs().max().item() < 1e-4
