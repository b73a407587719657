"""Randomised differential test (`-m gpu`): the fp32 head against the CPU oracle on scenes the goldens do not cover -
random padded geometries (patch counts that are not multiples of 32 or 64, non-square, resized originals), 1-11 objects,
void aliased with person#0 or not, vanishing objects (empty pair masks), class lists of random size (prompt lengths,
prompt de-duplication on / off by ratio).  Per scene: mask bits exact, existence logits 1e-3 on every pair, identical
top-K, greedy tokens of the first selected pairs identical; the mixed mode (the benchmarked one) on the same scenes stays
within 0.05 of the oracle's logits; the fp32s mode (split-fp16 products) meets the fp32 bar and decodes the fp32 head's
tokens."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

GEOS = [((512, 512), None, None), ((448, 576), (400, 520), (420, 546)), ((768, 1024), (720, 960), (750, 1000)),
        ((640, 832), None, None), ((1024, 1344), (480, 640), (1000, 1333)), ((384, 1152), None, None),
        ((832, 448), (790, 420), (820, 436))]


@pytest.fixture(scope="module")
def heads():
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_numpy
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 2, 512, 512), max_object_num=30)
    w = make_weights_numpy(cfg, seed=77)
    out = {}
    for dt in ("fp32", "fp32s", "mixed"):
        h = RelationTransformerHeadV4(dtype=dt, device="cuda:0", qformer_vocab_size=512, llm_config=cfg.llm,
                                      llm_feature_size=cfg.llm.hidden, tokenizers="word", max_object_num=30,
                                      on_parse_error="skip", suppress_eos=True)
        h.load_weights(w)
        out[dt] = h
    return cfg, w, out


def _seeds():
    """Ten scenes in the suite; PSG_FUZZ_SEEDS=lo:hi (a one-off sweep on the GPU box, `profiles/r06_fuzz_parity.txt`) adds
    scenes lo..hi-1, which draw their padded geometry at random as well (any multiple of 64 from 256 to 1408 per side,
    up to 30 objects)."""
    import os
    spec = os.environ.get("PSG_FUZZ_SEEDS")
    if not spec:
        return list(range(10))
    lo, hi = (int(v) for v in spec.split(":"))
    return list(range(lo, hi))


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene_against_the_oracle(heads, seed):
    from openpsg_amd.synthetic import make_scene
    from oracle import psg_oracle as O
    cfg, w, hs = heads
    rng = np.random.default_rng(1234 + seed)
    pad, ori, img = GEOS[seed % len(GEOS)]
    n = int(rng.integers(1, 12))
    if seed >= 10:                                              # sweep scenes: random geometry, more objects
        pad = (64 * int(rng.integers(4, 23)), 64 * int(rng.integers(4, 23)))
        ori = img = None
        if rng.random() < 0.5:                                  # a resized original inside the padded canvas
            img = (pad[0] - int(rng.integers(0, 63)), pad[1] - int(rng.integers(0, 63)))
            ori = (max(32, int(img[0] * rng.uniform(0.4, 1.0))), max(32, int(img[1] * rng.uniform(0.4, 1.0))))
        n = int(rng.integers(1, 31)) if rng.random() < 0.3 else n
    scene = make_scene(pad, n, seed=500 + seed, ori_hw=ori, img_hw=img, void_id=0 if seed % 2 else 133,
                       force_id0=bool(seed % 2), tiny_object=bool(seed % 3 == 0), num_categories=int(rng.integers(2, 134)))
    n = len(scene["object_id_list"])
    ids, tmask = H.qformer_prompts(scene)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        rq = O.relation_query(w, cfg, scene["mask_features"], scene["img_meta"], [int(i) for i in scene["object_id_list"]],
                              scene["pan_results"], ids, tmask)
    k = min(20, n * n)
    sel = O.select_topk(rq["exist_prob"], k)
    dev = torch.device("cuda:0")
    inputs = dict(mask_features=scene["mask_features"].to(dev), img_metas=[scene["img_meta"]],
                  object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"].to(dev))])
    head = hs["fp32"]
    out = head(inputs)
    torch.cuda.synchronize()
    last = head.last
    # object masks: exact
    L = rq["pair_masks"].shape[1]
    bits = last["bits"].cpu().numpy().view(np.uint64)
    om = np.unpackbits(bits.view(np.uint8), axis=-1, bitorder="little")[:, :L].astype(bool)
    want_om = rq["pair_masks"][torch.arange(n) * n + torch.arange(n)].numpy()
    assert np.array_equal(om, want_om)
    err = (last["exist_logit"].cpu() - rq["exist_logit"]).abs().max().item()
    assert err < 1e-3, err
    got_sel = last["selected"].cpu().tolist()
    p = rq["exist_prob"]
    assert len(got_sel) == k
    assert all(a == b or abs(float(p[a]) - float(p[b])) < 2e-6 for a, b in zip(got_sel, sel)), (got_sel, sel)
    empty = int((~rq["pair_masks"]).all(1).sum())
    checked = 0
    with torch.no_grad():
        pids, pmask = H.llm_prompts(scene, got_sel[:3])
        for i, si in enumerate(got_sel[:3]):
            x, mask = O.llm_inputs(w, rq["pair_feature"][si], pids[i], pmask[i])
            toks, _ = O.llm_generate(w, cfg, x, mask, suppress_eos=True)
            assert [int(t) for t in last["tokens_host"][i] if t >= 0] == toks, f"pair {si}"
            checked += 1
    assert set(out) == {"rel_pred", "rel_score"}
    # the benchmarked mode on the same scene
    hm = hs["mixed"]
    hm(inputs)
    torch.cuda.synchronize()
    em = (hm.last["exist_logit"].cpu() - rq["exist_logit"]).abs().max().item()
    print(f"seed {seed}: pad {pad}, N = {n}, L = {L}, empty pair masks {empty}; fp32 logits {err:.1e}; mixed logits {em:.1e}; "
          f"{checked} decodes token-exact")
    # the fp32 mode with split-fp16 products: the fp32 bar, and the fp32 head's own selection and tokens
    hs_ = hs["fp32s"]
    hs_(inputs)
    torch.cuda.synchronize()
    es = (hs_.last["exist_logit"].cpu() - rq["exist_logit"]).abs().max().item()
    assert es < 1e-3, es
    sel_s = hs_.last["selected"].cpu().tolist()
    assert all(a == b or abs(float(p[a]) - float(p[b])) < 2e-6 for a, b in zip(sel_s, sel)), (sel_s, sel)
    if sel_s == got_sel:
        assert np.array_equal(hs_.last["tokens_host"], last["tokens_host"]), "fp32s tokens differ from the fp32 head's"
    # the 16-bit mode is OUTSIDE the tolerance by design; its bound is empirical (0.016 typical, 0.035 the largest of the
    # 170 scenes of the round-6 sweep, profiles/r06_fuzz_parity.txt) and checked last so that it never hides the legs above
    assert em < 0.05, em
