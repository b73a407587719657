"""The pair-sharded paths on ONE GPU (`-m gpu`): `LoopbackWorld` drives R in {2, 4, 8} `HipBackend` ranks - all sharing one
head - through `step_one_image` (BASELINE C4: 100 masks, 10 000 pairs, decodes dealt round-robin) and `step` (R images,
unequal object counts and sizes, fixed-size and threshold selection), and the results must be those of the
single-GPU head (SURVEY 4 "fake world"; SURVEY 8e "bit-exactness across R").

What can be bit-exact and what cannot: the hand-written kernels compute every pair independently of its neighbours,
but the dense projections go through the library GEMM, whose tile / kernel choice depends on the row count - a shard
has fewer rows than the whole image.  So in the fp32 verification mode probabilities agree to 2e-5 (selection and
every token identical), and in the 16-bit modes they agree to the 16-bit rounding noise, bounded here.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk_head(dtype, max_obj, **kw):
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_device
    cfg = PSGConfig(qformer=QFormerConfig(vocab=30522), llm=tiny_llm(512, 2, 1024, 512), max_object_num=max_obj)
    w = make_weights_device(cfg, 7, torch.device("cuda:0"), llm_dtype=torch.float32)
    head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", llm_config=cfg.llm, llm_feature_size=512,
                                     tokenizers="word", max_object_num=max_obj, on_parse_error="skip", suppress_eos=True,
                                     **kw)
    head.load_weights(w)
    return head


def _same_selection(got, want, prob, tol=5e-5):
    """Selections agree up to near-ties: where the two lists differ, the probabilities of the two pairs do not (pairs
    of objects with the same class and the same patch mask are exact ties inside one GEMM call and 1e-6 apart across
    two)."""
    if got.numel() != want.numel():
        return False
    g, w = got.long(), want.long()
    return bool(((g == w) | ((prob[g] - prob[w]).abs() < tol)).all())


def _inputs(scene):
    return dict(mask_features=scene["mask_features"], img_metas=[scene["img_meta"]],
                object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"])])


@pytest.fixture(scope="module")
def c4():
    from openpsg_amd.synthetic import make_scene
    scene = make_scene((1024, 1024), 100, seed=4, device="cuda:0", tiny_object=True)
    heads = {}
    for dt in ("fp32", "mixed", "fp32s"):
        h = _mk_head(dt, 100)
        if dt == "fp32s":                       # the single-GPU reference of the bit-exact test: row-count-invariant
            h.llm_engine.row_invariant = True   # projections in the prompt pass too, as the dealt decodes use them
        h(_inputs(scene))
        torch.cuda.synchronize()
        heads[dt] = (h, dict(prob=h.last["exist_prob"].clone(), sel=h.last["selected"].clone(),
                             tokens=h.last["tokens_host"].copy()))
    return scene, heads


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_c4_image_sharded_over_fake_ranks_fp32(c4, world):
    """BASELINE C4 (100 masks, 10 000 pairs) through step_one_image: the image's constants (object ids, bitmasks, patches)
    broadcast by rank 0 - ranks != 0 get NO scene -, pair shards, identical top-20, features all-reduced, the 20 decodes
    dealt round-robin (K = 10 / 5 / 3-or-2 per rank), tokens re-assembled."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    scene, heads = c4
    head, ref = heads["fp32"]
    fw = LoopbackWorld(world)
    pipes = fw.pipelines(HipBackend(head))
    # only rank 0 holds the image; every other rank works from its broadcast (object ids, bitmasks, patches)
    outs = fw.run([p.step_one_image_gen(scene if r == 0 else None) for r, p in enumerate(pipes)])
    torch.cuda.synchronize()
    d = (outs[0]["exist_prob"] - ref["prob"]).abs().max().item()
    print(f"world {world}: max |prob - single GPU| = {d:.2e}")
    assert d < 2e-5
    for r in range(world):
        assert torch.equal(outs[r]["exist_prob"], outs[0]["exist_prob"])
        assert torch.equal(outs[r]["selected"], outs[0]["selected"])
        assert _same_selection(outs[r]["selected"], ref["sel"], ref["prob"])
        same = (outs[r]["selected"] == ref["sel"]).cpu().numpy()
        assert same.sum() >= 18
        assert np.array_equal(outs[r]["tokens"].cpu().numpy()[same], ref["tokens"][same])


@pytest.mark.parametrize("world", [2, 8])
def test_one_c4_image_sharded_is_bit_exact_with_row_invariant_projections(c4, world):
    """SURVEY 8e "bit-exactness across R": in the fp32s mode every Q-Former projection runs on psg_dense_gemm (one
    k-ordered accumulation per output element, whatever the row count of the call), every other kernel computes a pair
    independently of its neighbours - so the all-gathered probabilities of a sharded pass EQUAL the single-GPU head's,
    bit for bit, the top-20 is the same list, and - with the prompt pass on the same kernel - so is every decoded token."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    scene, heads = c4
    head, ref = heads["fp32s"]
    # (round 6: the split mode runs the input-space selection phase and the per-prompt de-duplication too - on the repo's own
    # row-count-invariant products, `qformer._cls_keys_split` - and a shard may or may not de-duplicate: still bit-exact)
    assert head.rq_engine.split and head.rq_engine.cls_input_space
    fw = LoopbackWorld(world)
    outs = fw.run([p.step_one_image_gen(scene if r == 0 else None) for r, p in enumerate(fw.pipelines(HipBackend(head)))])
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(outs[r]["exist_prob"], ref["prob"]), f"rank {r}: probabilities differ from the single-GPU head"
        assert torch.equal(outs[r]["selected"], ref["sel"])
    same = int((outs[0]["tokens"].cpu().numpy() == ref["tokens"]).all(axis=1).sum())
    print(f"fp32s, world {world}: probabilities bit-exact, selection 20/20, identical token sequences {same}/20")
    # the dealt prompt passes (10 / 3-or-2 pairs per rank) run on the row-count-invariant projections (psg_dense_gemm,
    # HipBackend.decode_dealt), the decode steps on kernels that are batch-invariant by construction: EVERY token
    assert same == 20
    for r in range(world):
        assert np.array_equal(outs[r]["tokens"].cpu().numpy(), ref["tokens"])


def test_one_c4_image_sharded_mixed_mode_is_within_rounding(c4):
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    scene, heads = c4
    head, ref = heads["mixed"]
    fw = LoopbackWorld(8)
    outs = fw.run([p.step_one_image_gen(scene) for p in fw.pipelines(HipBackend(head))])
    torch.cuda.synchronize()
    d = (outs[0]["exist_prob"] - ref["prob"]).abs().max().item()
    overlap = len(set(outs[0]["selected"].tolist()) & set(ref["sel"].tolist()))
    same = int((outs[0]["tokens"].cpu().numpy() == ref["tokens"]).all(axis=1).sum()) if overlap == 20 and torch.equal(
        outs[0]["selected"], ref["sel"]) else -1
    print(f"mixed, world 8: max |prob - single GPU| = {d:.2e}; top-20 overlap {overlap}/20; identical sequences {same}/20")
    assert d < 5e-3 and overlap >= 19
    for r in range(1, 8):
        assert torch.equal(outs[r]["selected"], outs[0]["selected"]) and torch.equal(outs[r]["tokens"], outs[0]["tokens"])


@pytest.mark.parametrize("world,selector,per_rank", [(4, "topk", 1), (4, "threshold", 1), (2, "topk", 1), (8, "topk", 1),
                                                     (2, "topk", 2), (4, "threshold", 2), (2, "topk", 3)])
def test_step_with_unequal_images_over_fake_ranks(world, selector, per_rank):
    """`step`: P * R images per step, every image's pairs sharded over all R ranks, image m decoded by rank m % R (with
    P > 1 a rank's decodes run side by side on the head's slot streams: bench.py's multi-GPU step).  The images differ
    in object count (different shard lengths, different K under the threshold selector) and in size (different patch
    counts)."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    from openpsg_amd.synthetic import make_scene
    kw = dict(pair_selector="threshold", exclude_diagonal=True, max_selected=24) if selector == "threshold" else {}
    head = _mk_head("fp32", 50, **kw)
    geo = ([((1024, 1024), 50), ((768, 1024), 23), ((512, 512), 9), ((1024, 1344), 31)] +
           [((1024, 1024), 50)] * 4)[:world * per_rank]            # 8 images: bench.py's weak-scaling step (N = 50 images) too
    geo = (geo * per_rank)[:world * per_rank]
    scenes = [make_scene(hw, n, seed=60 + m, device="cuda:0", tiny_object=True) for m, (hw, n) in enumerate(geo)]
    if selector == "threshold":
        # a threshold in the middle of each image's scores gives a data-dependent K; use one between the 7th and 8th
        # best of image 0 so that at least one image takes fewer than max_selected pairs
        head(_inputs(scenes[0]))
        p = torch.sort(head.last["exist_prob"], descending=True).values
        head.pair_selector_threshold = 0.5 * float(p[8] + p[9])
    refs = []
    for s in scenes:
        head(_inputs(s))
        torch.cuda.synchronize()
        refs.append(dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(),
                         tokens=head.last["tokens_host"].copy()))
    fw = LoopbackWorld(world)
    outs = fw.run([p.step_gen(scenes) for p in fw.pipelines(HipBackend(head))])
    torch.cuda.synchronize()
    ks = []
    for m in range(world * per_rank):
        d = (outs[0]["exist_prob"][m] - refs[m]["prob"]).abs().max().item()
        assert d < 2e-5, (m, d)
        ks.append(refs[m]["sel"].numel())
        for r in range(world):
            assert torch.equal(outs[r]["selected"][m], outs[0]["selected"][m])     # every rank made the same selection
            assert _same_selection(outs[r]["selected"][m], refs[m]["sel"], refs[m]["prob"]), (r, m)
            same = (outs[r]["selected"][m] == refs[m]["sel"]).cpu().numpy()
            assert np.array_equal(outs[r]["tokens"][m].cpu().numpy()[same], refs[m]["tokens"][same]), (r, m)
    print(f"world {world} x {per_rank} images per rank, {selector}: K per image {ks}")
    if selector == "threshold":
        assert len(set(ks)) > 1, "the threshold case should exercise different K per image"


@pytest.mark.parametrize("world,per_rank", [(4, 1), (2, 2)])
def test_step_in_the_headline_mode_is_bit_exact_with_the_single_gpu_head(world, per_rank):
    """Weak-scaling `step` in fp32s (bench.py's N > 1 job): the Q-Former's projections are the repo's own row-count-invariant
    GEMM and - round 6 - the library products of the prompt pass run in a form that is a pure function of the shape
    (llm._SPLIT_PLAN_TABLE), so an image's probabilities, selection and EVERY decoded token equal the single-GPU head's,
    bit for bit, on whichever rank it is decoded (SURVEY 8e "bit-exactness across R"; round 5 could only assert this for
    `step_one_image`)."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    from openpsg_amd.synthetic import make_scene
    head = _mk_head("fp32s", 50)
    geo = [((1024, 1024), 50), ((768, 1024), 23), ((512, 512), 9), ((1024, 1344), 31)][:world * per_rank]
    scenes = [make_scene(hw, n, seed=80 + m, device="cuda:0", tiny_object=True) for m, (hw, n) in enumerate(geo)]
    refs = []
    for s in scenes:
        head(_inputs(s))
        torch.cuda.synchronize()
        refs.append(dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(),
                         tokens=head.last["tokens_host"].copy()))
    fw = LoopbackWorld(world)
    outs = fw.run([p.step_gen(scenes) for p in fw.pipelines(HipBackend(head))])
    torch.cuda.synchronize()
    for m in range(len(scenes)):
        for r in range(world):
            assert torch.equal(outs[r]["exist_prob"][m], refs[m]["prob"]), (r, m)
            assert torch.equal(outs[r]["selected"][m], refs[m]["sel"]), (r, m)
            assert np.array_equal(outs[r]["tokens"][m].cpu().numpy(), refs[m]["tokens"]), (r, m)


def test_infer_tool_image_dealing_through_the_fake_world(tmp_path):
    """BASELINE C5's multi-GPU form (`torch.distributed.run ... tools/infer.py`): whole images dealt round-robin to the
    ranks, results merged in image order.  8 images at the C5 geometry (480x640 -> 1000x1333 -> pad 1024x1344), fp16,
    threshold selection, over 1 / 2 / 4 / 8 fake ranks: the same relation.json and panoptic PNGs as one rank."""
    import importlib.util
    import json
    import os
    spec = importlib.util.spec_from_file_location("psg_infer_tool", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "infer.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    head = _mk_head("fp16", 30, pair_selector="threshold", exclude_diagonal=True, max_selected=24)
    outs = {}
    for world in (1, 2, 4, 8):
        a = tool.parser().parse_args(["--segmenter", "synthetic", "--images", "8", "--objects", "8", "--ori-size", "480",
                                      "640", "--selector", "threshold", "--out", str(tmp_path / f"w{world}")])
        a.size_given = False
        results, path = tool.run_fake_world(a, head, world)
        assert len(results) == 8
        outs[world] = (results, json.load(open(path)))
        assert results[0]["pan_results"].shape == (480, 640)
    for world in (2, 4, 8):
        for a_, b_ in zip(outs[1][0], outs[world][0]):
            assert a_["rel_results"] == b_["rel_results"] and np.array_equal(a_["pan_results"], b_["pan_results"])
        ja, jb = outs[1][1], outs[world][1]
        assert [x["relations"] for x in ja] == [x["relations"] for x in jb]


def test_image_constants_message_round_trip():
    """The one-message broadcast of SURVEY 8e: object ids, bitmask words and patches survive the int32 packing bit for
    bit, and a rank that received them alone reproduces rank 0's pair shard exactly."""
    from openpsg_amd.dist import HipBackend, PairShardedPipeline
    from openpsg_amd.synthetic import make_scene
    head = _mk_head("fp32", 30)
    scene = make_scene((768, 1024), 13, seed=3, ori_hw=(720, 960), img_hw=(750, 1000), device="cuda:0", tiny_object=True)
    be = HipBackend(head)
    ids, bits, patches = be.image_constants(scene)
    msg = PairShardedPipeline.pack_constants(ids, bits, patches)
    assert msg.dtype == torch.int32 and msg.numel() == 4 + ids.numel() + 2 * bits.numel() + patches.numel()
    ids2, bits2, patches2 = PairShardedPipeline.unpack_constants(msg.clone())
    assert torch.equal(ids2, ids) and torch.equal(bits2, bits) and torch.equal(patches2, patches)
    here = be.query_shard(scene, patches, 40, 120)
    there = be.query_shard(be.scene_from_ids(ids2), patches2, 40, 120, bits2)
    assert torch.equal(here[1], there[1])


def _shard_fuzz_seeds():
    import os
    spec = os.environ.get("PSG_FUZZ_SHARD_SEEDS")               # lo:hi - a one-off sweep (profiles/r06_fuzz_shard.txt)
    if not spec:
        return [0, 1, 2, 3]
    lo, hi = (int(v) for v in spec.split(":"))
    return list(range(lo, hi))


@pytest.fixture(scope="module")
def fp32s_head():
    h = _mk_head("fp32s", 40)
    h.llm_engine.row_invariant = True
    return h


@pytest.mark.parametrize("seed", _shard_fuzz_seeds())
def test_random_image_sharded_over_a_random_world_is_bit_exact_in_the_headline_mode(fp32s_head, seed):
    """SURVEY 8e "bit-exactness across R" away from the benchmark's shape: 1-40 objects (1 pair ... 1600 pairs), random
    geometry, world sizes that do not divide the pair count and worlds LARGER than the pair count (empty shards, ranks
    that are dealt no decode), selections shorter than 20 - the sharded job gives the single-GPU head's probabilities,
    selection and every token, on every rank."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    from openpsg_amd.synthetic import make_scene
    rng = np.random.default_rng(4242 + seed)
    head = fp32s_head
    n = [1, 2, 3, 40][seed] if seed < 4 else int(rng.integers(1, 41))
    world = [8, 3, 5, 7][seed] if seed < 4 else int(rng.integers(2, 9))
    pad = (64 * int(rng.integers(4, 19)), 64 * int(rng.integers(4, 19)))
    scene = make_scene(pad, n, seed=900 + seed, device="cuda:0", tiny_object=bool(seed % 3 == 0),
                       num_categories=int(rng.integers(2, 134)))
    n = len(scene["object_id_list"])
    head(_inputs(scene))
    torch.cuda.synchronize()
    ref = dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(), tokens=head.last["tokens_host"].copy())
    fw = LoopbackWorld(world)
    outs = fw.run([p.step_one_image_gen(scene if r == 0 else None) for r, p in enumerate(fw.pipelines(HipBackend(head)))])
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(outs[r]["exist_prob"], ref["prob"]), f"rank {r}: probabilities differ from the single-GPU head"
        assert torch.equal(outs[r]["selected"], ref["sel"]), f"rank {r}: selection"
        assert np.array_equal(outs[r]["tokens"].cpu().numpy(), ref["tokens"]), f"rank {r}: tokens"
    print(f"shard seed {seed}: pad {pad}, N = {n} ({n * n} pairs), world {world}, selected {ref['sel'].numel()}: bit-exact on every rank")


@pytest.mark.parametrize("seed", _shard_fuzz_seeds())
def test_random_step_of_unequal_images_is_bit_exact_in_the_headline_mode(fp32s_head, seed):
    """The weak-scaling `step` (bench.py's N > 1 job) on random jobs: 2-8 ranks, one or two images per rank, every image
    with its own geometry and 1-40 objects (images of ONE pair next to images of 1600, shards that are empty on most
    ranks, selections shorter than 20): every image's probabilities, selection and tokens equal the single-GPU head's
    on every rank."""
    from openpsg_amd.dist import HipBackend, LoopbackWorld
    from openpsg_amd.synthetic import make_scene
    rng = np.random.default_rng(777 + seed)
    head = fp32s_head
    world = [2, 3, 8, 5][seed] if seed < 4 else int(rng.integers(2, 9))
    per_rank = 2 if (seed == 0 or (seed >= 4 and rng.random() < 0.25 and world <= 4)) else 1
    scenes = []
    for m in range(world * per_rank):
        n = 1 if (m == 1 and seed % 2 == 0) else int(rng.integers(1, 41))
        pad = (64 * int(rng.integers(4, 19)), 64 * int(rng.integers(4, 19)))
        scenes.append(make_scene(pad, n, seed=3000 + 50 * seed + m, device="cuda:0", num_categories=int(rng.integers(2, 134))))
    refs = []
    for s in scenes:
        head(_inputs(s))
        torch.cuda.synchronize()
        refs.append(dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(),
                         tokens=head.last["tokens_host"].copy()))
    fw = LoopbackWorld(world)
    outs = fw.run([p.step_gen(scenes) for p in fw.pipelines(HipBackend(head))])
    torch.cuda.synchronize()
    for m in range(len(scenes)):
        for r in range(world):
            assert torch.equal(outs[r]["exist_prob"][m], refs[m]["prob"]), (r, m)
            assert torch.equal(outs[r]["selected"][m], refs[m]["sel"]), (r, m)
            assert np.array_equal(outs[r]["tokens"][m].cpu().numpy(), refs[m]["tokens"]), (r, m)
    print(f"step seed {seed}: world {world} x {per_rank} image(s) per rank, objects {[len(s['object_id_list']) for s in scenes]}: "
          "bit-exact on every rank")
