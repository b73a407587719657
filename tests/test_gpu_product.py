"""Product outputs that are NOT empty (`-m gpu`): a Llama rigged so that greedy decoding emits
`over ; in front of </s>` (tests/helpers.rig_llm_chain) drives the head's token -> triple parse (V4:313-326),
the detector's result packing (DET2:170-190), the submission writer (INFER:149-187), the precomputed-segmenter
ingest and tools/infer.py, and the batched C5-geometry path - in fp32 and in the benchmarked bf16 mode."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from openpsg_amd.categories import relation_categories
from oracle import psg_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rigged_head(dtype, max_objects=30, **kw):
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_numpy
    tok = H.ChainTokenizer()
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 2, 512, 256), max_object_num=max_objects)
    w = make_weights_numpy(cfg, seed=33)
    chain = H.rig_llm_chain(w, cfg, tok)
    head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", qformer_vocab_size=512, llm_config=cfg.llm,
                                     llm_feature_size=256, tokenizers=(H.WordTokenizer("bert"), tok),
                                     max_object_num=max_objects, **kw)
    head.load_weights(w)
    return head, tok, chain


def _inputs(scene):
    return dict(mask_features=scene["mask_features"].cuda(), img_metas=[scene["img_meta"]],
                object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"].cuda())])


def _expected(sel, N, tok, chain):
    seen = []
    text = "<s> " + tok.decode(chain)                              # what the reference parses (implicit BOS)
    for si in sel:
        O.parse_relations(text, int(si), N, relation_categories, seen)
    return seen


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_head_forward_emits_the_rigged_relations(dtype):
    from openpsg_amd.synthetic import make_scene
    head, tok, chain = _rigged_head(dtype)                         # default on_parse_error='raise': nothing may raise
    scene = make_scene((512, 512), 10, seed=5)
    out = head(_inputs(scene))
    toks = head.last["tokens_host"]
    sel = head.last["selected_host"].tolist()
    assert toks.shape == (20, 16) and len(sel) == 20
    for row in toks:                                               # natural EOS: the chain, then -1 padding
        assert [int(t) for t in row if t >= 0] == chain
    want = _expected(sel, 10, tok, chain)
    ri = relation_categories.index
    assert len(want) == 40 and want[0] == [sel[0] // 10, sel[0] % 10, ri("over")]
    assert want[1] == [sel[0] // 10, sel[0] % 10, ri("in front of")]
    assert out["rel_pred"] == want and out["rel_score"] == [1] * 40
    assert head.llm_engine.last_replays < 4                        # stopped at the first all-EOS check, not after 16 steps
    # DET2:177-181 hands the ids over as 0-d int32 tensors on the model's device: same result, fetched in one copy
    inp = _inputs(scene)
    inp["object_info"][0]["object_id_list"] = [t.cuda() for t in scene["object_id_list"]]
    assert head(inp)["rel_pred"] == want


def test_detector_simple_test_submission_and_infer_tool(tmp_path):
    """DET2:170-190 -> INFER:149-187, once through the synthetic segmenter and once through the precomputed ingest
    + tools/infer.py; both must give the same non-empty relations."""
    from openpsg_amd.detector import OpenSeeDRelationV2, SyntheticSegmenter
    from openpsg_amd.results import save_segmenter_output, write_submission
    from openpsg_amd.preprocess import image_meta
    from tools import infer
    head, tok, chain = _rigged_head("bf16")
    n_img, n_obj = 3, 6
    metas = [image_meta((480, 640), filename=f"img_{i}.jpg") for i in range(n_img)]     # 1000x1333 -> pad 1024x1344
    # what OpenSeeD would have produced, kept for the ingest round trip
    seg = SyntheticSegmenter(n_obj, seed=0, device="cuda:0")
    seg_dir = tmp_path / "seg"
    os.makedirs(seg_dir)
    produced = []
    for m in metas:
        pan, info, feat = seg(None, m)
        produced.append((pan, info, feat))
        save_segmenter_output(str(seg_dir / (os.path.splitext(m["filename"])[0] + ".npz")), pan.cpu().numpy(),
                              [s["id"] for s in info], [s["category_id"] for s in info], feat.cpu().numpy())

    class Replay:                                                  # the detector sees exactly those outputs
        def __init__(self):
            self.i = 0

        def __call__(self, img, meta):
            self.i += 1
            return produced[self.i - 1]

    det = OpenSeeDRelationV2(relation_head=head, segmenter=Replay())
    results = [det.simple_test(None, [m])[0] for m in metas]
    for res in results:
        assert set(res) >= {"pan_results", "object_id_list", "object_score_list", "ins_results", "rel_results",
                            "rel_scores"}
        assert isinstance(res["pan_results"], np.ndarray) and res["pan_results"].shape == (480, 640)
        assert len(res["rel_results"]["object_id_list"]) == n_obj
        assert len(res["rel_results"]["relation"]) == 40 and res["rel_scores"] == [1] * 40
        sel = head.last["selected_host"].tolist()
    path = write_submission(results, str(tmp_path / "direct"))
    direct = json.load(open(path))
    assert len(direct) == n_img and direct[0]["pan_seg_file_name"] == "0.png"
    for rec, res in zip(direct, results):
        assert rec["relations"] == [[s, o, r + 1] for s, o, r in res["rel_results"]["relation"]]   # INFER:177
        assert len(rec["segments_info"]) == n_obj
    # the same through tools/infer.py with the precomputed segmenter
    lst = tmp_path / "files.txt"
    lst.write_text("\n".join(m["filename"] for m in metas) + "\n")
    a = infer.parser().parse_args(["--segmenter", "precomputed", "--seg-dir", str(seg_dir), "--list", str(lst),
                                   "--ori-size", "480", "640", "--out", str(tmp_path / "tool"), "--keep-scores"])
    a.size_given = False
    tool_results, tool_path = infer.run(a, head=head)
    tool = json.load(open(tool_path))
    assert [t["relations"] for t in tool] == [d["relations"] for d in direct]
    assert all(t["relation_scores"] == [1] * 40 for t in tool)                                   # predict.py:97
    assert (tmp_path / "tool" / "submission" / "panseg" / "img_0.png").exists()
    for tr, res in zip(tool_results, results):
        assert np.array_equal(tr["pan_results"], res["pan_results"])
    # and with two images in flight (--in-flight 2: head.submit through OpenSeeDRelationV2.simple_test_submit)
    a2 = infer.parser().parse_args(["--segmenter", "precomputed", "--seg-dir", str(seg_dir), "--list", str(lst),
                                    "--ori-size", "480", "640", "--out", str(tmp_path / "tool2"), "--keep-scores",
                                    "--in-flight", "2"])
    a2.size_given = False
    _, path2 = infer.run(a2, head=head)
    assert [t["relations"] for t in json.load(open(path2))] == [d["relations"] for d in direct]


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "mixed"])
def test_c5_geometry_batch_of_eight_images(dtype):
    """BASELINE config 5's geometry (1000x1333 -> pad 1024x1344, L = 336) x 8 images through simple_test_batch:
    the selected pairs of all images decode together; every image must give what its own simple_test gives."""
    from openpsg_amd.detector import OpenSeeDRelationV2, SyntheticSegmenter
    from openpsg_amd.preprocess import image_meta
    head, tok, chain = _rigged_head(dtype, pair_selector="threshold", exclude_diagonal=True, max_selected=24)
    metas = [[image_meta((480, 640), filename=f"{i}.jpg")] for i in range(8)]
    det = OpenSeeDRelationV2(relation_head=head, segmenter=SyntheticSegmenter(8, seed=11, device="cuda:0"))
    single = [det.simple_test(None, m)[0] for m in metas]
    det.segmenter = SyntheticSegmenter(8, seed=11, device="cuda:0")
    batched = det.simple_test_batch([None] * 8, metas)
    n_rel = 0
    for s, b in zip(single, batched):
        assert b[0]["rel_results"]["relation"] == s["rel_results"]["relation"]
        assert np.array_equal(b[0]["pan_results"], s["pan_results"])
        assert all(t[0] != t[1] for t in s["rel_results"]["relation"])      # exclude_diagonal
        n_rel += len(s["rel_results"]["relation"])
    assert n_rel >= 8 * 2 * 4                                               # at least max_llm_forward_num pairs per image


def test_two_images_in_flight_give_the_results_of_forward():
    """`head.submit` / `.result()`: images enqueued one ahead on two HIP streams (slot = k % 2), results taken one behind -
    the decode graphs, KV caches and static buffers are per slot.  Every image's selection, token ids and parsed
    triples equal `forward`'s, for scenes of different object counts and sizes, in any interleaving."""
    import numpy as np
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    cfg = PSGConfig(qformer=QFormerConfig(vocab=30522), llm=tiny_llm(512, 2, 1024, 512), max_object_num=50)
    w = make_weights_device(cfg, 7, torch.device("cuda:0"), llm_dtype=torch.float32)
    for dtype in ("fp32", "mixed"):
        head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", llm_config=cfg.llm, llm_feature_size=512,
                                         tokenizers="word", max_object_num=50, on_parse_error="skip", suppress_eos=True)
        head.load_weights(w)
        geo = [((1024, 1024), 50), ((768, 1024), 23), ((512, 512), 9), ((1024, 1024), 50), ((1024, 1344), 31), ((512, 512), 9)]
        scenes = [make_scene(hw, n, seed=70 + m, device="cuda:0", tiny_object=True) for m, (hw, n) in enumerate(geo)]
        ins = [dict(mask_features=s["mask_features"], img_metas=[s["img_meta"]],
                    object_info=[dict(object_id_list=s["object_id_list"], pan_results=s["pan_results"])]) for s in scenes]
        want = []
        for i in ins:
            r = head(i)
            want.append((r, head.last["tokens_host"].copy(), head.last["selected"].cpu().numpy()))
        got, pending = [], []
        for k, i in enumerate(ins * 2):                               # twelve images, two in flight
            pending.append(head.submit(i, slot=k % 2))
            if len(pending) > 1:
                r = pending.pop(0).result()
                got.append((r, head.last["tokens_host"].copy(), head.last["selected_host"].copy()))
        r = pending.pop(0).result()
        got.append((r, head.last["tokens_host"].copy(), head.last["selected_host"].copy()))
        assert len(got) == 2 * len(ins)
        for k, (r, toks, sel) in enumerate(got):
            wr, wt, ws = want[k % len(ins)]
            assert np.array_equal(sel, ws) and np.array_equal(toks, wt) and r == wr, f"{dtype}: image {k} differs"
        empty = head.submit(dict(ins[0], object_info=[dict(object_id_list=[], pan_results=scenes[0]["pan_results"])]), slot=0)
        assert empty.result() == dict(rel_pred=[], rel_score=[])
        # a slot whose result was not taken cannot be submitted again (its static token buffers would be overwritten and the
        # first handle would silently return the second image's tokens); a taken or dropped handle frees the slot
        from openpsg_amd._lib import PsgHipError
        first = head.submit(ins[0], slot=0)
        with pytest.raises(PsgHipError, match="slot 0"):
            head.submit(ins[1], slot=0)
        assert first.result() == want[0][0]
        second = head.submit(ins[1], slot=0)
        del second                                                     # dropped without a result
        import gc
        gc.collect()
        assert head.submit(ins[2], slot=0).result() == want[2][0]


def test_submit_keeps_the_callers_temporaries_alive_until_they_are_read():
    """The caller drops its input tensors right after `submit` (a detector's mask_features are a temporary) and then
    allocates and overwrites same-sized blocks on ITS stream while the slot stream has not yet read the inputs: the head
    must hold the blocks (record_stream + a reference in the pending handle).  Results equal `forward`'s."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_device
    cfg = PSGConfig(qformer=QFormerConfig(vocab=30522), llm=tiny_llm(512, 2, 1024, 512), max_object_num=30)
    w = make_weights_device(cfg, 7, torch.device("cuda:0"), llm_dtype=torch.float32)
    head = RelationTransformerHeadV4(dtype="mixed", device="cuda:0", llm_config=cfg.llm, llm_feature_size=512,
                                     tokenizers="word", max_object_num=30, on_parse_error="skip", suppress_eos=True)
    head.load_weights(w)
    scenes = [make_scene((1024, 1024), 30, seed=90 + m, tiny_object=True) for m in range(4)]      # host copies

    def fresh(s):                                                      # a new device copy per image, as a segmenter makes
        return dict(mask_features=s["mask_features"].cuda(), img_metas=[s["img_meta"]],
                    object_info=[dict(object_id_list=s["object_id_list"], pan_results=s["pan_results"].cuda())])
    want = []
    for s in scenes:
        r = head(fresh(s))
        want.append((r, head.last["tokens_host"].copy()))
    torch.cuda.synchronize()
    got, pending = [], []
    for k in range(3 * len(scenes)):
        i = fresh(scenes[k % len(scenes)])
        pending.append(head.submit(i, slot=k % 2))
        shape = i["mask_features"].shape
        del i                                                          # the caller's references are gone
        for _ in range(3):                                             # the allocator would hand the block out again
            junk = torch.full(shape, float("nan"), device="cuda:0")
            del junk
        if len(pending) > 1:
            r = pending.pop(0).result()
            got.append((r, head.last["tokens_host"].copy()))
    r = pending.pop(0).result()
    got.append((r, head.last["tokens_host"].copy()))
    for k, (r, toks) in enumerate(got):
        wr, wt = want[k % len(scenes)]
        assert np.array_equal(toks, wt) and r == wr, f"image {k} differs from forward"


def test_submit_with_natural_eos_does_not_wait_for_the_decode():
    """Natural EOS (the product default): `submit` enqueues the prompt pass and the first chunk of decode steps and
    returns without reading the all-done flag back; the remaining chunks and their read-backs run in `result()`.
    Results equal `forward`'s, image by image, with two images in flight."""
    from openpsg_amd.synthetic import make_scene
    head, tok, chain = _rigged_head("bf16")
    scenes = [make_scene((512, 512), 10, seed=5 + m) for m in range(3)]
    ins = [_inputs(s) for s in scenes]
    want = [head(i) for i in ins]
    assert all(len(wr["rel_pred"]) == 40 for wr in want)
    eng = head.llm_engine
    calls = []
    orig = torch.Tensor.item

    def spy(self_):
        calls.append(1)
        return orig(self_)
    pending = []
    for k in range(6):
        torch.Tensor.item = spy
        try:
            n0 = len(calls)
            p = head.submit(ins[k % 3], slot=k % 2)
            assert len(calls) == n0, "submit read a device scalar back (a host wait)"
        finally:
            torch.Tensor.item = orig
        assert "_finish" in p.out and eng.last_replays == 2          # prompt pass + first token | first chunk of steps
        pending.append((k % 3, p))
        if len(pending) > 1:
            j, q = pending.pop(0)
            assert q.result() == want[j]
    j, q = pending.pop(0)
    assert q.result() == want[j]
    assert eng.last_replays < 5                                     # stopped at an all-EOS check, not after all five graphs
