"""World-size-2 `gloo` test (CPU) of the pair-sharding pipeline: the collectives, shard arithmetic,
padding of uneven shards, feature routing by reduce-scatter and the deterministic selection.  The
compute backend here is the CPU oracle (test infrastructure); on a GPU node the same pipeline
drives the HIP head (`HipBackend`)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class OracleBackend:
    """Implements the five compute calls of PairShardedPipeline with oracle/psg_oracle.py."""
    device = torch.device("cpu")

    def __init__(self, cfg, w, k=5, max_new=4, threshold=None):
        self.cfg, self.w, self.k, self.max_new, self.threshold = cfg, w, k, max_new, threshold
        self.q_rows, self.hidden = cfg.qformer.q_rows, cfg.qformer.hidden
        self.calls = []

    def num_objects(self, scene):
        return len(scene["object_id_list"])

    def patch_embed(self, scene):
        from oracle import psg_oracle as O
        return O.patch_embed(self.w, scene["mask_features"], 16)[0].contiguous()

    def query_shard(self, scene, patches, p0, p1):
        from oracle import psg_oracle as O
        from tests import helpers as H
        ids, tmask = H.qformer_prompts(scene)
        fh, fw = scene["mask_features"].shape[-2:]
        grid = O.mask_grid(scene["pan_results"], scene["img_meta"]["img_shape"], scene["img_meta"]["pad_shape"],
                           (fh // 16, fw // 16))
        pm = O.pair_masks(O.object_masks(grid, [int(i) for i in scene["object_id_list"]]))
        self.calls.append((p0, p1))
        if p1 <= p0:
            return torch.zeros(0, self.hidden), torch.zeros(0)
        out = O.qformer_forward(self.w, self.cfg, ids[p0:p1], tmask[p0:p1], patches, pm[p0:p1])
        _, prob = O.existence_head(self.w, out)
        return out.reshape(-1, self.hidden).contiguous(), prob

    def select(self, prob, num_objects):
        """'topk' (V4:235-237), or with a threshold the data-dependent count of the commented V4:230-234 logic."""
        from oracle import psg_oracle as O
        k = min(self.k, prob.numel())
        if self.threshold is not None:
            k = max(1, min(int((prob > self.threshold).sum()), 12))
        return torch.tensor(O.select_topk(prob, k), dtype=torch.int32)

    def gather_features(self, hidden, rows):
        out = torch.zeros(rows.numel(), self.hidden)
        ok = rows >= 0
        if ok.any():
            out[ok] = hidden[rows[ok].long()]
        return out

    def decode(self, scene, selected, features):
        # stand-in for the LMM: keeps what the reduce-scatter delivered (checked by the test) and
        # returns tokens derived from the selection (checks the token all-gather)
        self.received = features.clone()
        return (selected.long()[:, None] * 10 + torch.arange(self.max_new)[None, :]).to(torch.int32)


class ConstantsOracleBackend(OracleBackend):
    """The same with the SURVEY 8e hand-over: rank 0 computes an image's constants (object ids, bitmasks, patches), the
    pipeline broadcasts them as one int32 message, and every other rank works from them alone (its `scene` is None)."""

    def image_constants(self, scene):
        from oracle import psg_oracle as O
        ids = [int(i) for i in scene["object_id_list"]]
        fh, fw = scene["mask_features"].shape[-2:]
        grid = O.mask_grid(scene["pan_results"], scene["img_meta"]["img_shape"], scene["img_meta"]["pad_shape"],
                           (fh // 16, fw // 16))
        om = O.object_masks(grid, ids)                                              # bool [N, L]
        L = om.shape[1]
        words = (L + 63) // 64
        pad = torch.zeros((om.shape[0], words * 64), dtype=torch.int64)
        pad[:, :L] = om.to(torch.int64)
        w_ = (pad.view(-1, words, 64) << torch.arange(64, dtype=torch.int64)).sum(-1)   # bit l & 63 of word l >> 6
        return torch.tensor(ids, dtype=torch.int32), w_.contiguous(), self.patch_embed(scene)

    def scene_from_ids(self, ids):
        return dict(object_id_list=[int(i) for i in ids.tolist()])

    def query_shard(self, scene, patches, p0, p1, bits=None):
        if bits is None:
            return super().query_shard(scene, patches, p0, p1)
        from oracle import psg_oracle as O
        from tests import helpers as H
        ids, tmask = H.qformer_prompts(scene)                                       # from the object ids alone
        L = patches.shape[0]
        om = ((bits[:, :, None] >> torch.arange(64, dtype=torch.int64)) & 1).reshape(bits.shape[0], -1)[:, :L].bool()
        pm = O.pair_masks(om)
        self.calls.append((p0, p1))
        if p1 <= p0:
            return torch.zeros(0, self.hidden), torch.zeros(0)
        out = O.qformer_forward(self.w, self.cfg, ids[p0:p1], tmask[p0:p1], patches, pm[p0:p1])
        _, prob = O.existence_head(self.w, out)
        return out.reshape(-1, self.hidden).contiguous(), prob


def _worker(rank, world, port, n_obj, ret, threshold=None):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.dist import PairShardedPipeline, shard_range
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_numpy
    from oracle import psg_oracle as O
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 1, 512, 512))
    w = make_weights_numpy(cfg, seed=3, with_llm=False)
    # n_obj: one count for every image, or one count per image (different object counts AND image sizes in one step)
    # (more counts than ranks: several images per rank and step, image m owned by rank m % world)
    counts = [n_obj] * world if isinstance(n_obj, int) else list(n_obj)
    n_img = len(counts)
    sizes = [(256, 256)] * n_img if isinstance(n_obj, int) else [[(256, 256), (256, 384)][m % 2] for m in range(n_img)]
    scenes = [make_scene(sizes[m], counts[m], seed=40 + m, tiny_object=True) for m in range(n_img)]
    be = OracleBackend(cfg, w, threshold=threshold)
    with torch.no_grad():
        out = PairShardedPipeline(be, dist.group.WORLD, decode=True).step(scenes)
        # single-process reference: the whole pair range of every image on one rank
        ok = True
        for m in range(n_img):
            B = counts[m] * counts[m]
            be1 = OracleBackend(cfg, w, threshold=threshold)
            h, prob = be1.query_shard(scenes[m], be1.patch_embed(scenes[m]), 0, B)
            # shard-wise BLAS calls round differently from one big call: compare with a tolerance
            ok &= out["exist_prob"][m].shape == prob.shape and torch.allclose(out["exist_prob"][m], prob, atol=1e-5)
            sel = be.select(out["exist_prob"][m], counts[m]).tolist()
            ok &= out["selected"][m].tolist() == sel
            want_tok = be1.decode(scenes[m], torch.tensor(sel), h[:1])
            ok &= torch.equal(out["tokens"][m], want_tok)
            if m % world == rank and m + world >= n_img:          # the last image this rank decoded
                # features routed through the reduce-scatter == rows of the single-rank hidden state
                rows = (torch.tensor(sel)[:, None] * 33 + 1 + torch.arange(32)[None, :]).reshape(-1)
                ok &= torch.allclose(be.received, h[rows], atol=1e-4)
        want_calls = [shard_range(counts[m] ** 2, world, rank)[:2] for m in range(n_img)]
        ok &= be.calls == want_calls
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_obj", [4, 3])          # 16 pairs (even shards) and 9 pairs (uneven: 5 + 4)
def test_pair_sharding_world2_gloo(n_obj):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_obj, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_pair_sharding_two_images_per_rank_world2_gloo():
    """A step of P * world images (P = 2: bench.py's multi-GPU step, each rank decoding its two images side by side):
    image m is embedded and decoded by rank m % world; patches, features and token ids travel as [P, ...] blocks."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, (4, 3, 2, 4), ret, 0.5), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


@pytest.mark.parametrize("threshold", [None, 0.5])
def test_pair_sharding_unequal_images_world2_gloo(threshold):
    """Images of one step with different object counts (4 and 3 objects: 16 and 9 pairs, shards 8+8 and 5+4) and
    different sizes (patch lists of 16 and 24 rows), with the fixed-size and the data-dependent selector (different
    K per image: padded feature / token exchanges)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, (4, 3), ret, threshold), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _one_image_worker(rank, world, port, n_obj, ret):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.dist import PairShardedPipeline, deal_indices
    from openpsg_amd.synthetic import make_scene
    from openpsg_amd.weights import make_weights_numpy
    from oracle import psg_oracle as O
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 1, 512, 512))
    w = make_weights_numpy(cfg, seed=3, with_llm=False)
    scene = make_scene((256, 256), n_obj, seed=77, tiny_object=True)
    if rank != 0:                               # only rank 0 holds the image; the others work from its broadcast alone
        scene = None
    be = ConstantsOracleBackend(cfg, w)
    real_pe = be.patch_embed
    be.patch_embed = lambda sc: real_pe(sc) if rank == 0 else (_ for _ in ()).throw(AssertionError("rank > 0"))
    with torch.no_grad():
        out = PairShardedPipeline(be, dist.group.WORLD, decode=True).step_one_image(scene)
        B = n_obj * n_obj
        full_scene = make_scene((256, 256), n_obj, seed=77, tiny_object=True)
        be1 = OracleBackend(cfg, w)
        h, prob = be1.query_shard(full_scene, be1.patch_embed(full_scene), 0, B)
        ok = torch.allclose(out["exist_prob"], prob, atol=1e-5)
        sel = O.select_topk(out["exist_prob"], min(be.k, B))
        ok &= out["selected"].tolist() == sel
        ok &= torch.equal(out["tokens"], be1.decode(full_scene, torch.tensor(sel), h[:1]))
        mine = deal_indices(len(sel), world, rank)              # this rank decoded exactly its dealt pairs
        rows = (torch.tensor([sel[i] for i in mine])[:, None] * 33 + 1 + torch.arange(32)[None, :]).reshape(-1)
        ok &= torch.allclose(be.received, h[rows], atol=1e-4)
        # the relation query alone (bench.py's `strong_scaling_rq`: C4's sharded part): the same probabilities and the same
        # selection on every rank, no feature exchange, no decode
        rq_only = PairShardedPipeline(be, dist.group.WORLD, decode=False).step_one_image(scene)
        ok &= torch.equal(rq_only["exist_prob"], out["exist_prob"]) and torch.equal(rq_only["selected"], out["selected"])
        ok &= "tokens" not in rq_only
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_obj", [4, 3])
def test_one_image_strong_scaling_world2_gloo(n_obj):
    """SURVEY 8e incl. item 3: one image for both ranks - its constants (object ids, bitmask words, patches) broadcast
    from rank 0 as one int32 message through gloo, rank 1 called with scene=None -, pair shards, identical top-K,
    selected features all-reduced, the K decodes dealt round-robin, tokens gathered in selection order."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_one_image_worker, args=(world, port, n_obj, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_range_covers_all_pairs():
    from openpsg_amd.dist import shard_range
    for B in (1, 9, 16, 100, 2500, 10000):
        for R in (1, 2, 4, 8):
            seen = []
            for r in range(R):
                p0, p1, shard = shard_range(B, R, r)
                assert p1 - p0 <= shard
                seen += list(range(p0, p1))
            assert seen == list(range(B))


def _image_worker(rank, world, port, n_img, ret):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from openpsg_amd.dist import gather_image_results, shard_images
    mine = shard_images(n_img, world, rank)
    local = [(i, dict(pan_results=np.full((2, 2), i), rel_results=dict(relation=[[i, i + 1, rank]]))) for i in mine]
    full = gather_image_results(local, n_img)
    ok = len(full) == n_img
    for i, r in enumerate(full):
        ok &= int(r["pan_results"][0, 0]) == i and r["rel_results"]["relation"] == [[i, i + 1, i % world]]
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_img", [8, 5, 1])       # even deal, uneven deal, fewer images than ranks
def test_image_sharding_world2_gloo(n_img):
    """C5: whole images dealt round-robin to the ranks; every rank gets all results back in image order."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_image_worker, args=(world, port, n_img, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_gather_image_results_rejects_gaps_and_duplicates():
    from openpsg_amd.dist import gather_image_results, shard_images
    assert shard_images(5, 2, 0) == [0, 2, 4] and shard_images(5, 2, 1) == [1, 3] and shard_images(1, 2, 1) == []
    assert gather_image_results([(1, "b"), (0, "a")], 2) == ["a", "b"]
    with pytest.raises(RuntimeError):
        gather_image_results([(0, "a")], 2)
    with pytest.raises(RuntimeError):
        gather_image_results([(0, "a"), (0, "b")], 1)
