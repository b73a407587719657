"""The single-process "fake world" (SURVEY 4): `LoopbackWorld` advances R rank generators of the pair-sharded pipelines
in lockstep and serves their collectives from Python lists.  Here with the CPU oracle as the compute backend (world
sizes 2-4, unequal images, data-dependent selection, dealt decodes); tests/test_gpu_fakeworld.py drives the HIP head
through the same object."""
import pytest
import torch

from tests.test_dist_gloo import ConstantsOracleBackend, OracleBackend


def _setup(threshold=None):
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.weights import make_weights_numpy
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 1, 512, 512))
    w = make_weights_numpy(cfg, seed=3, with_llm=False)
    return cfg, w


@pytest.mark.parametrize("world,counts,threshold", [(2, (4, 3), None), (3, (3, 4, 2), 0.5), (4, (3, 3, 3, 3), None),
                                                    (2, (4, 3, 2, 4), 0.5), (3, (2, 3, 3, 3, 2, 4), None)])
def test_loopback_step_matches_single_rank(world, counts, threshold):
    """len(counts) = P * world images per step: image m belongs to rank m % world."""
    from openpsg_amd.dist import LoopbackWorld, shard_range
    from openpsg_amd.synthetic import make_scene
    torch.set_num_threads(4)
    cfg, w = _setup()
    sizes = [(256, 256), (256, 384), (192, 256), (256, 256)]
    n_img = len(counts)
    scenes = [make_scene(sizes[m % 4], counts[m], seed=40 + m, tiny_object=True) for m in range(n_img)]
    bes = [OracleBackend(cfg, w, threshold=threshold) for _ in range(world)]
    fw = LoopbackWorld(world)
    with torch.no_grad():
        outs = fw.run([p.step_gen(scenes) for p in fw.pipelines(bes)])
        for m in range(n_img):
            B = counts[m] ** 2
            be1 = OracleBackend(cfg, w, threshold=threshold)
            h, prob = be1.query_shard(scenes[m], be1.patch_embed(scenes[m]), 0, B)
            sel = be1.select(prob, counts[m]).tolist()
            rows = (torch.tensor(sel)[:, None] * 33 + 1 + torch.arange(32)[None, :]).reshape(-1)
            if m + world >= n_img:                                                # the last image its rank decoded
                assert torch.allclose(bes[m % world].received, h[rows], atol=1e-4)
            for r in range(world):                                                # every rank holds every result
                assert torch.allclose(outs[r]["exist_prob"][m], prob, atol=1e-5)
                assert outs[r]["selected"][m].tolist() == sel
                assert torch.equal(outs[r]["tokens"][m], be1.decode(scenes[m], torch.tensor(sel), h[:1]))
    for r in range(world):
        assert bes[r].calls == [shard_range(counts[m] ** 2, world, r)[:2] for m in range(n_img)]


@pytest.mark.parametrize("world,n_obj", [(2, 4), (3, 3), (4, 4), (8, 3)])
def test_loopback_one_image_dealt_decodes(world, n_obj):
    from openpsg_amd.dist import LoopbackWorld, deal_indices
    from openpsg_amd.synthetic import make_scene
    torch.set_num_threads(4)
    cfg, w = _setup()
    scene = make_scene((256, 256), n_obj, seed=77, tiny_object=True)
    bes = [ConstantsOracleBackend(cfg, w) for _ in range(world)]
    fw = LoopbackWorld(world)
    with torch.no_grad():
        # only rank 0 holds the image; the others get scene=None and work from the broadcast constants
        outs = fw.run([p.step_one_image_gen(scene if r == 0 else None) for r, p in enumerate(fw.pipelines(bes))])
        B = n_obj * n_obj
        be1 = OracleBackend(cfg, w)
        h, prob = be1.query_shard(scene, be1.patch_embed(scene), 0, B)
        sel = be1.select(prob, n_obj).tolist()
        for r in range(world):
            assert torch.allclose(outs[r]["exist_prob"], prob, atol=1e-5)
            assert outs[r]["selected"].tolist() == sel
            assert torch.equal(outs[r]["tokens"], be1.decode(scene, torch.tensor(sel), h[:1]))
            mine = deal_indices(len(sel), world, r)
            if mine:
                rows = (torch.tensor([sel[i] for i in mine])[:, None] * 33 + 1 + torch.arange(32)[None, :]).reshape(-1)
                assert torch.allclose(bes[r].received, h[rows], atol=1e-4)
            else:
                assert not hasattr(bes[r], "received")                            # more ranks than selected pairs


def test_loopback_rejects_diverging_ranks():
    from openpsg_amd.dist import LoopbackWorld

    def a():
        yield ("all_gather", torch.zeros(1))

    def b():
        yield ("all_reduce", torch.zeros(1))
    with pytest.raises(RuntimeError):
        LoopbackWorld(2).run([a(), b()])


def test_constants_message_round_trip_on_cpu():
    """pack_constants / unpack_constants: ids, 64-bit mask words (including bit 63) and fp32 patches (including NaN and
    -0.0 bit patterns) survive the int32 message bit for bit."""
    from openpsg_amd.dist import PairShardedPipeline
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 133000, (7,), generator=g).to(torch.int32)
    bits = torch.randint(-2 ** 63, 2 ** 63 - 1, (7, 5), generator=g, dtype=torch.int64)
    bits[0, 0] = -2 ** 63                                                        # only bit 63 set
    patches = torch.randn(21, 16, generator=g)
    patches[3, 2], patches[4, 4] = float("nan"), -0.0
    msg = PairShardedPipeline.pack_constants(ids, bits, patches)
    assert msg.dtype == torch.int32 and msg.numel() == 4 + 7 + 2 * 35 + 21 * 16
    i2, b2, p2 = PairShardedPipeline.unpack_constants(msg)
    assert torch.equal(i2, ids) and torch.equal(b2, bits)
    assert torch.equal(p2.view(torch.int32), patches.view(torch.int32))
