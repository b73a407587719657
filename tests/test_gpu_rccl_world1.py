"""RCCL first-run insurance (`-m gpu`, one MI355X): every collective wrapper of `PairShardedPipeline` goes through the
`nccl` backend (= RCCL on ROCm) at world size 1 with the dtypes and the padded shapes the 8-GPU BASELINE-C4 run uses
(shard 1250 of 10 000 pairs, K = 20, 3 decodes per rank), plus both pipelines end to end against the single-GPU head.
The driver alone can launch 8 ranks; what can fail for reasons other than the rank count - dtype support of
reduce_scatter_tensor, the two-phase broadcast, all_gather_object - fails here first."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


class _NoCompute:
    device = torch.device("cuda", 0)

    def query_shard(self, *a):
        raise AssertionError("not used")


def test_collective_wrappers_with_c4_shapes_and_dtypes(rccl):
    from openpsg_amd.dist import PairShardedPipeline, gather_image_results
    dev = torch.device("cuda", 0)
    pipe = PairShardedPipeline(_NoCompute(), rccl.group.WORLD, decode=True)
    assert pipe.world == 1 and pipe.rank == 0 and rccl.get_backend() == "nccl"
    g = torch.Generator(device=dev).manual_seed(0)
    R8, K, nv, hidden, per = 8, 20, 32, 768, 3
    # all_gather: padded probability shard (fp32), dealt token block (int32), 16-bit features
    for t in (torch.rand(1250, device=dev, generator=g), torch.randint(0, 32000, (per, 16), device=dev, generator=g).int(),
              torch.randn(K * nv, hidden, device=dev, generator=g).half(),
              torch.randn(R8, 1250, device=dev, generator=g)):
        out = pipe._all_gather(t)
        assert out.shape == (1,) + tuple(t.shape) and out.dtype == t.dtype and torch.equal(out[0], t)
    # reduce_scatter(sum) of the selected pair features: fp16 / bf16 / fp32, [R, Kmax*nv, hidden]
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        send = torch.randn(1, K * nv, hidden, device=dev, generator=g).to(dt)
        recv = pipe._reduce_scatter_sum(send)
        assert recv.shape == (K * nv, hidden) and recv.dtype == dt and torch.equal(recv, send[0])
    # all_reduce(sum) of the strong-scaling feature exchange
    for dt in (torch.float16, torch.float32):
        t = torch.randn(K * nv, hidden, device=dev, generator=g).to(dt)
        want = t.clone()
        rccl.all_reduce(t, op=rccl.ReduceOp.SUM)
        assert torch.equal(t, want)
    # two-phase broadcast of the image-constants message (int32) and of a plain fp32 tensor
    msg = torch.randint(-2 ** 31, 2 ** 31 - 1, (4 + 100 + 2 * 100 * 4 + 256 * 256,), device=dev, generator=g, dtype=torch.int64).int()
    for t in (msg, torch.randn(256, 256, device=dev, generator=g)):
        got = pipe._broadcast(t, 0, dev)
        assert got.dtype == t.dtype and torch.equal(got, t)
    # host objects (results of tools/infer.py's image dealing)
    res = [(0, dict(rel_pred=[[1, 2, 3]], pan=np.arange(6).reshape(2, 3))), (1, dict(rel_pred=[], pan=np.zeros((1, 1))))]
    back = gather_image_results(res, 2, always_collective=True)
    assert back[0]["rel_pred"] == [[1, 2, 3]] and np.array_equal(back[0]["pan"], res[0][1]["pan"]) and back[1]["rel_pred"] == []


@pytest.mark.parametrize("dtype", ["fp32", "fp32s", "mixed"])
def test_pipelines_over_rccl_world1_match_the_head(rccl, dtype):
    """`step_one_image` and `step` driven by the real process group (every collective a RCCL call) == head.forward."""
    from openpsg_amd.dist import PairShardedPipeline
    from openpsg_amd.synthetic import make_scene
    from tests.test_gpu_fakeworld import _inputs, _mk_head
    head = _mk_head(dtype, 30)
    scene = make_scene((512, 768), 14, seed=8, device="cuda:0", tiny_object=True)
    head(_inputs(scene))
    ref = dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(), tokens=head.last["tokens_host"].copy())
    pipe = PairShardedPipeline(head, rccl.group.WORLD, decode=True)
    one = pipe.step_one_image(scene)
    assert torch.equal(one["exist_prob"], ref["prob"]) and torch.equal(one["selected"], ref["sel"])
    assert np.array_equal(one["tokens"].cpu().numpy(), ref["tokens"])
    many = pipe.step([scene])
    assert torch.equal(many["exist_prob"][0], ref["prob"]) and torch.equal(many["selected"][0], ref["sel"])
    assert np.array_equal(many["tokens"][0].cpu().numpy(), ref["tokens"])

    # two images per rank and step (the side-by-side decodes of `bench.py --gpus N --in-flight 2`): the collectives carry
    # [P, ...] blocks, a rank's P decodes run on the head's slot streams
    scene2 = make_scene((512, 512), 9, seed=9, device="cuda:0", tiny_object=True)
    head(_inputs(scene2))
    ref2 = dict(prob=head.last["exist_prob"].clone(), sel=head.last["selected"].clone(), tokens=head.last["tokens_host"].copy())
    both = pipe.step([scene, scene2])
    for m, rf in enumerate((ref, ref2)):
        assert torch.equal(both["exist_prob"][m], rf["prob"]) and torch.equal(both["selected"][m], rf["sel"])
        assert np.array_equal(both["tokens"][m].cpu().numpy(), rf["tokens"])
