"""Decode projections with their producer row operation in the same launch (psg_skinny_gemm_fused).

The fused kernel runs the SAME arithmetic in the SAME order as the separate RMSNorm kernel it replaces, so it is
compared BIT FOR BIT with psg_rmsnorm + psg_skinny_gemm, and at the engine level by its greedy tokens."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _partials(splits, rows, cols, g, dev, scale=0.05):
    from openpsg_amd import ops
    return ops.Partials((torch.randn(splits, rows, cols, generator=g) * scale).to(dev))


def _sync(dev):
    return torch.zeros(2, device=dev, dtype=torch.int32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N,dsplits", [(20, 4096, 12288, 8), (20, 4096, 22016, 8), (7, 4096, 32000, 8),
                                           (32, 4096, 4096, 3), (20, 1024, 3072, 4), (5, 1024, 2048, 0),
                                           (20, 4096, 12288, -1)])
def test_rmsnorm_prologue_bit_identical(dtype, M, K, N, dsplits):
    """x += sum(delta partials); n = rmsnorm(x); n @ w.T -- separate kernels vs one launch.  dsplits: > 0 split-K
    partials, 0 an activation-dtype delta, -1 no delta (first norm of a step)."""
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + K + dsplits)
    resid = (torch.randn(M, K, generator=g) * 0.7).to(dev).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev).to(dtype)
    ln = (1.0 + 0.2 * torch.randn(K, generator=g)).to(dev)
    if dsplits > 0:
        delta = _partials(dsplits, M, K, g, dev, 0.2)
    elif dsplits == 0:
        delta = (torch.randn(M, K, generator=g) * 0.3).to(dev).to(dtype)
    else:
        delta = None
    r1, n1 = resid.clone(), torch.empty_like(resid)
    ops.rmsnorm(r1, delta, ln, 1e-5, n1)
    want = ops.skinny_gemm(n1, w)
    for rep in range(3):                                       # repeated: the hand-off must not depend on timing
        r2, n2, sync = resid.clone(), torch.full_like(resid, float("nan")), _sync(dev)
        got = ops.skinny_gemm_fused(ops.PSG_PRO_RMSNORM, n2, w, sync, inp=delta, resid=r2, norm_w=ln, eps=1e-5,
                                    splits=want.splits)
        torch.cuda.synchronize()
        assert sync.tolist() == [M, 0], sync.tolist()          # every row arrived, no poll gave up
        assert torch.equal(r2, r1)
        assert torch.equal(n2.view(torch.int16), n1.view(torch.int16))
        assert torch.equal(got.t, want.t)


def test_fused_decode_tokens_equal_separate_kernels():
    """Engine level: the greedy decode with fused row operations emits the tokens (and first-step logits) of the
    decode with separate row kernels, eager and from the captured graph."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.llm import LlamaDecodeEngine
    from openpsg_amd.weights import make_weights_device
    dev = _dev()
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(1024, 3, 2048, 1024), max_new_tokens=16)
    w = make_weights_device(cfg, 5, dev, llm_dtype=torch.bfloat16)
    eng = LlamaDecodeEngine(w, cfg, dev, torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    K, Tp = 20, 16
    X = (torch.randn(K, 32 + Tp, cfg.llm.hidden, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    plen = torch.randint(9, Tp + 1, (K,), generator=g).to(torch.int32).to(dev)
    outs = {}
    for mode in ("", "rmsnorm"):
        eng.fuse_rowops = frozenset(mode.split())
        if mode and not eng._can_fuse(K):
            pytest.fail("fused path not available for the test model")
        eng._graphs.clear()
        eng.use_graph = False
        t_eager = eng.generate(X, plen, suppress_eos=True).clone()
        eng.use_graph = True
        for _ in range(2):
            t_graph = eng.generate(X, plen, suppress_eos=True)
            assert torch.equal(t_graph, t_eager), mode
        outs[mode] = t_eager
    for mode, t in outs.items():
        assert torch.equal(t, outs[""]), f"fuse_rowops={mode!r} changed the greedy tokens"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(20, 22016, 4096), (32, 22016, 4096), (20, 22000, 1024), (3, 5632, 2048)])
def test_wide_slab_variant_bit_identical(dtype, M, N, K):
    """Option skinny_wide: 11-wave (176-row) slabs where they save a round of slabs (the gate/up projection: 172 slabs
    of 128 rows over 64 column groups = 3 rounds, 126 slabs of 176 rows = 2).  Same per-row arithmetic -> the partials
    must be bit-identical to the 8-wave kernel, also with a partial last slab (N not a multiple of 176)."""
    from openpsg_amd import _lib, ops
    dev = _dev()
    g = torch.Generator().manual_seed(N + K + M)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev).to(dtype)
    try:
        _lib.set_option(0, "skinny_wide", 0)
        want = ops.skinny_gemm(x, w)
        _lib.set_option(0, "skinny_wide", 1)
        got = ops.skinny_gemm(x, w, splits=want.splits)
    finally:
        _lib.set_option(0, "skinny_wide", 1)
    torch.cuda.synchronize()
    assert torch.equal(got.t, want.t)
    ref = (x.float() @ w.float().T)
    assert (got.reduce(torch.float32) - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_splits_12_wave_plan_for_qkv_shape(dtype):
    """q/k/v projection (N = 12288, K = 4096, 20 rows): the planner halves the K slices (8 -> 4) and the launch uses
    12-wave slabs (64 slabs of 192 rows = one round): half the fp32 partials.  Bit-identical to the 8-wave kernel run
    with the same 4 slices, and the reduced result matches an fp32 matmul."""
    from openpsg_amd import _lib, ops
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(20, 4096, generator=g) * 0.5).to(dev).to(dtype)
    w = (torch.randn(12288, 4096, generator=g) * 0.02).to(dev).to(dtype)
    try:
        _lib.set_option(0, "skinny_wide", 1)
        got = ops.skinny_gemm(x, w)
        assert got.splits == 4, got.splits
        _lib.set_option(0, "skinny_wide", 0)
        assert ops.skinny_gemm(x, w).splits == 8
        want = ops.skinny_gemm(x, w, splits=4)
    finally:
        _lib.set_option(0, "skinny_wide", 1)
    torch.cuda.synchronize()
    assert torch.equal(got.t, want.t)
    ref = x.float() @ w.float().T
    assert (got.reduce(torch.float32) - ref).abs().max().item() < 1e-3 * ref.abs().max().item() + 1e-4
