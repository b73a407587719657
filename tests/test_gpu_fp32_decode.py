"""The decode-step projection at the reference's own precision (psg_skinny_gemm with PSG_F32, psg_gemm_f32.hip).

V4:99-100 loads the LLM without a dtype: the reference's q/k/v/o/gate/up/down projections and lm_head are fp32.
The kernel is exact fp32 arithmetic (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain; rows 16..19 through
v_mfma_f32_4x4x1_16b_f32), so it is held to fp32 round-off against an fp64 product: 4e-7 * sum|a b| per output
(the guide measures 3.5e-7 at K = 4096), not to a 16-bit tolerance.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


SHAPES = [(20, 4096, 4096), (20, 12288, 4096), (20, 22016, 4096), (20, 4096, 11008), (20, 32000, 4096),
          (1, 512, 256), (16, 4096, 4096), (17, 4096, 4096), (19, 768, 2752), (21, 4096, 4096), (32, 4096, 11008),
          (32, 48, 96), (5, 16, 32), (20, 1040, 256)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_fp32_skinny_gemm_vs_fp64(M, N, K):
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    # ASYMMETRIC operands: every row of x and of w is different, so a swapped row / column or a wrong K permutation
    # cannot cancel
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    part = ops.skinny_gemm(x, w)
    y = part.reduce(torch.float32)
    ref = x.double() @ w.double().t()
    bound = 4e-7 * (x.double().abs() @ w.double().abs().t()) + 1e-30
    rel = ((y.double() - ref).abs() / bound).max().item()
    lib = ((torch.nn.functional.linear(x, w).double() - ref).abs() / bound).max().item()
    print(f"M={M} N={N} K={K}: splits {part.splits}, err / (4e-7 sum|ab|) = {rel:.3f} (library SGEMM {lib:.3f})")
    assert rel < 1.0
    # batch invariance, bit for bit: a row's result does not depend on how many other rows ride along, nor on
    # whether it sits in the 16x16x4 block (rows 0..15) or in the 4x4x1 / second 16x16x4 tail of another batch size
    for m2 in sorted({1, min(M, 16), min(M, 20)} - {M}):
        y2 = ops.skinny_gemm(x[:m2].contiguous(), w, splits=part.splits).reduce(torch.float32)
        assert torch.equal(y2, y[:m2]), f"rows 0..{m2 - 1} differ between a batch of {m2} and of {M}"
    if M > 16:
        # a tail row (16..) computed alone in the main block of a batch of 1: the k-order inside a slice is the same
        # (block, piece, float) walk, but the tail sums four per-kq partials where the main block interleaves them -
        # equal to round-off, not bit for bit
        y1 = ops.skinny_gemm(x[16:17].contiguous(), w, splits=part.splits).reduce(torch.float32)
        assert ((y1[0].double() - y[16].double()).abs() / bound[16]).max().item() < 1.0
    # consumers that sum the split-K slices themselves agree with the materialised reduction
    if N % 8 == 0:
        out = torch.empty(M, N // 2, device=dev, dtype=torch.float32)
        ops.silu_mul(part, out)
        want = torch.empty_like(out)
        ops.silu_mul(y, want)
        assert torch.equal(out, want)


def test_fp32_skinny_gemm_rejects_bad_shapes():
    from openpsg_amd import ops
    from openpsg_amd._lib import PsgHipError
    dev = _dev()
    with pytest.raises(PsgHipError):
        ops.skinny_gemm(torch.zeros(33, 64, device=dev), torch.zeros(64, 64, device=dev))
    with pytest.raises(PsgHipError):
        ops.skinny_gemm(torch.zeros(4, 48, device=dev), torch.zeros(64, 48, device=dev))       # K % 32
    with pytest.raises(PsgHipError):
        ops.skinny_gemm(torch.zeros(4, 64, device=dev), torch.zeros(64, 64, device=dev).half())  # mixed dtypes


def test_fp32_engine_decode_uses_the_streaming_kernel_and_matches_library_path():
    """The fp32 engine's decode steps with the hand-written projections against the same engine with the library SGEMM
    (use_skinny = False): first-step logits to fp32 round-off, greedy tokens identical on a tiny LLM."""
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.llm import LlamaDecodeEngine
    from openpsg_amd.weights import make_weights_numpy
    dev = _dev()
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 2, 512, 512), max_object_num=30)
    w = make_weights_numpy(cfg, seed=3)
    eng = LlamaDecodeEngine(w, cfg, dev, torch.float32)
    K, Tp = 20, 9
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(K, cfg.qformer.num_query + Tp, cfg.llm.hidden, generator=g) * 0.5).to(dev)
    plen = torch.randint(3, Tp + 1, (K,), generator=g).to(dev, torch.int32)
    outs = {}
    for skinny in (True, False):
        eng.use_skinny = skinny
        eng._graphs.clear()
        toks, fl = eng.generate(X, plen, max_new_tokens=8, suppress_eos=True, return_first_logits=True)
        outs[skinny] = (toks.cpu(), fl.float().cpu())
    assert (outs[True][1] - outs[False][1]).abs().max().item() < 1e-4
    assert torch.equal(outs[True][0], outs[False][0])
