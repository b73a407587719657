"""`psg_dense_gemm_tiled` (`-m gpu`): the Llama prompt pass's projections (HF-LL:163-177) on the repo's own MFMA GEMM.
Every tile geometry must give the SAME bits (each output element is one k-ordered accumulation over the whole K), the
result must match an fp32 torch product to 16-bit rounding, ragged N (not a multiple of the tile) and M must work, and
the SwiGLU epilogue must equal the unfused pair GEMM -> psg_silu_mul bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = ["256x256", "256x192", "256x128", "256x64", "128x128", "auto"]


def _data(M, N, K, dt, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dt)
    return x, w


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (100, 1376, 256), (257, 4096, 512), (980, 2752, 1024), (49, 16, 128)])
def test_every_tile_gives_the_same_bits_and_matches_fp32(dt, M, N, K):
    from openpsg_amd import ops
    x, w = _data(M, N, K, dt, 11)
    want = (x.float() @ w.float().t())
    base = ops.dense_gemm(x, w, tile="256x256")
    tol = (2 ** -8 if dt == torch.bfloat16 else 2 ** -11) * 1.01
    err = ((base.float() - want).abs() / (want.abs() + 1.0)).max().item()
    assert err <= tol, err
    for t in TILES[1:]:
        got = torch.full((M + 3, N), float("nan"), device="cuda", dtype=dt)      # rows past M must stay untouched
        ops.dense_gemm(x, w, out=got[:M], tile=t)
        assert torch.equal(got[:M], base), f"tile {t} differs"
        assert torch.isnan(got[M:]).all()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,inter,K", [(980, 1376, 512), (33, 688, 256), (1, 8, 64)])
def test_swiglu_epilogue_equals_gemm_then_silu_mul(dt, M, inter, K):
    from openpsg_amd import ops
    x, w = _data(M, 2 * inter, K, dt, 5)
    wi = ops.interleave_gate_up(w)
    # the interleave itself: groups of 16 rows = 8 gate rows, 8 up rows
    idx = torch.arange(2 * inter, device="cuda")
    p, r = idx // 16, idx % 16
    src = torch.where(r < 8, 8 * p + r, inter + 8 * p + r - 8)
    assert torch.equal(wi, w[src])
    want = torch.empty(M, inter, device="cuda", dtype=dt)
    if (2 * inter) % 16 == 0:
        gu = ops.dense_gemm(x, w, tile="256x256")
        ops.silu_mul(gu, want)
    for t in TILES:
        got = ops.dense_gemm(x, wi, swiglu=True, tile=t)
        assert got.shape == (M, inter)
        assert torch.equal(got, want), f"tile {t}: fused SwiGLU differs from GEMM + silu_mul"
    # and against fp32 arithmetic
    gu32 = x.float() @ w.float().t()
    ref = torch.nn.functional.silu(gu32[:, :inter]) * gu32[:, inter:]
    tol = 2 ** -6 if dt == torch.bfloat16 else 2 ** -9
    assert ((want.float() - ref).abs() / (ref.abs() + 1.0)).max().item() < tol


def test_auto_tile_and_argument_checks():
    from openpsg_amd import ops
    x, w = _data(64, 256, 64, torch.float16, 1)
    with pytest.raises(Exception):
        ops.dense_gemm(x, w, gelu=True, tile="256x192")                          # GELU: 256 x 256 only
    with pytest.raises(Exception):
        ops.dense_gemm(x, w[:250], tile="auto")                                  # N % 16
    b = torch.zeros(256, device="cuda")
    with pytest.raises(Exception):
        ops.dense_gemm(x, w, bias=b, swiglu=True)


@pytest.mark.parametrize("gelu", [False, True])
def test_fp32_output_tiles_give_the_same_bits(gelu):
    """The split products of the fp32s Q-Former (fp16 operands, fp32 output with row / column scales, bias, optional GELU):
    the small-M projections run on smaller tiles - every geometry built with the fp32 output gives the bits of the
    256 x 256 tile (HF-IB:519-596)."""
    from openpsg_amd import ops
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for M, N, K in ((2500, 768, 2304), (660, 3072, 2304), (33, 2304, 2304), (920, 768, 9216)):
        x = torch.randn(M, K, generator=g, device="cuda:0").half()
        w = (torch.randn(N, K, generator=g, device="cuda:0") / K ** 0.5).half()
        b = torch.randn(N, generator=g, device="cuda:0")
        rs = torch.exp2(torch.randint(-8, 4, (M,), generator=g, device="cuda:0").float())
        cs = torch.exp2(torch.randint(-8, 4, (N,), generator=g, device="cuda:0").float())
        base = ops.dense_gemm(x, w, b, gelu=gelu, out_dtype=torch.float32, row_scale=rs, col_scale=cs, tile="256x256")
        for t in ("256x128", "256x64", "128x128", "auto"):
            got = ops.dense_gemm(x, w, b, gelu=gelu, out_dtype=torch.float32, row_scale=rs, col_scale=cs, tile=t)
            assert torch.equal(got, base), f"tile {t} differs at {M}x{N}x{K}"
