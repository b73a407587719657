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


def _fuzz_shapes():
    """PSG_FUZZ_GEMM=count (a one-off sweep, profiles/r06_fuzz_gemm.txt): `count` random shapes - 1..32 rows, N any multiple
    of 16 up to 8192, K any multiple of 32 up to 12288 - next to the fixed list."""
    import os
    import random
    n = int(os.environ.get("PSG_FUZZ_GEMM", "0"))
    r = random.Random(2024)
    return [(r.randint(1, 32), 16 * r.randint(1, 512), 32 * r.randint(1, 384)) for _ in range(n)]


@pytest.mark.parametrize("M,N,K", SHAPES + _fuzz_shapes())
def test_fp32_skinny_gemm_vs_fp64(M, N, K):
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    # ASYMMETRIC operands: every row of x and of w is different, so a swapped row / column or a wrong K permutation
    # cannot cancel
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    from openpsg_amd import _lib
    try:
        part = ops.skinny_gemm(x, w)
    except _lib.PsgHipError as e:                      # (sweep shapes only) more rows x K than the LDS holds beside the rings
        assert K > 11776 and M > 20 and "do not fit the LDS" in str(e), (M, N, K, str(e))
        return
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


# ---- the prompt pass of the reference-precision mode as split-fp16 products ('fp32s', psg_split.hip) -------------------
@pytest.mark.parametrize("M,N,K,spread", [(980, 4096, 4096, 1.0), (257, 768, 2752, 1.0), (64, 512, 256, 1e-4), (333, 1024, 1024, 1e3)])
def test_split_f16x3_product_is_fp32_grade(M, N, K, spread):
    """[xh | xh | xl] . [wh | wl | wh]^T with fp32 accumulation against an fp64 product: within 1.5e-6 * sum|x w| per
    output (3 * 2^-22 = 7e-7 per product + fp32 accumulation), i.e. the fp32 class - the library SGEMM sits at ~3e-7 on
    the same data, a plain fp16 GEMM at 5e-4.  `spread`: rows / columns of very different magnitude (1e-4 .. 1e3)
    exercise the per-row power-of-two scaling."""
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * torch.logspace(0, float(np.log10(spread)), M)[:, None]
    w = torch.randn(N, K, generator=g) / K ** 0.5 * torch.logspace(0, float(np.log10(spread)), N)[:, None]
    x, w = x.to(dev), w.to(dev)
    a3, inv_r = ops.split_f16x3(x)
    b3, inv_c = ops.split_f16x3(w, weights=True)
    assert a3.shape == (M, 3 * K) and torch.equal(a3[:, :K], a3[:, K:2 * K]) and torch.equal(b3[:, :K], b3[:, 2 * K:])
    assert bool(torch.isfinite(a3.float()).all()) and float(a3.float().abs().max()) < 16384.5
    y = ops.scale_rows_cols(torch.mm(a3, b3.t(), out_dtype=torch.float32), inv_r, inv_c)
    ref = x.double() @ w.double().t()
    bound = (x.double().abs() @ w.double().abs().t()) + 1e-300
    rel = ((y.double() - ref).abs() / bound).max().item()
    lib = ((torch.nn.functional.linear(x, w).double() - ref).abs() / bound).max().item()
    h16 = ((torch.nn.functional.linear(x.half(), w.half()).double() - ref).abs() / bound).max().item()
    print(f"M={M} N={N} K={K} spread={spread}: split-fp16 {rel:.2e}, library SGEMM {lib:.2e}, plain fp16 GEMM {h16:.2e} "
          f"(x sum|x w|)")
    assert rel < 1.5e-6


import numpy as np  # noqa: E402

from tests import helpers as H  # noqa: E402

CASES = ["G1_c1_512_n10", "G2_768x1024_n12", "G4_llm_wide_n6", "G5_c5geo_1024x1344_n8", "G6_llm_7b_width_n6"]


@pytest.mark.parametrize("case", CASES)
def test_fp32s_mode_meets_the_fp32_bar_on_the_reference_goldens(case):
    """dtype='fp32s' (fp32 everywhere, the prompt pass's projections as split-fp16 products) against the captures of the
    real reference head: existence logits within 1e-3, selection identical, every greedy token of every selected pair
    identical, first-step top-8 logits within 1e-3 - the same assertions as the exact-fp32 mode (test_gpu_parity.py)."""
    from openpsg_amd.head import RelationTransformerHeadV4
    g, cfg, w, scene = H.load_case(case)
    head = RelationTransformerHeadV4(dtype="fp32s", device="cuda:0", qformer_vocab_size=cfg.qformer.vocab,
                                     llm_config=cfg.llm, llm_feature_size=cfg.llm.hidden, tokenizers="word",
                                     max_object_num=cfg.max_object_num, on_parse_error="skip",
                                     suppress_eos=bool(g["suppress_eos"]))
    head.load_weights(w)
    assert head.llm_engine.prefill_split
    dev = _dev()
    head(dict(mask_features=scene["mask_features"].to(dev), img_metas=[scene["img_meta"]],
              object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"].to(dev))]))
    last = head.last
    assert np.abs(last["exist_logit"].cpu().numpy() - g["exist_logit"]).max() < 1e-3
    assert last["selected"].cpu().tolist() == g["selected"].tolist()
    toks = last["tokens_host"]
    fl = last["first_logits"].float().cpu().numpy()
    worst = 0.0
    for i in range(toks.shape[0]):
        want = g["gen_tokens"][i]
        assert [int(t) for t in toks[i] if t >= 0] == want[want >= 0].tolist(), f"pair #{i}: greedy tokens differ"
        worst = max(worst, float(np.abs(fl[i][g["gen_top8_idx"][i]] - g["gen_top8_val"][i]).max()))
    print(f"{case}: fp32s first-step top-8 logits within {worst:.2e} of the reference")
    assert worst < 1e-3


@pytest.mark.parametrize("M,N,K,gelu", [(2500 * 3, 768, 768, False), (980, 4096, 4096, False), (515, 3072, 768, True), (33, 2304, 768, False)])
def test_dense_gemm_fp32_output_with_split_scales(M, N, K, gelu):
    """psg_dense_gemm_ex: fp16 split operands in, fp32 out, power-of-two scales and the bias applied in the epilogue -
    the Q-Former's Linear layers in the fp32s mode - against fp64; and row-count invariance, bit for bit."""
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    a3, inv_r = ops.split_f16x3(x)
    b3, inv_c = ops.split_f16x3(w, weights=True)
    y = ops.dense_gemm(a3, b3, b, gelu=gelu, out_dtype=torch.float32, row_scale=inv_r, col_scale=inv_c)
    ref = x.double() @ w.double().t() + b.double()
    bound = (x.double().abs() @ w.double().abs().t()) + b.double().abs() + 1e-300
    if gelu:
        err = (y.double() - torch.nn.functional.gelu(ref)).abs().max().item()
        assert err < 2e-5                                             # the kernel's exact-erf GELU is the A-S 7.1.26 form (1.5e-7) 
    else:
        rel = ((y.double() - ref).abs() / bound).max().item()
        print(f"M={M} N={N} K={K}: {rel:.2e} x sum|x w|")
        assert rel < 1.5e-6
    for m2 in (1, min(M, 257)):
        a2, r2 = ops.split_f16x3(x[:m2].contiguous())
        y2 = ops.dense_gemm(a2, b3, b, gelu=gelu, out_dtype=torch.float32, row_scale=r2, col_scale=inv_c)
        assert torch.equal(y2, y[:m2])


@pytest.mark.parametrize("M,N,K,gelu", [(2500 * 3, 768, 768, False), (980, 4096, 4096, False), (515, 3072, 768, True),
                                        (33, 2304, 768, False), (700, 768, 3072, False), (260, 4096, 11008, False)])
def test_dense_gemm_split_three_products_from_one_staging(M, N, K, gelu):
    """psg_dense_gemm_split (round 6): both operands as interleaved hi / lo fp16 images (psg_split_f16x3 order 2), the
    three products xh.wh + xh.wl + xl.wh formed from ONE staging of each part - the Q-Former's / the row-invariant
    prompt pass's Linear layers in the fp32s mode (HF-IB:519-596, HF-LL:163-177 at V4:99-100's fp32): against fp64 with
    the bound of the K' = 3K form, row-count invariance bit for bit, tile invariance bit for bit, and the layout of the
    split image itself."""
    from openpsg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K + 1)
    x = torch.randn(M, K, generator=g)
    if not gelu:                                    # rows of very different magnitude (the GELU bound below is absolute)
        x = x * torch.exp2(torch.randint(-6, 6, (M, 1), generator=g).float())
    x = x.to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    a2, inv_r = ops.split_f16i2(x)
    b2, inv_c = ops.split_f16i2(w)
    # the image: per 32 k [hi | lo], the same row scale as the three-segment form
    a3, inv3 = ops.split_f16x3(x)
    assert torch.equal(inv_r, inv3)
    blocks = a2.view(M, K // 32, 2, 32)
    assert torch.equal(blocks[:, :, 0].reshape(M, K), a3[:, :K]) and torch.equal(blocks[:, :, 1].reshape(M, K), a3[:, 2 * K:])
    y = ops.dense_gemm_split(a2, b2, b, inv_r, inv_c, gelu=gelu)
    ref = x.double() @ w.double().t() + b.double()
    bound = (x.double().abs() @ w.double().abs().t()) + b.double().abs() + 1e-300
    if gelu:
        err = (y.double() - torch.nn.functional.gelu(ref)).abs().max().item()
        assert err < 2e-5
    else:
        rel = ((y.double() - ref).abs() / bound).max().item()
        print(f"M={M} N={N} K={K}: {rel:.2e} x sum|x w|")
        assert rel < 1.5e-6
    for m2 in (1, min(M, 257)):                                          # a row does not depend on the rows beside it
        s2, r2 = ops.split_f16i2(x[:m2].contiguous())
        assert torch.equal(ops.dense_gemm_split(s2, b2, b, r2, inv_c, gelu=gelu), y[:m2])
    for tile in ("256x256", "256x128", "256x64", "128x128"):             # ... nor on the tile it falls into
        assert torch.equal(ops.dense_gemm_split(a2, b2, b, inv_r, inv_c, gelu=gelu, tile=tile), y), tile


def _g6_decode(dtype, options):
    """G6 through a fresh head built under the given context options; returns (existence logits, tokens, first logits)."""
    import numpy as np
    from openpsg_amd import _lib
    from openpsg_amd.head import RelationTransformerHeadV4
    from tests import helpers as H
    g, cfg, w, scene = H.load_case("G6_llm_7b_width_n6")
    dev = torch.device("cuda:0")
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    sel = torch.from_numpy(g["selected"].astype(np.int32)).to(dev)
    old = {k: _lib.get_option(0, k) for k in options}
    for k, v in options.items():
        _lib.set_option(0, k, v)
    try:
        head = RelationTransformerHeadV4(dtype=dtype, device="cuda:0", qformer_vocab_size=cfg.qformer.vocab, llm_config=cfg.llm,
                                         llm_feature_size=cfg.llm.hidden, tokenizers="word", max_object_num=cfg.max_object_num,
                                         on_parse_error="skip", suppress_eos=bool(g["suppress_eos"]))
        head.load_weights(w)
        rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names, scene["pan_results"].to(dev))
        dec = head.decode_selected(rq, names, selected=sel)
        out = (rq["exist_logit"].cpu().clone(), dec["tokens_host"].copy(), dec["first_logits"].float().cpu().clone())
    finally:
        for k, v in old.items():
            _lib.set_option(0, k, v)
    del head
    torch.cuda.empty_cache()
    return g, out


def test_write_through_output_stores_change_no_value():
    """Option wt_stores (round 6: the fp32 decode chain's outputs stored write-through) is a cache policy: the
    existence logits, first-step logits and all 20 x 16 tokens of G6 are bit-identical with it on and off, in fp32 and fp32s."""
    import numpy as np
    for dtype in ("fp32", "fp32s"):
        _, off = _g6_decode(dtype, {"wt_stores": 0})
        _, on = _g6_decode(dtype, {"wt_stores": 1})
        _, every = _g6_decode(dtype, {"wt_stores": 7})
        for a in (on, every):
            assert torch.equal(off[0], a[0]) and np.array_equal(off[1], a[1]) and torch.equal(off[2], a[2]), dtype


def test_split_products_in_both_operand_layouts_agree_with_the_reference():
    """Option split_i2 (round 6): the Q-Former's fp32s products on psg_dense_gemm_split (interleaved hi / lo operands, three
    products from one staging) against the K' = 3K form of round 5: both inside 1e-3 of the reference's existence logits
    on G6, within 2e-5 of each other, the same selection, and the reference's tokens."""
    import numpy as np
    g, new = _g6_decode("fp32s", {"split_i2": 1})
    _, old = _g6_decode("fp32s", {"split_i2": 0})
    ref = torch.from_numpy(g["exist_logit"])
    e_new, e_old = (new[0] - ref).abs().max().item(), (old[0] - ref).abs().max().item()
    print(f"G6 existence logits vs the reference: split form {e_new:.2e}, K' = 3K form {e_old:.2e}; "
          f"between them {(new[0] - old[0]).abs().max().item():.2e}")
    assert e_new < 1e-3 and e_old < 1e-3 and (new[0] - old[0]).abs().max().item() < 2e-5
    assert np.array_equal(new[1], old[1])
    for i in range(new[1].shape[0]):
        want = g["gen_tokens"][i]
        assert [int(t) for t in new[1][i] if t >= 0] == want[want >= 0].tolist()


def test_fused_split_kernels_equal_the_separate_kernels_bit_for_bit():
    """psg_rmsnorm_split / psg_rope_kvwrite_scaled / psg_silu_mul_split against psg_scale_rows_cols + psg_rmsnorm /
    psg_rope_kvwrite / psg_silu_mul + psg_split_f16x3 on the prompt pass's shapes: identical bits (no GEMM involved)."""
    from openpsg_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(11)
    rows, D, I, heads, ctx = 200, 4096, 11008, 32, 64
    pw2 = lambda n: torch.exp2(torch.randint(-20, 4, (n,), generator=g, device=dev).float())   # noqa: E731
    # RMSNorm
    resid0 = torch.randn(rows, D, generator=g, device=dev)
    w = 1.0 + 0.1 * torch.randn(D, generator=g, device=dev)
    for with_delta in (False, True):
        y, rs, cs = torch.randn(rows, D, generator=g, device=dev) * 1e3, pw2(rows), pw2(D)
        ra, rb = resid0.clone(), resid0.clone()
        n = torch.empty_like(ra)
        ops.rmsnorm(ra, ops.scale_rows_cols(y.clone(), rs, cs) if with_delta else None, w, 1e-5, n)
        a3, inv = ops.split_f16x3(n)
        b3, binv = ops.rmsnorm_split(rb, ops.Scaled(y.clone(), rs, cs) if with_delta else None, w, 1e-5)
        assert torch.equal(ra, rb) and torch.equal(a3, b3) and torch.equal(inv, binv)
    # <= 64 rows of >= 4096 columns (the last layer's kept rows, a 33..64-row prompt pass): psg_rmsnorm spreads a row over
    # 1024 threads there - another sum-of-squares tree - and psg_rmsnorm_split must follow it
    for small in (20, 64):
        y, rs, cs = torch.randn(small, D, generator=g, device=dev) * 1e3, pw2(small), pw2(D)
        ra, rb, rc = resid0[:small].clone(), resid0[:small].clone(), resid0[:small].clone()
        n = torch.empty_like(ra)
        ops.rmsnorm(ra, ops.scale_rows_cols(y.clone(), rs, cs), w, 1e-5, n)
        a3, inv = ops.split_f16x3(n)
        b3, binv = ops.rmsnorm_split(rb, ops.Scaled(y.clone(), rs, cs), w, 1e-5)
        assert torch.equal(ra, rb) and torch.equal(a3, b3) and torch.equal(inv, binv)
        a2, inv2 = ops.split_f16x2(n)
        b2, binv2 = ops.rmsnorm_split(rc, ops.Scaled(y.clone(), rs, cs), w, 1e-5, planes=2)
        assert torch.equal(ra, rc) and torch.equal(a2, b2) and torch.equal(inv2, binv2)
    # ... with the product handed over as three K-segment slices (one batched library GEMM): summed in slice order
    y3, rs, cs = torch.randn(3, rows, D, generator=g, device=dev) * 1e3, pw2(rows), pw2(D)
    ra, rb = resid0.clone(), resid0.clone()
    n = torch.empty_like(ra)
    ops.rmsnorm(ra, ops.scale_rows_cols((y3[0] + y3[1]) + y3[2], rs, cs), w, 1e-5, n)
    a3, inv = ops.split_f16x3(n)
    b3, binv = ops.rmsnorm_split(rb, ops.Scaled(y3.clone(), rs, cs), w, 1e-5)
    assert torch.equal(ra, rb) and torch.equal(a3, b3) and torch.equal(inv, binv)
    assert torch.equal(ops.Scaled(y3.clone(), rs, cs).dense(), ops.scale_rows_cols(y3.sum(0), rs, cs))
    # SwiGLU
    y, rs, cs = torch.randn(rows, 2 * I, generator=g, device=dev) * 1e3, pw2(rows), pw2(2 * I)
    act = torch.empty(rows, I, device=dev)
    ops.silu_mul(ops.scale_rows_cols(y.clone(), rs, cs), act)
    a3, inv = ops.split_f16x3(act)
    b3, binv = ops.silu_mul_split(ops.Scaled(y.clone(), rs, cs), I)
    assert torch.equal(a3, b3) and torch.equal(inv, binv)
    # rotary + cache write
    K, S = 5, 40
    rows = K * S
    y, rs, cs = torch.randn(rows, 3 * D, generator=g, device=dev) * 1e3, pw2(rows), pw2(3 * D)
    inv_f = 1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    ang = torch.arange(ctx, dtype=torch.float32)[:, None] * inv_f[None, :]
    rope = (ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev))
    pos = torch.arange(S, dtype=torch.int32).repeat(K)
    pos[S - 3:S] = -1
    pos = pos.to(dev)
    pair = torch.arange(K, dtype=torch.int32)[:, None].expand(-1, S).reshape(-1).contiguous().to(dev)
    outs = []
    for fused in (False, True):
        q = torch.zeros(rows, D, device=dev)
        kc, vc = torch.zeros(K, heads, ctx, 128, device=dev), torch.zeros(K, heads, ctx, 128, device=dev)
        if fused:
            ops.rope_kvwrite_scaled(ops.Scaled(y.clone(), rs, cs), pair, pos, rope, heads, 128, ctx, q, kc, vc)
        else:
            ops.rope_kvwrite(ops.scale_rows_cols(y.clone(), rs, cs), pair, pos, rope, heads, 128, ctx, q, kc, vc)
        outs.append((q, kc, vc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fp32s_prompt_pass_with_fused_splits_matches_the_separate_kernels():
    """fp32s mode, G6 (Llama-2-7B width, 2 layers): the prompt pass with the operand splits / result scalings inside
    RMSNorm, rotary and SwiGLU against the separate launches: the same greedy tokens (the reference's) and first-step
    logits to 1e-4 (the kernels are bit-identical, previous test; the library GEMM between them is not run-to-run)."""
    import numpy as np
    from openpsg_amd import _lib
    from openpsg_amd.head import RelationTransformerHeadV4
    from tests import helpers as H
    g, cfg, w, scene = H.load_case("G6_llm_7b_width_n6")
    dev = torch.device("cuda:0")
    ids = [int(i) for i in scene["object_id_list"]]
    names = H.object_names(scene)
    sel = torch.from_numpy(g["selected"].astype(np.int32)).to(dev)
    outs = {}
    for flag in (0, 1):
        _lib.set_option(0, "llm_fuse_split", flag)
        try:
            head = RelationTransformerHeadV4(dtype="fp32s", device="cuda:0", qformer_vocab_size=cfg.qformer.vocab,
                                             llm_config=cfg.llm, llm_feature_size=cfg.llm.hidden, tokenizers="word",
                                             max_object_num=cfg.max_object_num, on_parse_error="skip",
                                             suppress_eos=bool(g["suppress_eos"]))
            head.load_weights(w)
            assert head.llm_engine.fuse_split == bool(flag)
            rq = head.run_relation_query(scene["mask_features"].to(dev), scene["img_meta"], ids, names,
                                         scene["pan_results"].to(dev))
            dec = head.decode_selected(rq, names, selected=sel)
            outs[flag] = (dec["tokens_host"].copy(), dec["first_logits"].float().cpu())
        finally:
            _lib.set_option(0, "llm_fuse_split", 1)
        del head
        torch.cuda.empty_cache()
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-4
    assert np.array_equal(outs[0][0], outs[1][0])
    for i in range(outs[1][0].shape[0]):
        want = g["gen_tokens"][i]
        assert [int(t) for t in outs[1][0][i] if t >= 0] == want[want >= 0].tolist()


def test_engine_takes_the_library_gemm_where_the_streaming_kernel_cannot_hold_the_rows():
    """The fp32 weight-streaming kernel keeps the rows' K slice in LDS: 32 rows fit up to K = 11776, 20 rows up to 20480.
    Beyond that psg_skinny_gemm_plan refuses BEFORE anything is launched and the engine's projection falls through to the
    library GEMM (dense result) instead of raising - a model wider than Llama-2-13B still decodes."""
    from openpsg_amd import _lib, ops
    from openpsg_amd.config import PSGConfig, QFormerConfig, tiny_llm
    from openpsg_amd.llm import LlamaDecodeEngine
    from openpsg_amd.weights import make_weights_numpy
    dev = _dev()
    assert ops.skinny_gemm_plan(32, 4096, 11008, torch.float32, dev) == 16 and ops.skinny_gemm_plan(20, 4096, 20480, torch.float32, dev) == 16
    with pytest.raises(_lib.PsgHipError, match="do not fit the LDS"):
        ops.skinny_gemm_plan(32, 4096, 12288, torch.float32, dev)
    with pytest.raises(_lib.PsgHipError, match="do not fit the LDS"):
        ops.skinny_gemm_plan(21, 4096, 20480, torch.float32, dev)
    cfg = PSGConfig(qformer=QFormerConfig(vocab=512), llm=tiny_llm(256, 1, 512, 512), max_object_num=30)
    w = make_weights_numpy(cfg, seed=3)
    eng = LlamaDecodeEngine(w, cfg, dev, torch.float32)
    g = torch.Generator().manual_seed(2)
    wide = (torch.randn(256, 12288, generator=g) / 110.0).to(dev)
    x = torch.randn(32, 12288, generator=g).to(dev)
    y32 = eng.linear(x, wide, decode=True)
    assert torch.is_tensor(y32) and torch.equal(y32, torch.nn.functional.linear(x, wide))     # the library's own result
    y20 = eng.linear(x[:20].contiguous(), wide, decode=True)
    assert isinstance(y20, ops.Partials)                                                     # 20 rows still stream
    ref = x[:20].double() @ wide.double().t()
    assert ((y20.reduce(torch.float32).double() - ref).abs() <= 4e-7 * (x[:20].double().abs() @ wide.double().abs().t())).all()
