"""psg_batch_gemm (`-m gpu`): the decode-step projections for 33..160 rows (several images' selected pairs decoded
together, `head.forward_batch`) against a plain fp32 product of the same 16-bit operands - every Llama-2-7B projection
shape, ragged shapes (N not a multiple of the slab, few K steps, fewer units than workgroups), both slab heights, slab-aligned
and stream-K ranges - and
against psg_skinny_gemm on 32 of the rows (HF-LL q/k/v/o/gate/up/down and lm_head: bias-free Linear layers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(M, N, K, dtype, bn=0, seed=0, mode=0):
    from openpsg_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(seed + M + N + K)
    x = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).to(dtype)
    _lib.set_option(0, "batch_gemm_bn", bn)
    _lib.set_option(0, "batch_gemm_mode", mode)
    try:
        part = ops.batch_gemm(x, w)
    finally:
        _lib.set_option(0, "batch_gemm_bn", 0)
        _lib.set_option(0, "batch_gemm_mode", 0)
    assert part.t.shape[1:] == (M, N) and torch.isfinite(part.t).all()
    got = part.t.sum(0)
    ref = x.double() @ w.double().t()
    err = (got.double() - ref).abs().max().item()
    # exact products of 16-bit operands, fp32 accumulation over K terms of magnitude ~ 1 / sqrt(K)
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), f"M={M} N={N} K={K} bn={bn}: max error {err:.3e}, slots {part.splits}"
    assert torch.equal(part.reduce(torch.float32), got) or (part.reduce(torch.float32) - got).abs().max() < 1e-5
    return part


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [40, 80, 160])
def test_llama_projection_shapes(M, dtype):
    for N, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096)):
        _check(M, N, K, dtype)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("bn", [128, 256])
@pytest.mark.parametrize("M", [33, 64, 65, 96, 97, 129, 159])
def test_ragged_shapes_both_slab_heights_aligned_and_stream_k_ranges(M, bn, mode):
    for N, K in ((272, 320), (4096, 64), (1040, 4096), (16, 128), (22016, 256), (12288, 4096)):
        _check(M, N, K, torch.float16, bn=bn, seed=7, mode=mode)


def test_rows_do_not_depend_on_their_neighbours_and_agree_with_the_skinny_kernel():
    from openpsg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 160, 4096, 4096
    x = torch.randn(M, K, generator=g, device=DEV).half()
    w = (torch.randn(N, K, generator=g, device=DEV) / 64).half()
    a = ops.batch_gemm(x, w).t.sum(0)
    x2 = x.clone()
    x2[32:] = torch.randn(M - 32, K, generator=g, device=DEV).half()
    b = ops.batch_gemm(x2, w).t.sum(0)
    assert torch.equal(a[:32], b[:32]), "a row's result changed with the other rows of the call"
    s = ops.skinny_gemm(x[:32].contiguous(), w).t.sum(0)
    assert (a[:32] - s).abs().max().item() < 2e-5 * max(1.0, s.abs().max().item())


def test_rejects_what_it_is_not_built_for():
    from openpsg_amd import ops
    from openpsg_amd._lib import PsgHipError
    x = torch.zeros(161, 128, device=DEV, dtype=torch.float16)
    w = torch.zeros(64, 128, device=DEV, dtype=torch.float16)
    with pytest.raises(PsgHipError):
        ops.batch_gemm(x, w)
    with pytest.raises(PsgHipError):
        ops.batch_gemm(x[:64].float(), w.float())
    with pytest.raises(PsgHipError):
        ops.batch_gemm(x[:64, :96].contiguous(), w[:, :96].contiguous())


def test_forward_batch_on_the_planned_projections_reproduces_the_per_image_decodes():
    """Three copies of the G6 scene (Llama-2-7B width, 2 layers, 20 selected pairs each = 60 decode rows per step) through
    `head.forward_batch` in the mixed mode with psg_batch_gemm available: the three copies decode to the SAME tokens (a
    row's projections do not depend on its position in the batch), those are the tokens of the single-image `forward`
    (<= 32 rows: psg_skinny_gemm) up to fp32 summation order (>= 95 %), and at least one projection shape was planned
    onto psg_batch_gemm (V4:293-312 decodes pair by pair; HF-LL:53-281)."""
    import numpy as np
    from openpsg_amd import llm as llm_mod
    from openpsg_amd.head import RelationTransformerHeadV4
    from tests import helpers as H
    g, cfg, w, scene = H.load_case("G6_llm_7b_width_n6")
    head = RelationTransformerHeadV4(dtype="mixed", device=DEV, qformer_vocab_size=cfg.qformer.vocab, llm_config=cfg.llm,
                                     llm_feature_size=cfg.llm.hidden, tokenizers="word", max_object_num=cfg.max_object_num,
                                     on_parse_error="skip", suppress_eos=True)
    head.load_weights(w)
    assert head.llm_engine.batch_gemm
    dev = torch.device(DEV)
    inp = dict(mask_features=scene["mask_features"].to(dev), img_metas=[scene["img_meta"]],
               object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"].to(dev))])
    single = head(inp)
    tok1 = head.last["tokens_host"].copy()
    batched = head.forward_batch([inp, inp, inp])
    toks = [head.last_batch[i]["tokens_host"] for i in range(3)]
    assert np.array_equal(toks[0], toks[1]) and np.array_equal(toks[0], toks[2])
    assert batched[0] == batched[1] == batched[2]
    same = int((toks[0] == tok1).sum())
    assert same >= 0.95 * tok1.size, f"{same} of {tok1.size} tokens equal the single-image decode"
    # 60 rows: the fixed plan table (llm._BATCH_PLAN_TABLE) puts the down projection on psg_batch_gemm
    eng = head.llm_engine
    x60 = torch.zeros((60, cfg.llm.inter), device=dev, dtype=eng.dtype)
    assert llm_mod._plan_batch_mm(x60, [eng.layers[0]["wdown"]])[0] == "own"
    again = head.forward_batch([inp, inp, inp])                        # graph replay
    assert again == batched and single is not None
