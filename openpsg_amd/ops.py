"""Tensor-level wrappers over the C ABI (include/psg_hip.h).

torch is plumbing here: it owns the HBM allocations and the stream; every wrapper passes raw
device pointers + sizes to libpsg_hip.so on `torch.cuda.current_stream()`.  There is no host
fallback: tensors must be CUDA(HIP) tensors, contiguous, of the dtype the kernel expects.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import PSG_PRO_RMSNORM, PSG_PRO_DECODE_ATTN, PSG_PRO_SILU_MUL  # noqa: F401
from ._lib import (PSG_BF16, PSG_F16, PSG_F32, PSG_EMPTY_UNIFORM, PSG_EMPTY_UNMASKED, PSG_XATTN_MFMA,
                   PSG_XATTN_SIMPLE, PsgHipError, check)

_DT = {torch.float32: PSG_F32, torch.bfloat16: PSG_BF16, torch.float16: PSG_F16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise PsgHipError(f"unsupported activation dtype {t.dtype}") from None


def _p(t, dtype=None, name="tensor"):
    """device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PsgHipError(f"{name} must live in HBM (got a {t.device} tensor); this path has no CPU fallback")
    if not t.is_contiguous():
        raise PsgHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise PsgHipError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def _env(t: torch.Tensor):
    if not t.is_cuda:
        raise PsgHipError(f"tensor lives on {t.device}: this path runs only on the GPU (no CPU fallback)")
    return _lib.load(), _lib.ctx(t.device.index or 0), torch.cuda.current_stream(t.device).cuda_stream


def patch_embed(feat: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, patch: int = 16) -> torch.Tensor:
    """feat [1,C,Hf,Wf] fp32, weight [Cout,C,16,16] fp32 -> patches [L, Cout] fp32 (V4:410)."""
    import ctypes
    lib, ctx, st = _env(feat)
    _, Cc, Hf, Wf = feat.shape
    Cout = weight.shape[0]
    nbytes = ctypes.c_int64(0)
    check(lib.psg_patch_embed_workspace(ctx, Cc, Hf, Wf, Cout, patch, ctypes.byref(nbytes)),
          "psg_patch_embed_workspace")
    ws = torch.empty(nbytes.value // 4, device=feat.device, dtype=torch.float32)
    out = torch.empty(((Hf // patch) * (Wf // patch), Cout), device=feat.device, dtype=torch.float32)
    check(lib.psg_patch_embed(ctx, _p(feat, torch.float32, "mask_features"), Cc, Hf, Wf,
                              _p(weight, torch.float32, "patch_embed.weight"), _p(bias, torch.float32), Cout, patch,
                              _p(out), _p(ws), nbytes.value, st), "psg_patch_embed")
    return out


def mask_grid(pan: torch.Tensor, img_hw, pad_hw, grid_hw) -> torch.Tensor:
    lib, ctx, st = _env(pan)
    gh, gw = int(grid_hw[0]), int(grid_hw[1])
    out = torch.empty(gh * gw, device=pan.device, dtype=torch.float32)
    check(lib.psg_mask_grid(ctx, _p(pan, torch.int32, "pan_results"), pan.shape[0], pan.shape[1], int(img_hw[0]),
                            int(img_hw[1]), int(pad_hw[0]), int(pad_hw[1]), gh, gw, _p(out), st), "psg_mask_grid")
    return out


def object_bitmasks(grid: torch.Tensor, object_ids: torch.Tensor) -> torch.Tensor:
    """-> int64 view of uint64 bits [N, ceil(L/64)]."""
    lib, ctx, st = _env(grid)
    L, N = grid.numel(), object_ids.numel()
    words = (L + 63) // 64
    bits = torch.empty((N, words), device=grid.device, dtype=torch.int64)
    check(lib.psg_object_bitmasks(ctx, _p(grid, torch.float32, "grid"), L, _p(object_ids, torch.int32, "object_ids"), N,
                                  _p(bits), words, st), "psg_object_bitmasks")
    return bits


def qformer_embed(ids, word_emb, pos_emb, query_rows, ln_w, ln_b, eps, out):
    lib, ctx, st = _env(out)
    B, T = ids.shape
    nq, hidden = query_rows.shape
    assert out.shape == (B * (nq + T), hidden)
    check(lib.psg_qformer_embed(ctx, _p(ids, torch.int32, "ids"), B, T, _p(word_emb, torch.float32),
                                _p(pos_emb, torch.float32), _p(query_rows, torch.float32), nq,
                                _p(ln_w, torch.float32), _p(ln_b, torch.float32), float(eps), hidden, _p(out), _dt(out),
                                st), "psg_qformer_embed")
    return out


def qformer_embed_split(ids, word_emb, pos_emb, query_rows, ln_w, ln_b, eps, out_query, out_text):
    """The same embedding, written as ONE block of query rows [nq, hidden] (identical for every pair) and the
    text rows [B*T, hidden]."""
    lib, ctx, st = _env(out_text)
    B, T = ids.shape
    nq, hidden = query_rows.shape
    assert out_query.shape == (nq, hidden) and out_text.shape == (B * T, hidden)
    common = (_p(word_emb, torch.float32), _p(pos_emb, torch.float32), _p(query_rows, torch.float32))
    tail = (_p(ln_w, torch.float32), _p(ln_b, torch.float32), float(eps), hidden)
    check(lib.psg_qformer_embed(ctx, None, 1, 0, *common, nq, *tail, _p(out_query), _dt(out_query), st),
          "psg_qformer_embed")
    check(lib.psg_qformer_embed(ctx, _p(ids, torch.int32, "ids"), B, T, *common, 0, *tail, _p(out_text), _dt(out_text),
                                st), "psg_qformer_embed")


def add_layernorm(x, residual, bias, gamma, beta, eps, out=None):
    lib, ctx, st = _env(x)
    out = x if out is None else out
    rows, hidden = x.shape
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
    check(lib.psg_add_layernorm(ctx, _p(x), _p(residual), _p(bias, torch.float32), _p(gamma, torch.float32),
                                _p(beta, torch.float32), float(eps), rows, hidden, _p(out, x.dtype), _dt(x), st),
          "psg_add_layernorm")
    return out


def add_layernorm_periodic(x, residual_table, bias, gamma, beta, eps, out=None):
    """LayerNorm(x + bias + residual_table[row % table_rows]); in place unless `out` is given."""
    lib, ctx, st = _env(x)
    out = x if out is None else out
    rows, hidden = x.shape
    assert residual_table.shape[1] == hidden and residual_table.dtype == x.dtype
    check(lib.psg_add_layernorm_periodic(ctx, _p(x), _p(residual_table), residual_table.shape[0],
                                         _p(bias, torch.float32), _p(gamma, torch.float32), _p(beta, torch.float32),
                                         float(eps), rows, hidden, _p(out, x.dtype), _dt(x), st),
          "psg_add_layernorm_periodic")
    return out


def add_layernorm_indexed(x, residual_table, block_index, group, bias, gamma, beta, eps, out=None):
    """LayerNorm(x + bias + residual_table[block_index[row // group] * group + row % group]); in place unless `out`."""
    lib, ctx, st = _env(x)
    out = x if out is None else out
    rows, hidden = x.shape
    assert residual_table.shape[1] == hidden and residual_table.dtype == x.dtype and rows % group == 0
    assert block_index.numel() == rows // group
    check(lib.psg_add_layernorm_indexed(ctx, _p(x), _p(residual_table), _p(block_index, torch.int32, "block_index"), group,
                                        _p(bias, torch.float32), _p(gamma, torch.float32), _p(beta, torch.float32),
                                        float(eps), rows, hidden, _p(out, x.dtype), _dt(x), st),
          "psg_add_layernorm_indexed")
    return out


def add_layernorm_res32(x, residual32, bias, gamma, beta, eps, out16=None, out32=None, period=0, index=None,
                        want16=True, want32=True):
    """Mixed mode: LayerNorm(x (16-bit) + bias + residual32 (fp32; plain, `period`-periodic table, or table blocks chosen
    by `index` per group of `period` rows)) -> (16-bit copy in place of x unless out16 is given, fp32 copy)."""
    lib, ctx, st = _env(x)
    rows, hidden = x.shape
    if want16 and out16 is None:
        out16 = x
    if want32 and out32 is None:
        out32 = torch.empty((rows, hidden), device=x.device, dtype=torch.float32)
    if residual32 is not None:
        assert residual32.dtype == torch.float32 and residual32.shape[1] == hidden
        assert period > 0 or residual32.shape[0] == rows
    check(lib.psg_add_layernorm_res32(ctx, _p(x), _p(residual32, torch.float32, "residual"), int(period),
                                      _p(index, torch.int32, "index"), _p(bias, torch.float32), _p(gamma, torch.float32),
                                      _p(beta, torch.float32), float(eps), rows, hidden,
                                      _p(out16, x.dtype) if want16 else None, _p(out32, torch.float32) if want32 else None,
                                      _dt(x), st), "psg_add_layernorm_res32")
    return (out16 if want16 else None), (out32 if want32 else None)


def bias_gelu(x, bias=None, out=None):
    lib, ctx, st = _env(x)
    out = x if out is None else out
    rows, cols = x.shape
    check(lib.psg_bias_gelu(ctx, _p(x), _p(bias, torch.float32), rows, cols, _p(out, x.dtype), _dt(x), st),
          "psg_bias_gelu")
    return out


def qformer_self_attn(qkv, text_mask, B, T, nq, heads, query_rows_only, out):
    lib, ctx, st = _env(qkv)
    hidden = qkv.shape[1] // 3
    assert qkv.shape[0] == B * (nq + T) and out.shape == (B * (nq + T), hidden) and out.dtype == qkv.dtype
    check(lib.psg_qformer_self_attn(ctx, _p(qkv), _p(text_mask, torch.uint8, "text_mask"), B, T, nq, heads,
                                    1 if query_rows_only else 0, _p(out), _dt(qkv), st), "psg_qformer_self_attn")
    return out


def qformer_self_attn_cls(q_cls, kv, text_mask, B, T, nq, heads):
    """Attention of the cls row (row 0) of every pair over its nq + T keys -> [B, hidden] (one row per pair).
    q_cls [B, hidden]: projected cls queries; kv [B*(nq+T), 2*hidden]: K | V of every row."""
    lib, ctx, st = _env(kv)
    hidden = kv.shape[1] // 2
    assert kv.shape[0] == B * (nq + T) and q_cls.shape == (B, hidden) and q_cls.dtype == kv.dtype
    out = torch.empty((B, hidden), device=kv.device, dtype=kv.dtype)
    check(lib.psg_qformer_self_attn_cls(ctx, _p(q_cls), _p(kv), _p(text_mask, torch.uint8, "text_mask"), B, T, nq,
                                        heads, _p(out), _dt(kv), st), "psg_qformer_self_attn_cls")
    return out


def qformer_cls_attn_input(x, g, text_mask, B, T, nq, heads, x_text=None, text_index=None):
    """cls-row attention in the input space (psg_qformer_cls_attn_input).  x [B*(nq+T), hidden]: the layer's input rows
    (query rows, then text rows) - or, with x_text [U*T, hidden] / text_index int32 [B], x [B*nq, hidden] holds the
    query rows only and pair p reads the text block (and text_mask row) text_index[p].
    g fp32 [heads, B, hidden] = W_k,h^T q_h  ->  xbar fp32 [heads, B, hidden] = sum_j p_j x_j."""
    lib, ctx, st = _env(x)
    hidden = x.shape[1]
    assert g.shape == (heads, B, hidden) and g.is_contiguous()
    if x_text is None:
        assert x.shape[0] == B * (nq + T) and text_index is None
        xt = x[B * nq:]
    else:
        assert x.shape[0] >= B * nq and x_text.dtype == x.dtype and x_text.shape[1] == hidden and text_index is not None
        xt = x_text
    xbar = torch.empty((heads, B, hidden), device=x.device, dtype=torch.float32)
    check(lib.psg_qformer_cls_attn_input(ctx, _p(x), _p(xt) if T > 0 else None,
                                         _p(text_index, torch.int32, "text_index") if text_index is not None else None,
                                         _p(g, torch.float32, "g"), _p(text_mask, torch.uint8, "text_mask"), B, T, nq,
                                         heads, hidden, _p(xbar), _dt(x), st), "psg_qformer_cls_attn_input")
    return xbar


def qformer_self_attn_shared(qkv_query, qkv_text, text_mask, B, T, nq, heads, out):
    """Layer-0 self-attention with ONE shared [nq, 3*hidden] projection of the query rows."""
    lib, ctx, st = _env(qkv_query)
    hidden = qkv_query.shape[1] // 3
    assert qkv_query.shape == (nq, 3 * hidden) and qkv_text.shape == (B * T, 3 * hidden)
    assert out.shape == (B * (nq + T), hidden) and out.dtype == qkv_query.dtype == qkv_text.dtype
    check(lib.psg_qformer_self_attn_shared(ctx, _p(qkv_query, name="qkv_query"), _p(qkv_text),
                                           _p(text_mask, torch.uint8, "text_mask"), B, T, nq, heads, _p(out),
                                           _dt(qkv_query), st), "psg_qformer_self_attn_shared")
    return out


def qformer_cross_attn(q, k, v, bits, pair_index, N, nq, heads, out=None, empty_policy=PSG_EMPTY_UNIFORM,
                       variant=None):
    lib, ctx, st = _env(q)
    P = pair_index.numel()
    L, hidden = k.shape
    assert q.shape == (P * nq, hidden) and v.shape == k.shape and k.dtype == q.dtype == v.dtype
    if variant is None:
        # the matrix-core kernel keeps all keys of one head in LDS (L <= 512 patches = images up to ~1450 px);
        # larger inputs take the row kernel (the reference runs them too)
        variant = PSG_XATTN_MFMA if L <= 512 else PSG_XATTN_SIMPLE
    out = torch.empty_like(q) if out is None else out
    check(lib.psg_qformer_cross_attn(ctx, _p(q), _p(k), _p(v), _p(bits, torch.int64, "bits"), bits.shape[1],
                                     _p(pair_index, torch.int32, "pair_index"), int(N), P, L, nq, heads,
                                     int(empty_policy), int(variant), _p(out, q.dtype), _dt(q), st),
          "psg_qformer_cross_attn")
    return out


def qformer_cross_attn_indexed(q_u, q_index, q_cls, k, v, bits, pair_index, N, heads, out=None,
                               empty_policy=PSG_EMPTY_UNIFORM):
    """`qformer_cross_attn` (33 query rows per pair) with the queries stored once per prompt: q_u [U * 33, hidden], q_index
    int32 [P] (prompt of each pair), q_cls [P, hidden] (row 0 of every pair).  Returns the context [P * 33, hidden], or
    None when the LDS-DMA kernel cannot run on these inputs (the caller then expands q)."""
    PSG_ERR_UNSUPPORTED = -2                                # include/psg_hip.h
    lib, ctx, st = _env(q_u)
    P = pair_index.numel()
    L, hidden = k.shape
    assert q_cls.shape == (P, hidden) and q_index.numel() == P and q_u.shape[1] == hidden and q_u.dtype == k.dtype == v.dtype
    out = torch.empty((P * 33, hidden), device=q_u.device, dtype=q_u.dtype) if out is None else out
    rc = lib.psg_qformer_cross_attn_indexed(ctx, _p(q_u), _p(q_index, torch.int32, "q_index"), _p(q_cls, q_u.dtype),
                                            _p(k), _p(v), _p(bits, torch.int64, "bits"), bits.shape[1],
                                            _p(pair_index, torch.int32, "pair_index"), int(N), P, L, heads, int(empty_policy),
                                            _p(out, q_u.dtype), _dt(q_u), st)
    if rc == PSG_ERR_UNSUPPORTED:
        return None
    check(rc, "psg_qformer_cross_attn_indexed")
    return out


def exist_head(x, w, b, P, nq):
    lib, ctx, st = _env(x)
    hidden = x.shape[1]
    logit = torch.empty(P, device=x.device, dtype=torch.float32)
    prob = torch.empty(P, device=x.device, dtype=torch.float32)
    check(lib.psg_exist_head(ctx, _p(x), _p(w, torch.float32), _p(b, torch.float32), P, nq, hidden, _p(logit),
                             _p(prob), _dt(x), st), "psg_exist_head")
    return logit, prob


def topk(score, k):
    lib, ctx, st = _env(score)
    idx = torch.empty(k, device=score.device, dtype=torch.int32)
    val = torch.empty(k, device=score.device, dtype=torch.float32)
    check(lib.psg_topk(ctx, _p(score, torch.float32, "score"), score.numel(), k, _p(idx), _p(val), st), "psg_topk")
    return idx, val


def gather_rows(src, idx, dst):
    """dst[r] = src[idx[r]] (rows; idx < 0 -> zeros).  src/dst may be row-strided 2-D views."""
    lib, ctx, st = _env(src)
    assert src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    if not src.is_cuda or not dst.is_cuda:
        raise PsgHipError("gather_rows: tensors must live in HBM")
    n, cols = dst.shape
    assert idx.numel() == n and src.shape[1] == cols
    check(lib.psg_gather_rows(ctx, src.data_ptr(), _dt(src), _p(idx, torch.int32, "idx"), n, cols, src.stride(0),
                              dst.data_ptr(), _dt(dst), dst.stride(0), st), "psg_gather_rows")
    return dst


def gather_pair_rows(xq, xt, text_index, text_mask, pair_index, sel, first, count, slot_off, nq, T, want_aux=True):
    """Rows of the selected pairs (global ids `sel`, int32 [K]) out of a selection-phase pass: see psg_gather_pair_rows.
    Returns (rows [K*(nq+T), cols], text mask [K, T] uint8, pair ids [K] int32, mine [K] uint8) - the last three None
    when want_aux is False (a second table of the same pass, e.g. the fp32 twins)."""
    lib, ctx, st = _env(xq)
    K = sel.numel()
    cols = xq.shape[1]
    out = torch.empty((K * (nq + T), cols), device=xq.device, dtype=xq.dtype)
    tm = torch.empty((K, T), device=xq.device, dtype=torch.uint8) if want_aux else None
    pi = torch.empty(K, device=xq.device, dtype=torch.int32) if want_aux else None
    mine = torch.empty(K, device=xq.device, dtype=torch.uint8) if want_aux else None
    check(lib.psg_gather_pair_rows(ctx, _p(xq), _p(xt, xq.dtype) if T > 0 else None, _p(text_index, torch.int32, "text_index"),
                                   _p(text_mask, torch.uint8, "text_mask") if want_aux and T > 0 else None,
                                   _p(pair_index, torch.int32, "pair_index") if want_aux else None,
                                   _p(sel, torch.int32, "sel"), K, int(first), int(count), int(slot_off), int(nq), int(T), cols,
                                   _p(out), _p(tm) if T > 0 else None, _p(pi), _p(mine), _dt(xq), st), "psg_gather_pair_rows")
    return out, tm, pi, mine


class Partials:
    """fp32 split-K slices [S, rows, cols] written by `skinny_gemm`; the consumer kernels
    (rmsnorm, rope_kvwrite, silu_mul, greedy_step) sum them while loading."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        assert t.dim() == 3 and t.dtype == torch.float32 and t.is_contiguous()
        self.t = t

    @property
    def splits(self):
        return self.t.shape[0]

    @property
    def shape(self):
        return self.t.shape[1:]

    def reduce(self, dtype=torch.bfloat16):
        lib, ctx, st = _env(self.t)
        y = torch.empty(self.shape, device=self.t.device, dtype=dtype)
        check(lib.psg_reduce_partials(ctx, _p(self.t), self.splits, y.numel(), _p(y), _DT[dtype], st),
              "psg_reduce_partials")
        return y


def _in(x, dtype):
    """(pointer, splits) of a kernel input that is either an activation tensor or split-K partials."""
    if isinstance(x, Partials):
        return _p(x.t, torch.float32), x.splits
    return _p(x, dtype), 0


def rmsnorm(resid, delta, w, eps, out):
    """resid += delta; out = RMSNorm(resid) * w.  `out` has the activation dtype; `resid` has it too, or is fp32 with
    16-bit activations (mixed mode: the residual stream is never rounded to 16 bits)."""
    lib, ctx, st = _env(resid)
    rows, hidden = resid.shape
    dp, ds = (None, 0) if delta is None else _in(delta, out.dtype)
    if resid.dtype != out.dtype and resid.dtype != torch.float32:
        raise PsgHipError(f"rmsnorm: residual stream must be {out.dtype} or float32, got {resid.dtype}")
    check(lib.psg_rmsnorm(ctx, _p(resid), dp, ds, _p(w, torch.float32), float(eps), rows, hidden,
                          _p(out), _dt(out), _dt(resid), st), "psg_rmsnorm")
    return out


def rope_kvwrite(qkv, tok_pair, tok_pos, rope, heads, head_dim, ctx_len, q_out, k_cache, v_cache, rope_pos=None):
    """rope = (cos, sin) fp32 tables [>= ctx_len, head_dim/2].  rope_pos int32 [rows]: rotary positions when they
    differ from the cache slots (training forward)."""
    lib, ctx, st = _env(q_out)
    rows = q_out.shape[0]
    qp, qs = _in(qkv, q_out.dtype)
    assert rope[0].shape[0] >= ctx_len and rope[0].shape[1] == head_dim // 2
    if rope_pos is not None:
        # positions of the reference's PADDED sequence may exceed the compact context length: the kernel indexes the
        # rotary tables with them unchecked, so bound them here (training forward only; one scalar read-back)
        top = int(rope_pos.max().item()) if rope_pos.numel() else -1
        if top >= rope[0].shape[0]:
            raise PsgHipError(f"rope_kvwrite: rotary position {top} exceeds the {rope[0].shape[0]}-row rotary table")
    check(lib.psg_rope_kvwrite(ctx, qp, qs, _p(tok_pair, torch.int32), _p(tok_pos, torch.int32),
                               _p(rope_pos, torch.int32), _p(rope[0], torch.float32), _p(rope[1], torch.float32), rows, heads, head_dim, ctx_len, _p(q_out),
                               _p(k_cache, q_out.dtype), _p(v_cache, q_out.dtype), _dt(q_out), st), "psg_rope_kvwrite")
    return q_out


def llm_attn(q, k_cache, v_cache, tok_pair, tok_pos, heads, head_dim, ctx_len, out):
    lib, ctx, st = _env(q)
    check(lib.psg_llm_attn(ctx, _p(q), _p(k_cache, q.dtype), _p(v_cache, q.dtype), _p(tok_pair, torch.int32),
                           _p(tok_pos, torch.int32), q.shape[0], heads, head_dim, ctx_len, _p(out, q.dtype), _dt(q),
                           st), "psg_llm_attn")
    return out


def prefill_attn(q, k_cache, v_cache, tok_pos, pairs, rows_per_pair, heads, head_dim, ctx_len, out):
    """Causal attention of a pair-major prompt batch on the matrix cores (bf16 / fp16 / exact fp32, rows_per_pair <= 64)."""
    lib, ctx, st = _env(q)
    assert q.shape[0] == pairs * rows_per_pair
    check(lib.psg_prefill_attn(ctx, _p(q, name="q"), _p(k_cache, q.dtype), _p(v_cache, q.dtype),
                               _p(tok_pos, torch.int32), pairs, rows_per_pair, heads, head_dim, ctx_len,
                               _p(out, q.dtype), _dt(q), st), "psg_prefill_attn")
    return out


def prefill_attn_rope(qkv, tok_pos, rope, pairs, rows_per_pair, heads, head_dim, ctx_len, k_cache, v_cache, out):
    """Rotary + KV-cache write + causal attention of a pair-major prompt batch in one launch (bf16 / fp16)."""
    lib, ctx, st = _env(out)
    assert qkv.shape == (pairs * rows_per_pair, 3 * heads * head_dim) and rope[0].shape[1] == head_dim // 2
    assert rope[0].shape[0] >= rows_per_pair
    check(lib.psg_prefill_attn_rope(ctx, _p(qkv, out.dtype, "qkv"), _p(tok_pos, torch.int32),
                                    _p(rope[0], torch.float32), _p(rope[1], torch.float32), pairs, rows_per_pair, heads,
                                    head_dim, ctx_len, _p(k_cache, out.dtype), _p(v_cache, out.dtype), _p(out),
                                    _dt(out), st), "psg_prefill_attn_rope")
    return out


def decode_attn(qkv, tok_pair, tok_pos, rope, heads, head_dim, ctx_len, k_cache, v_cache, out):
    """Fused rotary + KV append + attention for rows that each hold the newest token of their pair."""
    lib, ctx, st = _env(out)
    qp, qs = _in(qkv, out.dtype)
    assert rope[0].shape[0] >= ctx_len and rope[0].shape[1] == head_dim // 2
    check(lib.psg_decode_attn(ctx, qp, qs, _p(tok_pair, torch.int32), _p(tok_pos, torch.int32),
                              _p(rope[0], torch.float32), _p(rope[1], torch.float32), out.shape[0], heads, head_dim, ctx_len,
                              _p(k_cache, out.dtype), _p(v_cache, out.dtype), _p(out), _dt(out), st), "psg_decode_attn")
    return out


def decode_layer_supported(rows, hidden, inter, heads, dtype, device) -> bool:
    """Does psg_decode_layer (one persistent launch per decoder layer of the decode step) take this shape?"""
    if dtype not in _DT or not torch.device(device).type == "cuda":
        return False
    dev = torch.device(device)
    lib, ctx = _lib.load(), _lib.ctx(dev.index or 0)
    return bool(lib.psg_decode_layer_supported(ctx, int(rows), int(hidden), int(inter), int(heads), _DT[dtype]))


DL_TIMEOUT_WORD = 255 * 64         # csrc/psg_decode_layer.hip: PSG_DL_TIMEOUT * PSG_DL_SLOT - set by a hand-off poll that gave up


def decode_layer_counters(device) -> int:
    """int32 words of one psg_decode_layer launch's counter block (to be zeroed by the caller)."""
    import ctypes
    dev = torch.device(device)
    lib, ctx = _lib.load(), _lib.ctx(dev.index or 0)
    nf, nc = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.psg_decode_layer_workspace(ctx, 16, 4096, 11008, ctypes.byref(nf), ctypes.byref(nc)),
          "psg_decode_layer_workspace")
    return int(nc.value)


def decode_layer_workspace(rows, hidden, inter, device):
    """(workspace fp32 tensor, counter words per launch) of psg_decode_layer for `rows` decode rows."""
    import ctypes
    dev = torch.device(device)
    lib, ctx = _lib.load(), _lib.ctx(dev.index or 0)
    nf, nc = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.psg_decode_layer_workspace(ctx, int(rows), int(hidden), int(inter), ctypes.byref(nf), ctypes.byref(nc)),
          "psg_decode_layer_workspace")
    return torch.empty(nf.value, device=dev, dtype=torch.float32), int(nc.value)


def decode_layer(resid, delta, ln1, ln2, wqkv, wo, wgu, wdown, tok_pair, tok_pos, rope, heads, ctx_len, eps, k_cache,
                 v_cache, workspace, counters, down_part) -> Partials:
    """One decoder layer of the decode step in ONE persistent launch (psg_decode_layer), bit-identical to
    rmsnorm -> skinny_gemm -> decode_attn -> skinny_gemm -> rmsnorm -> skinny_gemm -> silu_mul -> skinny_gemm.
    resid [M, hidden] fp32 (updated in place); delta: the previous layer's down-projection Partials or None;
    counters: int32 [decode_layer_counters()], ZERO (word 255 * 64 != 0 afterwards: a bounded poll gave up); down_part fp32 [16, M, hidden] receives this layer's down partials (returned)."""
    lib, ctx, st = _env(resid)
    M, hidden = resid.shape
    inter = wdown.shape[1]
    assert wqkv.shape == (3 * hidden, hidden) and wo.shape == (hidden, hidden) and wgu.shape == (2 * inter, hidden)
    assert down_part.shape == (16, M, hidden) and counters.dtype == torch.int32
    assert counters.numel() >= decode_layer_counters(resid.device)
    dp, ds = (None, 0) if delta is None else _in(delta, resid.dtype)
    check(lib.psg_decode_layer(ctx, _p(resid, torch.float32, "resid"), dp, ds, _p(ln1, torch.float32), _p(ln2, torch.float32),
                               _p(wqkv, resid.dtype, "wqkv"), _p(wo, resid.dtype), _p(wgu, resid.dtype), _p(wdown, resid.dtype),
                               _p(tok_pair, torch.int32), _p(tok_pos, torch.int32), _p(rope[0], torch.float32),
                               _p(rope[1], torch.float32), M, hidden, inter, int(heads), int(ctx_len), float(eps),
                               _p(k_cache, resid.dtype), _p(v_cache, resid.dtype), _p(workspace, torch.float32),
                               _p(counters), _p(down_part, torch.float32), _dt(resid), st), "psg_decode_layer")
    return Partials(down_part)


def decode_layer_table(layers, k_caches, v_caches, device=None):
    """Table for `decode_layers`: one row of 8 pointers per decoder layer (the tensors must outlive it); on the layers'
    device unless `device` says otherwise (a caller inside a graph capture stages it through pinned memory)."""
    rows = [[L["ln1"].data_ptr(), L["ln2"].data_ptr(), L["wqkv"].data_ptr(), L["wo"].data_ptr(), L["wgu"].data_ptr(),
             L["wdown"].data_ptr(), k.data_ptr(), v.data_ptr()] for L, k, v in zip(layers, k_caches, v_caches)]
    return torch.tensor(rows, dtype=torch.int64, device=layers[0]["wqkv"].device if device is None else device)


def decode_layers(resid, delta, table, n_layers, tok_pair, tok_pos, rope, heads, ctx_len, eps, inter, workspace, counters,
                  down_parts) -> Partials:
    """All decoder layers of a decode step chained inside ONE persistent launch (psg_decode_layers).
    table: `decode_layer_table`; counters int32 [n_layers * decode_layer_counters()], ZERO; down_parts fp32 [2, 16, M, hidden].
    Returns the last layer's down-projection Partials (for the final rmsnorm)."""
    lib, ctx, st = _env(resid)
    M, hidden = resid.shape
    assert down_parts.shape == (2, 16, M, hidden) and counters.dtype == torch.int32
    assert counters.numel() >= n_layers * decode_layer_counters(resid.device) and table.shape == (n_layers, 8)
    dp, ds = (None, 0) if delta is None else _in(delta, resid.dtype)
    check(lib.psg_decode_layers(ctx, _p(resid, torch.float32, "resid"), dp, ds, _p(table), int(n_layers),
                                _p(tok_pair, torch.int32), _p(tok_pos, torch.int32), _p(rope[0], torch.float32),
                                _p(rope[1], torch.float32), M, hidden, int(inter), int(heads), int(ctx_len), float(eps),
                                _p(workspace, torch.float32), _p(counters), _p(down_parts, torch.float32), _dt(resid), st),
          "psg_decode_layers")
    return Partials(down_parts[(n_layers - 1) & 1])


def silu_mul(gate_up, out):
    lib, ctx, st = _env(out)
    rows, inter = out.shape
    gp, gs = _in(gate_up, out.dtype)
    check(lib.psg_silu_mul(ctx, gp, gs, rows, inter, _p(out), _dt(out), st), "psg_silu_mul")
    return out


def greedy_step(logits, step, max_new, eos, suppress_token, tokens, done, next_ids, tok_pos, dtype=None, embed=None,
                x_out=None):
    """embed [vocab, hidden] + x_out [K, hidden]: the chosen token's embedding row goes to x_out in the same launch (the
    next decode step's input; x_out may be fp32 next to a 16-bit table)."""
    lib, ctx, st = _env(tokens)
    K, vocab = logits.shape
    if isinstance(logits, Partials):
        lp, ls, dt = _p(logits.t), logits.splits, _DT[dtype or torch.bfloat16]
    else:
        lp, ls, dt = _p(logits), 0, _dt(logits)
    if x_out is not None:
        assert embed is not None and x_out.shape == (K, embed.shape[1])
        ep, ed, hid, xp, xd = _p(embed), _dt(embed), embed.shape[1], _p(x_out), _dt(x_out)
    else:
        ep, ed, hid, xp, xd = None, 0, 0, None, 0
    check(lib.psg_greedy_step(ctx, lp, ls, K, vocab, int(step), int(max_new), int(eos), int(suppress_token),
                              _p(tokens, torch.int32), _p(done, torch.int32), _p(next_ids, torch.int32),
                              _p(tok_pos, torch.int32), ep, ed, hid, xp, xd, dt, st), "psg_greedy_step")


def skinny_gemm_plan(M, N, K, dtype, device) -> int:
    """Split count psg_skinny_gemm would use for x [M, K] of `dtype` against w [N, K]; raises PsgHipError where the kernel
    does not take the shape (psg_skinny_gemm_plan: row count, divisibility, the fp32 kernel's LDS bound)."""
    import ctypes
    lib = _lib.load()
    dev = torch.device(device)
    s = ctypes.c_int(0)
    check(lib.psg_skinny_gemm_plan(_lib.ctx(dev.index or 0), int(M), int(N), int(K), _DT[dtype], ctypes.byref(s)),
          "psg_skinny_gemm_plan")
    return s.value


def skinny_gemm(x, w, splits=None) -> Partials:
    """Decode-step projection: fp32 split-K partials of x @ w.T (x: <= 32 rows of bf16 / fp16 / fp32, w of the
    same type); every weight byte streams from HBM once.  Hand the result to a consumer kernel or call .reduce()."""
    lib, ctx, st = _env(x)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if x.dtype not in _DT or w.dtype != x.dtype:
        raise PsgHipError(f"skinny_gemm: x / w must both be bf16, fp16 or fp32, got {x.dtype} / {w.dtype}")
    if splits is None:
        import ctypes
        s = ctypes.c_int(0)
        check(lib.psg_skinny_gemm_plan(ctx, M, N, K, _dt(x), ctypes.byref(s)), "psg_skinny_gemm_plan")
        splits = s.value
    part = torch.empty((splits, M, N), device=x.device, dtype=torch.float32)
    check(lib.psg_skinny_gemm(ctx, _p(x, name="x"), _p(w, name="w"), _p(part), M, N, K, splits, _dt(x), st),
          "psg_skinny_gemm")
    return Partials(part)


def skinny_gemm_w16(x, w16, splits=None) -> Partials:
    """`skinny_gemm` of fp32 rows against weights STORED as fp16 (values that are fp16 numbers: a frozen fp16 checkpoint
    upcast on load): bit-identical to skinny_gemm(x, w16.float()) at half the weight bytes (psg_skinny_gemm_w16)."""
    import ctypes
    lib, ctx, st = _env(x)
    M, K = x.shape
    N = w16.shape[0]
    assert w16.shape[1] == K
    if x.dtype != torch.float32 or w16.dtype != torch.float16:
        raise PsgHipError(f"skinny_gemm_w16: x must be fp32 and w fp16, got {x.dtype} / {w16.dtype}")
    if splits is None:
        s = ctypes.c_int(0)
        check(lib.psg_skinny_gemm_plan(ctx, M, N, K, _DT[torch.float32], ctypes.byref(s)), "psg_skinny_gemm_plan")
        splits = s.value
    part = torch.empty((splits, M, N), device=x.device, dtype=torch.float32)
    check(lib.psg_skinny_gemm_w16(ctx, _p(x, name="x"), _p(w16, name="w"), _p(part), M, N, K, splits, st),
          "psg_skinny_gemm_w16")
    return Partials(part)


def split_f16x2(x):
    """fp32 rows -> (fp16 planes [2, rows, K]: high and low parts, inv_scale fp32 [rows]) - the operand of `split_gemm_w16`."""
    lib, ctx, st = _env(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    rows, K = x.shape
    out = torch.empty((2, rows, K), device=x.device, dtype=torch.float16)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib.psg_split_f16x2(ctx, x.data_ptr(), rows, K, x.stride(0), _p(out), _p(inv), st), "psg_split_f16x2")
    return out, inv


def rmsnorm_split2(resid, delta, w, eps):
    """resid (fp32) += delta (Partials or None); RMSNorm(resid) * w as the two fp16 planes of `split_f16x2`:
    (fp16 [2, rows, hidden], inv_scale) - psg_rmsnorm + psg_split_f16x2 in one launch, bit-identical."""
    lib, ctx, st = _env(resid)
    rows, hidden = resid.shape
    assert resid.dtype == torch.float32 and (delta is None or isinstance(delta, Partials))
    out = torch.empty((2, rows, hidden), device=resid.device, dtype=torch.float16)
    inv = torch.empty(rows, device=resid.device, dtype=torch.float32)
    dp, ds = (None, 0) if delta is None else (_p(delta.t, torch.float32), delta.splits)
    check(lib.psg_rmsnorm_split2(ctx, _p(resid), dp, ds, _p(w, torch.float32), float(eps), rows, hidden, _p(out), _p(inv), st),
          "psg_rmsnorm_split2")
    return out, inv


def split_gemm_w16(x2, inv_scale, w16, mode=0) -> Partials:
    """Decode-step projection of fp32 rows against a weight that is an fp16 value (frozen fp16 checkpoint): two fp16
    products (high and low part of x) on the 16-bit matrix cores, fp32 slices [S, M, N] like `skinny_gemm`'s;
    2^-22 relative against the fp32 product (psg_split_gemm_w16).  x2, inv_scale from `split_f16x2`."""
    import ctypes
    lib, ctx, st = _env(x2)
    _, M, K = x2.shape
    N = w16.shape[0]
    assert x2.shape[0] == 2 and w16.shape[1] == K and x2.dtype == torch.float16 and w16.dtype == torch.float16
    s = ctypes.c_int(0)
    check(lib.psg_split_gemm_w16_plan(ctx, M, N, K, int(mode), ctypes.byref(s)), "psg_split_gemm_w16_plan")
    part = torch.empty((s.value, M, N), device=x2.device, dtype=torch.float32)
    check(lib.psg_split_gemm_w16(ctx, _p(x2), _p(inv_scale, torch.float32), _p(w16), _p(part), M, N, K, s.value, int(mode), st),
          "psg_split_gemm_w16")
    return Partials(part)


def batch_gemm(x, w, slab_rows=0, mode=0) -> Partials:
    """Decode-step projection for 33..160 rows (several images' pairs decoded together): fp32 split-K slices of x @ w.T
    like `skinny_gemm`'s, the weight streamed from HBM once (psg_batch_gemm; bf16 / fp16).
    slab_rows 256 / 128, mode 1 (slab-aligned ranges) / 2 (stream-K): a variant; 0 = the library's estimate."""
    import ctypes
    lib, ctx, st = _env(x)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x.dtype:
        raise PsgHipError(f"batch_gemm: x / w must both be bf16 or fp16, got {x.dtype} / {w.dtype}")
    s = ctypes.c_int(0)
    check(lib.psg_batch_gemm_plan(ctx, M, N, K, _dt(x), int(slab_rows), int(mode), ctypes.byref(s)), "psg_batch_gemm_plan")
    part = torch.empty((s.value, M, N), device=x.device, dtype=torch.float32)
    check(lib.psg_batch_gemm(ctx, _p(x, name="x"), _p(w, name="w"), _p(part), M, N, K, s.value, _dt(x), int(slab_rows),
                             int(mode), st), "psg_batch_gemm")
    return Partials(part)


def skinny_gemm_fused(kind, x_out, w, sync, *, inp=None, resid=None, norm_w=None, eps=0.0, attn=None, splits=None):
    """Decode projection with its producer row operation in the same launch (psg_skinny_gemm_fused):
    the prologue `kind` computes x_out [M, K] from `inp` (Partials / activation tensor / None), then
    part = x_out @ w.T as in `skinny_gemm`.  sync: two zeroed int32 device words owned by this launch.
    attn = (tok_pair, tok_pos, rope, heads, ctx_len, k_cache, v_cache) for the decode-attention prologue."""
    import ctypes
    from ._lib import Prologue
    lib, ctx, st = _env(x_out)
    M, K = x_out.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x_out.dtype and sync.dtype == torch.int32 and sync.numel() >= 2
    if splits is None:
        s = ctypes.c_int(0)
        check(lib.psg_skinny_gemm_plan(ctx, M, N, K, _dt(x_out), ctypes.byref(s)), "psg_skinny_gemm_plan")
        splits = s.value
    pro = Prologue()
    pro.kind = int(kind)
    if inp is not None:
        pro.input, pro.in_splits = _in(inp, x_out.dtype)
    if resid is not None:
        pro.resid = _p(resid, x_out.dtype)
        pro.norm_w = _p(norm_w, torch.float32)
        pro.eps = float(eps)
    if attn is not None:
        tok_pair, tok_pos, rope, heads, ctx_len, kc, vc = attn
        pro.tok_pair, pro.tok_pos = _p(tok_pair, torch.int32), _p(tok_pos, torch.int32)
        pro.rope_cos, pro.rope_sin = _p(rope[0], torch.float32), _p(rope[1], torch.float32)
        pro.heads, pro.ctx = int(heads), int(ctx_len)
        pro.k_cache, pro.v_cache = _p(kc, x_out.dtype), _p(vc, x_out.dtype)
    pro.sync = sync.data_ptr()
    part = torch.empty((splits, M, N), device=x_out.device, dtype=torch.float32)
    check(lib.psg_skinny_gemm_fused(ctx, ctypes.byref(pro), _p(x_out), _p(w), _p(part), M, N, K, splits, _dt(x_out), st),
          "psg_skinny_gemm_fused")
    return Partials(part)


def masked_mean_pool(feat, pan, img_hw, pad_hw, object_ids):
    """Masked-mean object embeddings of the v1-v3 detectors (openseed_relation.py:453-468).
    feat [1,C,Hf,Wf] fp32, pan [H0,W0] int32 id map, object_ids int32 [N] -> [N, C] fp32."""
    import ctypes
    lib, ctx, st = _env(feat)
    _, Cc, Hf, Wf = feat.shape
    N = object_ids.numel()
    nbytes = ctypes.c_int64(0)
    check(lib.psg_masked_mean_pool_workspace(ctx, Cc, Hf, Wf, N, ctypes.byref(nbytes)),
          "psg_masked_mean_pool_workspace")
    ws = torch.empty(nbytes.value // 4, device=feat.device, dtype=torch.int32)
    out = torch.empty((N, Cc), device=feat.device, dtype=torch.float32)
    check(lib.psg_masked_mean_pool(ctx, _p(feat, torch.float32), Cc, Hf, Wf, _p(pan, torch.int32, "pan_results"),
                                   pan.shape[0], pan.shape[1], int(img_hw[0]), int(img_hw[1]), int(pad_hw[0]),
                                   int(pad_hw[1]), _p(object_ids, torch.int32), N, _p(out), _p(ws), nbytes.value, st),
          "psg_masked_mean_pool")
    return out


def masked_split_mean_pool(feat, pan, img_hw, pad_hw, object_ids, output_size):
    """`_mask_pooling(output_size > 1)` of the v1 detector (openseed_relation.py:175-200): per object, its masked pixels in
    row-major order cut into `output_size` chunks, one mean per chunk -> [N, output_size, C] fp32."""
    import ctypes
    lib, ctx, st = _env(feat)
    _, Cc, Hf, Wf = feat.shape
    N = object_ids.numel()
    nbytes = ctypes.c_int64(0)
    check(lib.psg_masked_mean_pool_workspace(ctx, Cc, Hf, Wf, N, ctypes.byref(nbytes)), "psg_masked_mean_pool_workspace")
    ws = torch.empty(nbytes.value // 4, device=feat.device, dtype=torch.int32)
    out = torch.empty((N, int(output_size), Cc), device=feat.device, dtype=torch.float32)
    check(lib.psg_masked_split_mean_pool(ctx, _p(feat, torch.float32), Cc, Hf, Wf, _p(pan, torch.int32, "pan_results"),
                                         pan.shape[0], pan.shape[1], int(img_hw[0]), int(img_hw[1]), int(pad_hw[0]),
                                         int(pad_hw[1]), _p(object_ids, torch.int32), N, int(output_size), _p(out), _p(ws),
                                         nbytes.value, st), "psg_masked_split_mean_pool")
    return out


def bilinear_scores(sub, obj, num_relations):
    """einsum('nrsc,nroc->nrso') of the closed-set heads (relation_transformer_head_v2.py:204-209).
    sub / obj [B, N, R*C] fp32 (the Linear outputs before the reference's reshape + permute) -> [B, R, N, N] fp32."""
    lib, ctx, st = _env(sub)
    B, N, RC = sub.shape
    assert obj.shape == sub.shape and RC % num_relations == 0
    pred = torch.empty((B, num_relations, N, N), device=sub.device, dtype=torch.float32)
    check(lib.psg_bilinear_scores(ctx, _p(sub, torch.float32, "sub"), _p(obj, torch.float32, "obj"), B, N,
                                  num_relations, RC // num_relations, _p(pred), st), "psg_bilinear_scores")
    return pred


def train_object_bitmasks(thing_masks, sem, is_thing, category, thing_index, grid_hw):
    """V4:371-399.  thing_masks uint8 [n_thing,H,W], sem int32 [H,W], per-object int32 vectors -> int64 bits [N, words]."""
    lib, ctx, st = _env(sem)
    H, W = sem.shape
    N = is_thing.numel()
    gh, gw = int(grid_hw[0]), int(grid_hw[1])
    words = (gh * gw + 63) // 64
    bits = torch.empty((N, words), device=sem.device, dtype=torch.int64)
    n_thing = 0 if thing_masks is None else thing_masks.shape[0]
    if n_thing and tuple(thing_masks.shape[1:]) != (H, W):
        raise PsgHipError(f"train_object_bitmasks: thing masks are {tuple(thing_masks.shape[1:])}, the semantic map is "
                          f"{(H, W)}; both must be at the padded image resolution (V4:371-399)")
    check(lib.psg_train_object_bitmasks(ctx, _p(thing_masks, torch.uint8, "thing_masks") if n_thing else None, n_thing,
                                        _p(sem, torch.int32, "sem"), H, W, _p(is_thing, torch.int32),
                                        _p(category, torch.int32), _p(thing_index, torch.int32), N, gh, gw, _p(bits),
                                        words, st), "psg_train_object_bitmasks")
    return bits


def bce_with_logits(logit, label, weight=1.0):
    lib, ctx, st = _env(logit)
    out = torch.empty(1, device=logit.device, dtype=torch.float32)
    check(lib.psg_bce_with_logits(ctx, _p(logit, torch.float32, "logit"), _p(label, torch.float32, "label"),
                                  logit.numel(), float(weight), _p(out), st), "psg_bce_with_logits")
    return out[0]


def cross_entropy_rows(logits, labels):
    """per-row -log softmax(logits)[label] in fp32; label < 0 -> 0 (ignored)."""
    lib, ctx, st = _env(logits)
    rows, vocab = logits.shape
    loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    check(lib.psg_cross_entropy_rows(ctx, _p(logits), rows, vocab, _p(labels, torch.int32, "labels"), _p(loss),
                                     _dt(logits), st), "psg_cross_entropy_rows")
    return loss


TILES = {"auto": 0, "256x256": 1, "256x192": 2, "256x128": 3, "256x64": 4, "128x128": 5}      # enum psg_tile


def dense_gemm(x, w, bias=None, gelu=False, out=None, out_dtype=None, row_scale=None, col_scale=None, tile="256x256",
               swiglu=False):
    """out = [gelu](x @ w.T [* row_scale[:, None] * col_scale[None, :]] + bias) in one pass (bf16 / fp16 operands;
    N % 16 == 0, K % 64 == 0).  out_dtype=torch.float32: fp32 result (needed for the scales: the split-fp16 products of
    the fp32s mode).  Row-count invariant: a row's result does not depend on how many other rows the call has, nor on
    the tile ("auto": the geometry that fills the CUs in the fewest rounds for [M, N]).
    swiglu=True: w is a gate / up weight interleaved by `interleave_gate_up`; out [M, N / 2] = silu(gate) * up."""
    lib, ctx, st = _env(x)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype
    odt = x.dtype if out_dtype is None else out_dtype
    out = torch.empty((M, N // 2 if swiglu else N), device=x.device, dtype=odt) if out is None else out
    assert out.shape == (M, N // 2 if swiglu else N)
    assert not (swiglu and gelu)
    check(lib.psg_dense_gemm_tiled(ctx, _p(x, name="x"), _p(w, name="w"), _p(bias, torch.float32, "bias"),
                                   2 if swiglu else 1 if gelu else 0, _p(out, odt), M, N, K, _dt(x), _DT[odt],
                                   _p(row_scale, torch.float32, "row_scale"), _p(col_scale, torch.float32, "col_scale"),
                                   TILES[tile], st), "psg_dense_gemm")
    return out


def dense_gemm_split(x2, w2, bias, row_scale, col_scale, gelu=False, out=None, tile="auto"):
    """fp32 out = [gelu]((xh.wh + xh.wl + xl.wh) * row_scale[:, None] * col_scale[None, :] + bias) for two fp32 matrices
    given as `split_f16i2` images ([rows, 2K] fp16: per 32 k the high parts, then the low parts): the fp32-grade product
    of the fp32s mode with every operand value staged once (psg_dense_gemm_split).  Row-count and tile invariant."""
    lib, ctx, st = _env(x2)
    M, K2 = x2.shape
    N = w2.shape[0]
    assert w2.shape[1] == K2 and x2.dtype == torch.float16 and w2.dtype == torch.float16
    out = torch.empty((M, N), device=x2.device, dtype=torch.float32) if out is None else out
    assert out.shape == (M, N) and out.dtype == torch.float32
    check(lib.psg_dense_gemm_split(ctx, _p(x2, name="x2"), _p(w2, name="w2"), _p(bias, torch.float32, "bias"),
                                   1 if gelu else 0, _p(out, torch.float32), M, N, K2, _p(row_scale, torch.float32, "row_scale"),
                                   _p(col_scale, torch.float32, "col_scale"), TILES[tile], st), "psg_dense_gemm_split")
    return out


def interleave_gate_up(w_gate_up):
    """[2 inter, K] (gate rows, then up rows) -> the row order the SwiGLU epilogue of `dense_gemm` reads: groups of 16
    rows = 8 gate rows, then the 8 up rows of the same columns."""
    lib, ctx, st = _env(w_gate_up)
    n2, K = w_gate_up.shape
    out = torch.empty_like(w_gate_up)
    check(lib.psg_interleave_gate_up(ctx, _p(w_gate_up, name="gate_up"), _p(out), n2 // 2, K, st), "psg_interleave_gate_up")
    return out


def split_f16x3(x, weights=False):
    """fp32 rows -> three fp16 K segments for an fp32-grade product on the 16-bit matrix cores (psg_split_f16x3):
    activations [hi | hi | lo], weights [hi | lo | hi].  Returns (fp16 [rows, 3K], inv_scale fp32 [rows])."""
    lib, ctx, st = _env(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    rows, K = x.shape
    out = torch.empty((rows, 3 * K), device=x.device, dtype=torch.float16)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib.psg_split_f16x3(ctx, x.data_ptr(), rows, K, x.stride(0), 1 if weights else 0, _p(out), _p(inv), st),
          "psg_split_f16x3")
    return out, inv


def split_f16i2(x):
    """fp32 rows -> (fp16 [rows, 2K]: per 32 k [hi(32) | lo(32)], inv_scale fp32 [rows]) - an operand of
    `dense_gemm_split` (activations and weights alike; psg_split_f16x3 order 2).  K % 32 == 0."""
    lib, ctx, st = _env(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] % 32 == 0
    rows, K = x.shape
    out = torch.empty((rows, 2 * K), device=x.device, dtype=torch.float16)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib.psg_split_f16x3(ctx, x.data_ptr(), rows, K, x.stride(0), 2, _p(out), _p(inv), st), "psg_split_f16x3")
    return out, inv


def scale_rows_cols(y, row_scale, col_scale):
    """y[m][n] *= row_scale[m] * col_scale[n] in place (fp32)."""
    lib, ctx, st = _env(y)
    rows, N = y.shape
    assert row_scale.numel() == rows and col_scale.numel() == N
    check(lib.psg_scale_rows_cols(ctx, _p(y, torch.float32, "y"), rows, N, _p(row_scale, torch.float32),
                                  _p(col_scale, torch.float32), st), "psg_scale_rows_cols")
    return y


class Scaled:
    """A raw split-fp16 GEMM result y [rows, N] fp32 with the power-of-two scales that turn it into the product:
    value = y * (row_scale[m] * col_scale[n]).  Consumed by rmsnorm_split / rope_kvwrite_scaled / silu_mul_split."""
    __slots__ = ("y", "row_scale", "col_scale")

    def __init__(self, y, row_scale, col_scale):
        # y [rows, N], or [slices, rows, N]: the K segments of the product as separate slices (rmsnorm_split sums them)
        assert y.dtype == torch.float32 and y.dim() in (2, 3) and y.is_contiguous()
        assert row_scale.numel() == y.shape[-2] and col_scale.numel() == y.shape[-1]
        self.y, self.row_scale, self.col_scale = y, row_scale, col_scale

    def dense(self):
        y = self.y if self.y.dim() == 2 else self.y.sum(0)
        return scale_rows_cols(y, self.row_scale, self.col_scale)


def rmsnorm_split(resid, delta, w, eps, planes=3):
    """resid += delta (a `Scaled` or None); RMSNorm(resid) * w as split-fp16 segments: (fp16 [rows, 3 hidden], inv_scale);
    planes=2: as two planes (fp16 [2, rows, hidden]: high, low) for a weight that is an fp16 value."""
    lib, ctx, st = _env(resid)
    rows, hidden = resid.shape
    out = torch.empty((rows, 3 * hidden) if planes == 3 else (2, rows, hidden), device=resid.device, dtype=torch.float16)
    inv = torch.empty(rows, device=resid.device, dtype=torch.float32)
    dp, rp, cp = (None, None, None) if delta is None else (_p(delta.y), _p(delta.row_scale, torch.float32),
                                                           _p(delta.col_scale, torch.float32))
    nsl = 1 if delta is None or delta.y.dim() == 2 else delta.y.shape[0]
    assert delta is None or delta.y.shape[-2:] == (rows, hidden)
    check(lib.psg_rmsnorm_split(ctx, _p(resid, torch.float32, "resid"), dp, rp, cp, nsl, _p(w, torch.float32), float(eps),
                                rows, hidden, _p(out), _p(inv), int(planes), st), "psg_rmsnorm_split")
    return out, inv


def rope_kvwrite_scaled(qkv, tok_pair, tok_pos, rope, heads, head_dim, ctx_len, q_out, k_cache, v_cache):
    lib, ctx, st = _env(q_out)
    rows = q_out.shape[0]
    assert isinstance(qkv, Scaled) and qkv.y.shape[-2:] == (rows, 3 * heads * head_dim)
    nsl = 1 if qkv.y.dim() == 2 else qkv.y.shape[0]
    check(lib.psg_rope_kvwrite_scaled(ctx, _p(qkv.y), _p(qkv.row_scale, torch.float32), _p(qkv.col_scale, torch.float32),
                                      _p(tok_pair, torch.int32), _p(tok_pos, torch.int32), _p(rope[0], torch.float32),
                                      _p(rope[1], torch.float32), nsl, rows, heads, head_dim, ctx_len, _p(q_out, torch.float32),
                                      _p(k_cache, torch.float32), _p(v_cache, torch.float32), st), "psg_rope_kvwrite_scaled")


def silu_mul_split(gate_up, inter, planes=3):
    """silu(gate) * up of a `Scaled` gate|up result as split-fp16 segments: (fp16 [rows, 3 inter], inv_scale);
    planes=2: as two planes (fp16 [2, rows, inter])."""
    lib, ctx, st = _env(gate_up.y)
    rows = gate_up.y.shape[-2]
    nsl = 1 if gate_up.y.dim() == 2 else gate_up.y.shape[0]
    assert gate_up.y.shape[-1] == 2 * inter
    out = torch.empty((rows, 3 * inter) if planes == 3 else (2, rows, inter), device=gate_up.y.device, dtype=torch.float16)
    inv = torch.empty(rows, device=gate_up.y.device, dtype=torch.float32)
    check(lib.psg_silu_mul_split(ctx, _p(gate_up.y), _p(gate_up.row_scale, torch.float32), _p(gate_up.col_scale, torch.float32),
                                 nsl, rows, inter, _p(out), _p(inv), int(planes), st), "psg_silu_mul_split")
    return out, inv
