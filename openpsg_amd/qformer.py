"""Relation-query engine: prepare_inference + relation Q-Former + existence head + selector.

Host orchestration of the path the reference runs at
relation_transformer_head_v4.py:146-237 (prepare_inference :408-435, HF InstructBlipQFormerModel
:179-185, binary head :206-209, selector :235-237), re-organised for one MI355X:

  * the cross-attention K/V projection of the patch tensor is done ONCE per image per layer
    (the reference `expand`s patches to all N^2 pairs, V4:168, so HF recomputes it per pair);
  * pair masks are per-object bitmasks OR-ed inside the attention kernel (never materialised);
  * query rows and text rows live in ONE activation matrix, rows [0, B*33) = query rows
    (pair-major) and rows [B*33, B*(33+T)) = text rows, so each dense projection is a single
    hipBLASLt GEMM over all pairs, and the row-wise HIP kernels stream it once;
  * the last layer computes only what `[:, :33]` (V4:185) can observe: no text-row FFN, no
    text-row attention output;
  * nothing synchronises with the host: selection stays on the device.

Dense projections go through torch (`F.linear` -> hipBLASLt); everything else is libpsg_hip.so.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import PSG_EMPTY_UNIFORM, PSG_EMPTY_UNMASKED, PsgHipError
from .config import PSGConfig


class RelationQueryEngine:
    def __init__(self, weights: dict, cfg: PSGConfig, device, dtype=torch.bfloat16, xattn_variant=None, resid_dtype=None,
                 split=False):
        """split (fp32 engines, head dtype 'fp32s'): every projection as a split-fp16 product, see `_lin`.
        resid_dtype=torch.float32 with a 16-bit `dtype` = mixed mode: the projections keep 16-bit operands, but every
        LayerNorm reads its residual in fp32 and writes its result twice - fp32 (the next residual: the residual stream is
        never rounded to 16 bits) and 16-bit (the next projection's operand); psg_add_layernorm_res32."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise PsgHipError(f"activation dtype must be float32, bfloat16 or float16, got {dtype}")
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.res32 = resid_dtype == torch.float32 and dtype != torch.float32
        self.split = bool(split) and dtype == torch.float32
        self.split_i2 = bool(_lib.get_option(self.device.index or 0, "split_i2")) if self.split else False
        self._split_w, self._bias32 = {}, {}
        self.own_gemm = int(_lib.get_option(self.device.index or 0, "qformer_own_gemm"))
        self.xattn_variant = xattn_variant
        self._xidx_cache = {}
        self.index_xattn = True          # layer 0: the cross-attention reads the per-prompt queries through an index
        q = cfg.qformer
        f32 = lambda k: weights[k].to(device=self.device, dtype=torch.float32).contiguous()  # noqa: E731
        act = lambda t: t.to(device=self.device, dtype=dtype).contiguous()                   # noqa: E731
        pre = "relation_qformer.embeddings."
        self.word_emb = f32(pre + "word_embeddings.weight")
        self.pos_emb = f32(pre + "position_embeddings.weight")
        self.emb_ln = (f32(pre + "layernorm.weight"), f32(pre + "layernorm.bias"))
        self.query_rows = torch.cat([f32("rel_cls_query")[0], f32("relation_query")[0]], dim=0).contiguous()
        # patch embedding stays fp32 (exact f32 MFMA kernel): one 8.6 GFLOP GEMM per image over fp32 mask_features
        self.patch_w = f32("patch_embed.proj.weight")
        self.patch_b = f32("patch_embed.proj.bias")
        self.exist_w = f32("binary_rel_cls_pred.weight").reshape(-1).contiguous()
        self.exist_b = f32("binary_rel_cls_pred.bias")
        self.layers = []
        for l in range(q.layers):
            p = f"relation_qformer.encoder.layer.{l}."
            a, x = p + "attention.", p + "crossattention."
            L = dict(
                wqkv=act(torch.cat([weights[a + f"attention.{n}.weight"] for n in ("query", "key", "value")], 0)),
                bqkv=act(torch.cat([weights[a + f"attention.{n}.bias"] for n in ("query", "key", "value")], 0)),
                wo=act(weights[a + "output.dense.weight"]), bo=f32(a + "output.dense.bias"),
                ln_a=(f32(a + "output.LayerNorm.weight"), f32(a + "output.LayerNorm.bias")),
                wq_x=act(weights[x + "attention.query.weight"]), bq_x=act(weights[x + "attention.query.bias"]),
                wk_x=act(weights[x + "attention.key.weight"]), bk_x=act(weights[x + "attention.key.bias"]),
                wv_x=act(weights[x + "attention.value.weight"]), bv_x=act(weights[x + "attention.value.bias"]),
                wo_x=act(weights[x + "output.dense.weight"]), bo_x=f32(x + "output.dense.bias"),
                ln_x=(f32(x + "output.LayerNorm.weight"), f32(x + "output.LayerNorm.bias")),
                w1q=act(weights[p + "intermediate_query.dense.weight"]), b1q=f32(p + "intermediate_query.dense.bias"),
                w2q=act(weights[p + "output_query.dense.weight"]), b2q=f32(p + "output_query.dense.bias"),
                ln_q=(f32(p + "output_query.LayerNorm.weight"), f32(p + "output_query.LayerNorm.bias")),
                w1t=act(weights[p + "intermediate.dense.weight"]), b1t=f32(p + "intermediate.dense.bias"),
                w2t=act(weights[p + "output.dense.weight"]), b2t=f32(p + "output.dense.bias"),
                ln_t=(f32(p + "output.LayerNorm.weight"), f32(p + "output.LayerNorm.bias")),
            )
            if l == q.layers - 1:
                # selection phase in the input space (forward_pairs_cls): per-head W_k and W_v^T of the ROUNDED weights in
                # fp32 (4.7 MB) - the two small batched projections around psg_qformer_cls_attn_input run in fp32
                H, hd = q.hidden, q.hidden // q.heads
                L["wk"] = L["wqkv"][H:2 * H].view(q.heads, hd, H)
                L["wk32"] = L["wk"].float().contiguous()
                L["wv32t"] = L["wqkv"][2 * H:].float().view(q.heads, hd, H).transpose(1, 2).contiguous()
            self.layers.append(L)
        self.empty_policy = PSG_EMPTY_UNIFORM if cfg.empty_row_policy == "uniform" else PSG_EMPTY_UNMASKED
        opt = lambda name: bool(_lib.get_option(self.device.index or 0, name))  # noqa: E731  (options of the psg_ctx)
        self.share_query_qkv = opt("qformer_share_qkv")
        # selection phase of the last layer: cls-row attention in the input space (no K | V projection of all rows)
        # (its two small batched products are library calls whose kernel choice follows the pair count: the row-count
        # invariant modes - split, qformer_own_gemm = 2 - project K | V through `_lin` instead)
        # (round 6: the split mode has row-count-invariant forms of the two products - block-diagonal weights through
        # psg_dense_gemm_split, `_cls_keys_split` / `_cls_values_split` - so it takes the input-space phase and, with it, the
        # per-prompt de-duplication; option qformer_split_cls_input_space = 0 restores the K | V projection of all rows)
        self.cls_input_space = opt("qformer_cls_input_space") and not (self.own_gemm >= 2) and (
            not self.split or (self.split_i2 and opt("qformer_split_cls_input_space")))
        self._cls_big = None
        self._bmm_out_dtype = None       # torch.bmm(..., out_dtype=fp32) available? (probed at first use)
        # two-layer Q-Former: everything in front of layer 0's cross-attention and every text row entering the last
        # layer depend on the PROMPT (class pair) only - computed once per distinct prompt when the caller hands the
        # prompt table over (forward_pairs_cls(prompts=...)) and it has fewer rows than 0.9 x the pairs
        self.dedup_prompts = opt("qformer_dedup_prompts")

    # ---- A4: prepare_inference (V4:408-435) ----------------------------------------------------
    def patch_embed(self, mask_features: torch.Tensor) -> torch.Tensor:
        """[1,C,h,w] fp32 -> patches [L, C] fp32 (timm PatchEmbed: conv k=16 s=16, flatten, transpose)."""
        if self.cfg.patch_size == 16 and mask_features.shape[-1] % 4 == 0 and self.patch_w.shape[0] % 128 == 0:
            return ops.patch_embed(mask_features.contiguous(), self.patch_w, self.patch_b, 16)
        x = F.conv2d(mask_features, self.patch_w, self.patch_b, stride=self.cfg.patch_size)   # odd geometries
        return x.flatten(2).transpose(1, 2)[0].contiguous()

    def object_bitmasks(self, pan: torch.Tensor, img_meta: dict, object_ids: torch.Tensor, feat_hw) -> torch.Tensor:
        gh, gw = feat_hw[0] // self.cfg.patch_size, feat_hw[1] // self.cfg.patch_size
        grid = ops.mask_grid(pan, img_meta["img_shape"][:2], img_meta["pad_shape"][:2], (gh, gw))
        return ops.object_bitmasks(grid, object_ids)

    def cross_kv(self, patches: torch.Tensor):
        """Shared cross-attention K/V, one pair of [L,768] tensors per layer (HF-IB:465-466)."""
        pa = patches.to(self.dtype)
        return [(self._lin(pa, L["wk_x"], L["bk_x"]), self._lin(pa, L["wv_x"], L["bv_x"])) for L in self.layers]

    # ---- A6 + A7: Q-Former over a list of pairs ---------------------------------------------------
    # Activations travel as (X, X32): X in the activation dtype (the projections' operand), X32 its fp32 twin in mixed mode
    # (the residual of the next LayerNorm), else None.
    def _embed(self, ids):
        """HF-IB:728-757 for P pairs.  Returns (X [P*(33+T), 768], X32, shared0): with shared0 the embedded query rows are
        ONE [33, 768] block for all pairs (learned tokens through the embedding LayerNorm; 16-bit modes): layer 0
        projects it once and uses it as a periodic residual, so rows [33, P*33) of X stay unwritten."""
        q = self.cfg.qformer
        nq, H = q.q_rows, q.hidden
        P, T = ids.shape
        R, RQ = P * (nq + T), P * nq
        X = torch.empty((R, H), device=self.device, dtype=self.dtype)
        X32 = torch.empty((R, H), device=self.device, dtype=torch.float32) if self.res32 else None
        first = X32 if self.res32 else X
        shared0 = len(self.layers) > 1 and T > 0 and self.share_query_qkv
        if shared0:
            ops.qformer_embed_split(ids, self.word_emb, self.pos_emb, self.query_rows, self.emb_ln[0], self.emb_ln[1],
                                    q.ln_eps, first[:nq], first[RQ:])
            if self.res32:
                X[:nq].copy_(X32[:nq])
                X[RQ:].copy_(X32[RQ:])
        else:
            ops.qformer_embed(ids, self.word_emb, self.pos_emb, self.query_rows, self.emb_ln[0], self.emb_ln[1],
                              q.ln_eps, first)
            if self.res32:
                X.copy_(X32)
        return X, X32, shared0

    def _ln(self, x, r16, r32, bias, ln, out16=None, period=0, index=None, want32=True):
        """LayerNorm(x + bias + residual) -> (result in the activation dtype - in place of x unless out16 is given -,
        fp32 twin or None).  period / index: residual rows from a periodic table / from table blocks chosen per group."""
        eps = self.cfg.qformer.ln_eps
        if self.res32:
            return ops.add_layernorm_res32(x, r32, bias, ln[0], ln[1], eps, out16=out16, period=period, index=index,
                                           want32=want32)
        if index is not None:
            return ops.add_layernorm_indexed(x, r16, index, period, bias, ln[0], ln[1], eps, out=out16), None
        if period:
            return ops.add_layernorm_periodic(x, r16, bias, ln[0], ln[1], eps, out=out16), None
        return ops.add_layernorm(x, r16, bias, ln[0], ln[1], eps, out=out16), None

    @staticmethod
    def _s(t, a, b=None):
        return None if t is None else (t[a:] if b is None else t[a:b])

    def _cross(self, li, qx, nq, kv, bits, num_objects, pair_index, segments):
        """Masked cross-attention of `nq` rows per pair (33, or 1 = the cls row alone) against the image's patches."""
        q = self.cfg.qformer
        if segments is None:
            return ops.qformer_cross_attn(qx, kv[li][0], kv[li][1], bits, pair_index, num_objects, nq, q.heads,
                                          empty_policy=self.empty_policy, variant=self.xattn_variant)
        cx = torch.empty_like(qx)
        for ps, pc, kv_m, bits_m, n_m in segments:
            ops.qformer_cross_attn(qx[ps * nq:(ps + pc) * nq], kv_m[li][0], kv_m[li][1], bits_m,
                                   pair_index[ps:ps + pc], n_m, nq, q.heads, out=cx[ps * nq:(ps + pc) * nq],
                                   empty_policy=self.empty_policy, variant=self.xattn_variant)
        return cx

    def _layer(self, li, X, X32, P, T, text_mask, pair_index, kv, bits, num_objects, segments, shared0=False,
               hidden_out=None):
        """One Q-Former layer (HF-IB:446-596) over P pairs; the last layer computes the query rows only.
        Returns (Xn, Xn32)."""
        q = self.cfg.qformer
        nq, H = q.q_rows, q.hidden
        R, RQ = P * (nq + T), P * nq
        L = self.layers[li]
        last = li == len(self.layers) - 1
        ctx = torch.empty((R, H), device=self.device, dtype=self.dtype)
        if li == 0 and shared0:
            # the query rows entering layer 0 are identical for every pair: project the first pair's 33 rows once
            qkv_q = self._lin(X[:nq], L["wqkv"], L["bqkv"])
            qkv = self._lin(X[RQ:], L["wqkv"], L["bqkv"])
            ops.qformer_self_attn_shared(qkv_q, qkv, text_mask, P, T, nq, q.heads, ctx)
        else:
            qkv = self._lin(X, L["wqkv"], L["bqkv"])
            ops.qformer_self_attn(qkv, text_mask, P, T, nq, q.heads, last, ctx)
        del qkv
        ra = RQ if last else R
        A = self._lin(ctx[:ra], L["wo"])
        A32 = torch.empty((ra, H), device=self.device, dtype=torch.float32) if self.res32 else None
        if li == 0 and shared0:
            self._ln_into(A[:RQ], X[:nq], self._s(X32, 0, nq), L["bo"], L["ln_a"], self._s(A32, 0, RQ), period=nq)
            self._ln_into(A[RQ:], X[RQ:], self._s(X32, RQ), L["bo"], L["ln_a"], self._s(A32, RQ))
        else:
            self._ln_into(A, X[:ra], self._s(X32, 0, ra), L["bo"], L["ln_a"], A32)
        del ctx
        qx = self._lin(A[:RQ], L["wq_x"], L["bq_x"])
        cx = self._cross(li, qx, nq, kv, bits, num_objects, pair_index, segments)
        Cq = self._lin(cx, L["wo_x"])
        _, Cq32 = self._ln(Cq, A[:RQ], self._s(A32, 0, RQ), L["bo_x"], L["ln_x"])
        del qx, cx
        if last and hidden_out is not None:
            assert hidden_out.shape == (RQ, H) and hidden_out.dtype == self.dtype and hidden_out.is_contiguous()
            Xn = hidden_out
        else:
            Xn = torch.empty((RQ if last else R, H), device=self.device, dtype=self.dtype)
        Xn32 = torch.empty((Xn.shape[0], H), device=self.device, dtype=torch.float32) if self.res32 else None
        iq = self._ffn1(Cq, L["w1q"], L["b1q"])
        hq = self._lin(iq, L["w2q"])
        self._ln_into(hq, Cq, Cq32, L["b2q"], L["ln_q"], self._s(Xn32, 0, RQ), out16=Xn[:RQ])
        del iq, hq
        if not last and T > 0:
            it = self._ffn1(A[RQ:], L["w1t"], L["b1t"])
            ht = self._lin(it, L["w2t"])
            self._ln_into(ht, A[RQ:], self._s(A32, RQ), L["b2t"], L["ln_t"], self._s(Xn32, RQ), out16=Xn[RQ:])
            del it, ht
        return Xn, Xn32

    def _ln_into(self, x, r16, r32, bias, ln, out32, out16=None, period=0, index=None):
        """_ln writing its fp32 twin into a caller-owned slice (mixed mode); returns (16-bit result, out32)."""
        eps = self.cfg.qformer.ln_eps
        if self.res32:
            o16, _ = ops.add_layernorm_res32(x, r32, bias, ln[0], ln[1], eps, out16=out16, out32=out32, period=period,
                                             index=index)
            return o16, out32
        return self._ln(x, r16, None, bias, ln, out16=out16, period=period, index=index)

    def forward_pairs(self, kv, bits, num_objects, pair_index, ids, text_mask, hidden_out=None, segments=None):
        """pair_index int32 [P] (p = i*N + j), ids int32 [P,T], text_mask uint8 [P,T].
        Returns (hidden [P*33, 768] in the activation dtype, exist_logit [P] fp32, exist_prob [P] fp32).
        hidden_out: caller-owned [P*33, 768] buffer the last layer writes into (no copy when pairs are chunked).
        segments: [(first pair, pair count, kv, bits, num_objects)] - the pairs come from several images (pair
        sharding); everything but the cross-attention runs over all of them at once."""
        P, T = ids.shape
        X, X32, shared0 = self._embed(ids)
        for li in range(len(self.layers)):
            X, X32 = self._layer(li, X, X32, P, T, text_mask, pair_index, kv, bits, num_objects, segments, shared0,
                                 hidden_out if li == len(self.layers) - 1 else None)
        logit, prob = ops.exist_head(X32 if X32 is not None else X, self.exist_w, self.exist_b, P, self.cfg.qformer.q_rows)
        return X, logit, prob

    def forward_pairs_cls(self, kv, bits, num_objects, pair_index, ids, text_mask, segments=None, prompts=None):
        """Selection phase: everything the existence logits depend on, and nothing else.

        The existence head reads row 0 (the cls row) of the last layer's output (V4:206-209), and rows 1..32
        (`pair_feature`, V4:215) are used for the SELECTED pairs only (V4:235-237, 293-301).  In the last layer every
        row-wise operation (projections, FFN, LayerNorm) is independent across rows and the attentions need the other
        rows only as keys / values, so the cls row of all P pairs is computed here with the full K/V and rows 1..32
        are computed afterwards for the chosen pairs (`pair_hidden`) - identical results, the last layer's query-row
        work shrinks from 33 rows to 1 for all but the selected pairs.
        prompts = (ids_u int32 [U, T], mask_u uint8 [U, T], inv int32 [P][, rows int32 [P*33]]): the distinct prompts of
        these pairs and each pair's row in that table (ids == ids_u[inv]); lets the prompt-only work run on U rows instead
        of P.  rows (optional, cached by the caller) = inv[p] * 33 + r, the query rows of each pair's prompt block.
        Returns (state, exist_logit [P], exist_prob [P]); state feeds `pair_hidden`."""
        q = self.cfg.qformer
        nq, H = q.q_rows, q.hidden
        P, T = ids.shape
        RQ = P * nq
        nl = len(self.layers)
        in_space = (self.cls_input_space and H == 768 and q.heads == 12
                    and (nq + T) * (H + (8 if self.dtype != torch.float32 else 0)) * (
                        4 if self.dtype == torch.float32 else 2) + q.heads * 256 <= 160 * 1024)
        if (prompts is not None and segments is None and nl == 2 and T > 0 and in_space and self.dedup_prompts
                and prompts[0].shape[0] <= 0.9 * P):
            return self._forward_pairs_cls_dedup(kv, bits, num_objects, pair_index, text_mask, prompts)
        X, X32, shared0 = self._embed(ids)
        for li in range(nl - 1):
            X, X32 = self._layer(li, X, X32, P, T, text_mask, pair_index, kv, bits, num_objects, segments, shared0)
        logit, prob = self._cls_phase(X[:RQ], self._s(X32, 0, RQ), X[RQ:], None, text_mask, P, T, kv, bits, num_objects,
                                      pair_index, segments, in_space, X)
        state = dict(X=X, X32=X32, P=P, T=T, text_mask=text_mask, pair_index=pair_index, kv=kv, bits=bits,
                     num_objects=num_objects, segments=segments)
        return state, logit, prob

    def _cls_phase(self, Xq, Xq32, Xt, text_index, mask, P, T, kv, bits, num_objects, pair_index, segments, in_space, X=None):
        """Last layer for the cls row of every pair.  Xq [P*33, H] query rows entering the layer (Xq32: fp32 twin); Xt:
        text rows, block text_index[p] (None: block p) per pair, `mask` indexed the same way.  X: the two as one tensor
        (K | V form)."""
        q = self.cfg.qformer
        nq, H = q.q_rows, q.hidden
        li, L = len(self.layers) - 1, self.layers[-1]
        x_cls = Xq.view(P, nq, H)[:, 0].contiguous()                         # [P, H] residual of the cls rows
        x_cls32 = Xq32.view(P, nq, H)[:, 0].contiguous() if Xq32 is not None else None
        q_cls = self._lin(x_cls, L["wqkv"][:H], L["bqkv"][:H])              # queries of the cls rows only
        hd = H // q.heads
        if in_space:
            # keys / values never materialised: the cls queries go back through W_k (g_h = W_k,h^T q_h), the kernel
            # reads the layer's input rows once, the weighted row means go through W_v (psg_qformer_cls_attn_input)
            if self.split:
                g = self._cls_keys_split(q_cls, L)                                                # fp32 [heads, P, H]
            else:
                g = self._bmm_f32(q_cls.view(P, q.heads, hd).transpose(0, 1), L["wk"], L["wk32"])  # fp32 [heads, P, H]
            if X is not None:
                xbar = ops.qformer_cls_attn_input(X, g, mask, P, T, nq, q.heads)
            else:
                xbar = ops.qformer_cls_attn_input(Xq, g, mask, P, T, nq, q.heads, x_text=Xt, text_index=text_index)
            if self.split:
                ctx = self._cls_values_split(xbar, L)
            else:
                ctx = (torch.bmm(xbar, L["wv32t"]).permute(1, 0, 2).reshape(P, H) + L["bqkv"][2 * H:].float()).to(self.dtype)
            del g, xbar
        else:
            assert X is not None
            kvs = self._lin(X, L["wqkv"][H:], L["bqkv"][H:])                 # keys | values of every row
            ctx = ops.qformer_self_attn_cls(q_cls, kvs, mask, P, T, nq, q.heads)
            del kvs
        A = self._lin(ctx, L["wo"])
        _, A32 = self._ln(A, x_cls, x_cls32, L["bo"], L["ln_a"])
        qx = self._lin(A, L["wq_x"], L["bq_x"])
        cx = self._cross(li, qx, 1, kv, bits, num_objects, pair_index, segments)
        Cq = self._lin(cx, L["wo_x"])
        _, Cq32 = self._ln(Cq, A, A32, L["bo_x"], L["ln_x"])
        iq = self._ffn1(Cq, L["w1q"], L["b1q"])
        hq = self._lin(iq, L["w2q"])
        Xc, Xc32 = self._ln(hq, Cq, Cq32, L["b2q"], L["ln_q"])
        return ops.exist_head(Xc32 if Xc32 is not None else Xc, self.exist_w, self.exist_b, P, 1)

    def _forward_pairs_cls_dedup(self, kv, bits, num_objects, pair_index, text_mask, prompts):
        """forward_pairs_cls with the prompt-only work done per DISTINCT prompt (two layers).  Layer 0's input is the
        learned query block plus the prompt's embeddings, so its whole self-attention block (HF-IB:471-530) is a
        function of the prompt; a pair enters at the cross-attention (its object masks).  The text rows never see the
        cross-attention at all: their layer-0 output - the last layer's text keys / values - is per prompt too."""
        q = self.cfg.qformer
        nq, H = q.q_rows, q.hidden
        ids_u, mask_u, inv = prompts[:3]
        U, T = ids_u.shape
        P = inv.numel()
        RQu = U * nq
        L = self.layers[0]
        Xu, Xu32, shared0 = self._embed(ids_u)
        ctx = torch.empty((U * (nq + T), H), device=self.device, dtype=self.dtype)
        if shared0:
            qkv_q = self._lin(Xu[:nq], L["wqkv"], L["bqkv"])
            qkv = self._lin(Xu[RQu:], L["wqkv"], L["bqkv"])
            ops.qformer_self_attn_shared(qkv_q, qkv, mask_u, U, T, nq, q.heads, ctx)
        else:
            qkv = self._lin(Xu, L["wqkv"], L["bqkv"])
            ops.qformer_self_attn(qkv, mask_u, U, T, nq, q.heads, False, ctx)
        del qkv
        A = self._lin(ctx, L["wo"])
        A32 = torch.empty((A.shape[0], H), device=self.device, dtype=torch.float32) if self.res32 else None
        if shared0:
            self._ln_into(A[:RQu], Xu[:nq], self._s(Xu32, 0, nq), L["bo"], L["ln_a"], self._s(A32, 0, RQu), period=nq)
            self._ln_into(A[RQu:], Xu[RQu:], self._s(Xu32, RQu), L["bo"], L["ln_a"], self._s(A32, RQu))
        else:
            self._ln_into(A, Xu, Xu32, L["bo"], L["ln_a"], A32)
        del ctx
        it = self._ffn1(A[RQu:], L["w1t"], L["b1t"])                        # text rows: straight to their layer-0 output
        ht = self._lin(it, L["w2t"])
        Xt_u, Xt_u32 = self._ln(ht, A[RQu:], self._s(A32, RQu), L["b2t"], L["ln_t"])
        del it, ht
        # the pair enters at the cross-attention: its queries are its prompt's 33 projected rows (projected per prompt,
        # gathered per pair - the cross-attention kernel streams its Q tiles by DMA and takes no index), the residual
        # of the output LayerNorm is read from the prompt's block through the index
        qx_u = self._lin(A[:RQu], L["wq_x"], L["bq_x"])
        if len(prompts) > 3:                                              # cached with the prompt table (names only)
            rows = prompts[3]
        else:
            rows = (inv.to(torch.int64)[:, None] * nq + torch.arange(nq, device=self.device)[None, :]).reshape(-1).to(torch.int32)
        cx = None
        if (self.index_xattn and self.dtype in (torch.bfloat16, torch.float16) and nq == 33
                and self.xattn_variant in (None, ops.PSG_XATTN_MFMA)):
            # the LDS-DMA kernel looks the prompt's block up itself (pair tiles: one scalar load; cls tiles: the P cls
            # rows gathered here) - no [P x 33, H] expansion of the queries (127 MB written and read again at C2)
            if len(prompts) > 3:                                          # index tensors of a cached prompt table: built once
                key = (rows.data_ptr(), inv.data_ptr(), P)               # (the entry keeps both tensors alive: the pointers
                ent = self._xidx_cache.get(key)                          # cannot be handed out again while it exists)
                if ent is None:
                    if len(self._xidx_cache) > 64:
                        self._xidx_cache.clear()
                    ent = self._xidx_cache[key] = (rows[::nq].contiguous(), inv.to(torch.int32).contiguous(), rows, inv)
            else:                                                         # `rows` is a temporary of this call: nothing to key on
                ent = (rows[::nq].contiguous(), inv.to(torch.int32).contiguous())
            q_cls = torch.empty((P, H), device=self.device, dtype=self.dtype)
            ops.gather_rows(qx_u, ent[0], q_cls)
            cx = ops.qformer_cross_attn_indexed(qx_u, ent[1], q_cls, kv[0][0], kv[0][1], bits, pair_index,
                                                num_objects, q.heads, empty_policy=self.empty_policy)
        if cx is None:
            qx = torch.empty((P * nq, H), device=self.device, dtype=self.dtype)
            ops.gather_rows(qx_u, rows, qx)
            cx = self._cross(0, qx, nq, kv, bits, num_objects, pair_index, None)
            del qx
        Cq = self._lin(cx, L["wo_x"])
        _, Cq32 = self._ln(Cq, A[:RQu], self._s(A32, 0, RQu), L["bo_x"], L["ln_x"], period=nq, index=inv)
        del cx, qx_u
        iq = self._ffn1(Cq, L["w1q"], L["b1q"])
        hq = self._lin(iq, L["w2q"])
        Xq, Xq32 = self._ln(hq, Cq, Cq32, L["b2q"], L["ln_q"])
        del iq, hq, Cq
        logit, prob = self._cls_phase(Xq, Xq32, Xt_u, inv, mask_u, P, T, kv, bits, num_objects, pair_index, None, True)
        state = dict(Xq=Xq, Xq32=Xq32, Xt_u=Xt_u, Xt_u32=Xt_u32, inv=inv, P=P, T=T, text_mask=text_mask,
                     pair_index=pair_index, kv=kv, bits=bits, num_objects=num_objects, segments=None)
        return state, logit, prob

    def pair_hidden(self, state, sel, segments=None):
        """Last layer in full for the pairs `sel` (int32 [K], positions in the pair list of `forward_pairs_cls`;
        negative = no pair, computed as pair 0 and to be ignored).  Returns hidden [K*33, 768].
        segments: [(first slot, slot count, kv, bits, num_objects)] when the slots belong to several images of one
        `forward_pairs_cls(segments=...)` pass (everything but the cross-attention runs over all slots at once)."""
        q = self.cfg.qformer
        nq = q.q_rows
        P, T = state["P"], state["T"]
        assert segments is not None or state["segments"] is None, "pair_hidden: pass the slots' segments"
        s64 = sel.to(torch.int64).clamp(min=0)
        K = s64.numel()
        ar = torch.arange(nq, device=self.device)
        rows = [(s64[:, None] * nq + ar[None, :]).reshape(-1)]
        Xs32 = None
        if "Xt_u" in state:                                                  # text rows live in the per-prompt table
            trows = (state["inv"].to(torch.int64).index_select(0, s64)[:, None] * T
                     + torch.arange(T, device=self.device)[None, :]).reshape(-1)
            Xs = torch.cat([state["Xq"].index_select(0, rows[0]), state["Xt_u"].index_select(0, trows)])
            if state.get("Xq32") is not None:
                Xs32 = torch.cat([state["Xq32"].index_select(0, rows[0]), state["Xt_u32"].index_select(0, trows)])
        else:
            if T > 0:
                rows.append((P * nq + s64[:, None] * T + torch.arange(T, device=self.device)[None, :]).reshape(-1))
            allrows = torch.cat(rows)
            Xs = state["X"].index_select(0, allrows)                        # [K*(33+T), 768]: query rows, then text rows
            if state.get("X32") is not None:
                Xs32 = state["X32"].index_select(0, allrows)
        tm = state["text_mask"].index_select(0, s64) if T > 0 else state["text_mask"]
        pi = state["pair_index"].index_select(0, s64)
        return self._layer(len(self.layers) - 1, Xs, Xs32, K, T, tm, pi, state["kv"], state["bits"], state["num_objects"],
                           segments)[0]

    def pair_hidden_sel(self, state, sel, first, count, slot_off=0):
        """`pair_hidden` for GLOBAL pair ids of a single-image pass: the rows of the selected pairs, their text masks and
        pair ids come out of ONE gather kernel (psg_gather_pair_rows) instead of ~20 index-arithmetic launches.
        sel int32 [K]; the chunk holds the pairs [first, first + count) at positions slot_off.. of the pass; slots of pairs
        outside it are computed as the chunk's first pair.  Returns (hidden [K*33, 768], mine uint8 [K])."""
        q = self.cfg.qformer
        nq = q.q_rows
        P, T = state["P"], state["T"]
        assert state["segments"] is None
        if "Xt_u" in state:                                                  # text rows live in the per-prompt table
            xq, xt, tix = state["Xq"], state["Xt_u"], state["inv"]
            xq32, xt32 = state.get("Xq32"), state.get("Xt_u32")
        else:
            X, X32 = state["X"], state.get("X32")
            xq, xt, tix = X[:P * nq], X[P * nq:], None
            xq32, xt32 = (None, None) if X32 is None else (X32[:P * nq], X32[P * nq:])
        sel = sel.to(torch.int32).contiguous()
        Xs, tm, pi, mine = ops.gather_pair_rows(xq, xt, tix, state["text_mask"], state["pair_index"], sel, first, count,
                                                slot_off, nq, T)
        Xs32 = None
        if xq32 is not None:
            Xs32 = ops.gather_pair_rows(xq32, xt32, tix, None, None, sel, first, count, slot_off, nq, T, want_aux=False)[0]
        if T == 0:
            tm = state["text_mask"]
        hk = self._layer(len(self.layers) - 1, Xs, Xs32, sel.numel(), T, tm, pi, state["kv"], state["bits"],
                         state["num_objects"], None)[0]
        return hk, mine

    def _cls_big_weights(self, L):
        """Block-diagonal forms of the last layer's key / value weights as interleaved hi / lo images (split mode): the two
        per-head products of the input-space selection phase become ONE psg_dense_gemm_split call each - 12x the flops of
        the batched product (zeros), still ~0.1 ms at 2500 pairs, and row-count invariant where the library's batched GEMM
        is not (a pair shard reproduces the full pass bit for bit).
          keys:   g[p, h H + c]  = sum_d q[p, 64 h + d] W_k[64 h + d, c]          W_kbig [heads H, H]
          values: ctx[p, 64 h + d] = sum_c xbar[p, h H + c] W_v[64 h + d, c]     W_vbig [H, heads H]"""
        if self._cls_big is None:
            q = self.cfg.qformer
            H, heads = q.hidden, q.heads
            hd = H // heads
            wk = L["wqkv"][H:2 * H].float()                                  # [H, H]: row 64 h + d = key feature d of head h
            wv = L["wqkv"][2 * H:].float()
            kbig = torch.zeros((heads * H, H), device=self.device, dtype=torch.float32)
            vbig = torch.zeros((H, heads * H), device=self.device, dtype=torch.float32)
            for h in range(heads):
                kbig[h * H:(h + 1) * H, h * hd:(h + 1) * hd] = wk[h * hd:(h + 1) * hd].t()
                vbig[h * hd:(h + 1) * hd, h * H:(h + 1) * H] = wv[h * hd:(h + 1) * hd]
            self._cls_big = (ops.split_f16i2(kbig), ops.split_f16i2(vbig), L["bqkv"][2 * H:].float().contiguous())
        return self._cls_big

    def _cls_keys_split(self, q_cls, L):
        """g_h = W_k,h^T q_h for every pair and head, fp32 [heads, P, H] (the layout psg_qformer_cls_attn_input reads)."""
        q = self.cfg.qformer
        P = q_cls.shape[0]
        kb, _, _ = self._cls_big_weights(L)
        a2, inv_r = ops.split_f16i2(q_cls.contiguous())
        g = ops.dense_gemm_split(a2, kb[0], None, inv_r, kb[1], tile="auto" if P < 16384 else "256x256")
        return g.view(P, q.heads, q.hidden).permute(1, 0, 2).contiguous()

    def _cls_values_split(self, xbar, L):
        """ctx = W_v applied to the weighted row means xbar [heads, P, H] (+ the value bias), fp32 [P, H]."""
        q = self.cfg.qformer
        P = xbar.shape[1]
        _, vb, bias = self._cls_big_weights(L)
        x2, inv_r = ops.split_f16i2(xbar.permute(1, 0, 2).reshape(P, q.heads * q.hidden).contiguous())
        return ops.dense_gemm_split(x2, vb[0], bias, inv_r, vb[1], tile="auto" if P < 16384 else "256x256")

    def _bmm_f32(self, a, b, b32):
        """fp32 result of a batched product of activation-dtype operands (exact products, fp32 accumulation): the
        library's 16-bit-in / fp32-out batched GEMM where this PyTorch has it, else the fp32 GEMM on widened copies."""
        if a.dtype == torch.float32:
            return torch.bmm(a, b32)
        if self._bmm_out_dtype is None:
            try:
                torch.bmm(a[:, :1], b, out_dtype=torch.float32)
                self._bmm_out_dtype = True
            except (TypeError, RuntimeError):
                self._bmm_out_dtype = False
        if self._bmm_out_dtype:
            return torch.bmm(a, b, out_dtype=torch.float32)
        return torch.bmm(a.float(), b32)

    def _lin(self, x, w, b=None, gelu=False):
        """Linear layer of the Q-Former (HF-IB: every `nn.Linear` on the path), optionally with the exact-erf GELU.
          * split mode (head dtype 'fp32s'): fp32 in / out as a split-fp16 product on the 16-bit matrix cores -
            [xh | xh | xl] . [wh | wl | wh]^T through psg_dense_gemm with fp32 accumulation, the operands' power-of-two
            row scales undone in its epilogue (psg_split.hip; ~7e-7 per product, the class of the library SGEMM).  The
            kernel walks the whole K per output tile, so a row's result does not depend on the row count of the call:
            a pair shard reproduces the full pass bit for bit;
          * 16-bit modes with context option qformer_own_gemm = 2: psg_dense_gemm for every projection (the same
            row-count invariance; 1 = only the two FFN1 projections, where the fused GELU saves a pass);
          * otherwise the library GEMM."""
        N, K = w.shape
        fits = N % 256 == 0 and K % 64 == 0 and x.shape[0] > 0
        if self.split and fits and x.dtype == torch.float32:
            i2 = self.split_i2 and K % 32 == 0                 # round 6: every operand value staged once (psg_dense_gemm_split)
            key = (w.data_ptr(), N, K, i2)
            ws = self._split_w.get(key)
            if ws is None:
                ws = self._split_w[key] = ops.split_f16i2(w) if i2 else ops.split_f16x3(w, weights=True)
            # (a row's result does not depend on the tile: below ~30 k rows the 256 x 256 tile leaves most CUs idle -
            # 2500 x 768 is 30 tiles - and the geometry that fills them in the fewest rounds is taken instead)
            tile = "auto" if x.shape[0] < 16384 else "256x256"
            if i2:
                a2, inv_r = ops.split_f16i2(x)
                return ops.dense_gemm_split(a2, ws[0], b, inv_r, ws[1], gelu=gelu, tile=tile)
            a3, inv_r = ops.split_f16x3(x)
            return ops.dense_gemm(a3, ws[0], b, gelu=gelu, out_dtype=torch.float32, row_scale=inv_r, col_scale=ws[1],
                                  tile=tile)
        if self.own_gemm >= 2 and fits and x.dtype != torch.float32 and x.is_contiguous():
            if b is not None and b.dtype != torch.float32:
                key = (b.data_ptr(), N)
                b32 = self._bias32.get(key)
                if b32 is None:
                    b32 = self._bias32[key] = b.float().contiguous()
                b = b32
            return ops.dense_gemm(x, w, b, gelu=gelu)
        if gelu:
            y = F.linear(x, w)
            ops.bias_gelu(y, b)
            return y
        return F.linear(x, w, b)

    def _ffn1(self, x, w, b):
        """intermediate(_query): Linear + exact-erf GELU (HF-IB:563-577).  16-bit modes: one pass through
        psg_dense_gemm (own MFMA GEMM with the bias + GELU epilogue fused; context option qformer_own_gemm);
        fp32 verification mode and odd shapes: library GEMM + psg_bias_gelu."""
        if self.split or self.own_gemm >= 2:
            return self._lin(x, w, b, gelu=True)
        if (self.dtype != torch.float32 and w.shape[0] % 256 == 0 and w.shape[1] % 64 == 0 and x.shape[0] >= 256
                and self.own_gemm):
            return ops.dense_gemm(x, w, b, gelu=True)
        y = F.linear(x, w)
        ops.bias_gelu(y, b)
        return y

    def select(self, prob: torch.Tensor, k: int):
        """V4:235-237 on the device: descending, ties -> lower pair index.  int32 [k] (-1 if n < k)."""
        return ops.topk(prob, k)[0]
