"""Gradient path of the training branch: the reference back-propagates `binary_rel_cls_loss` and
`rel_llm_loss` (relation_transformer_head_v4.py:327-351, 463-482) into patch_embed, the relation Q-Former, the two
query parameters, the existence head and `language_projection`; the LLM is frozen (configs/psg/baseline_v4_ov.py:65), so
its weights get no gradient but the loss reaches the trainable parameters THROUGH its layers.

The graph is torch.autograd's; its nodes are
  * the dense projections (`F.linear`, library GEMM; their weight gradients too), and
  * `torch.autograd.Function`s whose forward and backward are the fp32 kernels of csrc/psg_train_bwd.hip: LayerNorm,
    RMSNorm, the three attentions (Q-Former self / cross with the pair masks, Llama causal), GELU, the SwiGLU gate,
    rotary, cross entropy and BCE-with-logits.
Training batches are tiny (<= 32 sampled pairs, <= 4 LLM pairs, V4:29-30, 38), so this path is written for exactness
against autograd on the CPU oracle (tests/test_gpu_train.py), not for speed.  fp32 only; no CPU path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops
from ._lib import PsgHipError, check


def _env(t):
    return ops._env(t)


def _f32(t, name="tensor"):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise PsgHipError(f"{name}: the gradient path runs in fp32 on the GPU (got {t.dtype} on {t.device})")
    return t.contiguous()


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = _f32(x, "layernorm x"), _f32(gamma), _f32(beta)
        hidden = x.shape[-1]
        rows = x.numel() // hidden
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib, c, st = _env(x)
        check(lib.psg_train_layernorm_fwd(c, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), rows, hidden,
                                          y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), st), "psg_train_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _f32(dy)
        hidden = x.shape[-1]
        rows = x.numel() // hidden
        dx = torch.empty_like(x)
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
        lib, c, st = _env(x)
        check(lib.psg_train_layernorm_bwd(c, x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          rows, hidden, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), st),
              "psg_train_layernorm_bwd")
        return dx, dg, db, None


class RMSNormFn(torch.autograd.Function):
    """HF-LL:53-67 with a FROZEN weight (no weight gradient)."""

    @staticmethod
    def forward(ctx, x, w, eps):
        x, w = _f32(x, "rmsnorm x"), _f32(w)
        hidden = x.shape[-1]
        rows = x.numel() // hidden
        y = torch.empty_like(x)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        lib, c, st = _env(x)
        check(lib.psg_train_rmsnorm_fwd(c, x.data_ptr(), w.data_ptr(), float(eps), rows, hidden, y.data_ptr(),
                                        rstd.data_ptr(), st), "psg_train_rmsnorm_fwd")
        ctx.save_for_backward(x, w, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rstd = ctx.saved_tensors
        dy = _f32(dy)
        hidden = x.shape[-1]
        dx = torch.empty_like(x)
        lib, c, st = _env(x)
        check(lib.psg_train_rmsnorm_bwd(c, x.data_ptr(), dy.data_ptr(), w.data_ptr(), rstd.data_ptr(), x.numel() // hidden,
                                        hidden, dx.data_ptr(), st), "psg_train_rmsnorm_bwd")
        return dx, None, None


class AttnFn(torch.autograd.Function):
    """softmax(q.k * scale + additive mask) v.  q [B, Sq, H*D]; k, v [Bk, Sk, H*D] with Bk == B or 1 (shared by all
    sequences); keep uint8 [B, Mq, Sk], Mq == Sq or 1.  An all-masked row gives a uniform softmax (additive finfo.min).
    drop: None, or (uint8 keep mask [B, heads, Sq, Sk], 1 / (1 - p)) - dropout on the attention probabilities."""

    @staticmethod
    def forward(ctx, q, k, v, keep, heads, scale, drop=None):
        q, k, v = _f32(q, "attention q"), _f32(k), _f32(v)
        keep = keep.to(torch.uint8).contiguous()
        B, Sq, hid = q.shape
        Bk, Sk, _ = k.shape
        D = hid // heads
        Mq = keep.shape[1]
        assert keep.shape == (B, Mq, Sk) and v.shape == k.shape and Bk in (B, 1)
        dmask, dscale = (None, 1.0) if drop is None else (drop[0].to(torch.uint8).contiguous(), float(drop[1]))
        assert dmask is None or dmask.shape == (B, heads, Sq, Sk)
        p = torch.empty((B, heads, Sq, Sk), device=q.device, dtype=torch.float32)
        out = torch.empty_like(q)
        lib, c, st = _env(q)
        check(lib.psg_train_attn_fwd(c, q.data_ptr(), k.data_ptr(), v.data_ptr(), keep.data_ptr(), B, Bk, heads, Sq, Sk, D,
                                     Mq, float(scale), None if dmask is None else dmask.data_ptr(), dscale, p.data_ptr(),
                                     out.data_ptr(), st), "psg_train_attn_fwd")
        ctx.save_for_backward(q, k, v, p)
        ctx.heads, ctx.scale, ctx.dmask, ctx.dscale = heads, float(scale), dmask, dscale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, p = ctx.saved_tensors
        dout = _f32(dout)
        B, Sq, hid = q.shape
        Bk, Sk, _ = k.shape
        dq = torch.empty_like(q)
        dk, dv = torch.zeros_like(k), torch.zeros_like(v)
        lib, c, st = _env(q)
        check(lib.psg_train_attn_bwd(c, q.data_ptr(), k.data_ptr(), v.data_ptr(), p.data_ptr(), dout.data_ptr(), B, Bk,
                                     ctx.heads, Sq, Sk, hid // ctx.heads, ctx.scale,
                                     None if ctx.dmask is None else ctx.dmask.data_ptr(), ctx.dscale, dq.data_ptr(),
                                     dk.data_ptr(), dv.data_ptr(), st), "psg_train_attn_bwd")
        return dq, dk, dv, None, None, None, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "gelu x")
        y = torch.empty_like(x)
        lib, c, st = _env(x)
        check(lib.psg_train_gelu_fwd(c, x.data_ptr(), x.numel(), y.data_ptr(), st), "psg_train_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = _f32(dy)
        dx = torch.empty_like(x)
        lib, c, st = _env(x)
        check(lib.psg_train_gelu_bwd(c, x.data_ptr(), dy.data_ptr(), x.numel(), dx.data_ptr(), st), "psg_train_gelu_bwd")
        return dx


class SiluMulFn(torch.autograd.Function):
    """gate_up [rows, 2*inter] -> silu(gate) * up [rows, inter] (HF-LL:163-177)."""

    @staticmethod
    def forward(ctx, gu):
        gu = _f32(gu, "gate_up")
        rows, two = gu.shape
        y = torch.empty((rows, two // 2), device=gu.device, dtype=torch.float32)
        lib, c, st = _env(gu)
        check(lib.psg_train_silu_mul_fwd(c, gu.data_ptr(), rows, two // 2, y.data_ptr(), st), "psg_train_silu_mul_fwd")
        ctx.save_for_backward(gu)
        return y

    @staticmethod
    def backward(ctx, dy):
        gu, = ctx.saved_tensors
        dy = _f32(dy)
        d = torch.empty_like(gu)
        lib, c, st = _env(gu)
        check(lib.psg_train_silu_mul_bwd(c, gu.data_ptr(), dy.data_ptr(), gu.shape[0], gu.shape[1] // 2, d.data_ptr(), st),
              "psg_train_silu_mul_bwd")
        return d


class RopeFn(torch.autograd.Function):
    """Half-split rotary (HF-LL:130-160) on x [rows, heads*head_dim] at table rows `pos` (int32 [rows])."""

    @staticmethod
    def forward(ctx, x, pos, cos, sin, heads):
        x = _f32(x, "rope x")
        if pos.numel() and (int(pos.max()) >= cos.shape[0] or int(pos.min()) < 0):      # training only: one read-back
            raise PsgHipError(f"rope: position {int(pos.max())} outside the {cos.shape[0]}-row rotary table")
        ctx.save_for_backward(pos, cos, sin)
        ctx.heads = heads
        return RopeFn._run(x, pos, cos, sin, heads, 1.0)

    @staticmethod
    def _run(x, pos, cos, sin, heads, sign):
        rows, hid = x.shape
        y = torch.empty_like(x)
        lib, c, st = _env(x)
        check(lib.psg_train_rope(c, x.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), cos.shape[0], rows, heads,
                                 hid // heads, float(sign), y.data_ptr(), st), "psg_train_rope")
        return y

    @staticmethod
    def backward(ctx, dy):
        pos, cos, sin = ctx.saved_tensors
        return RopeFn._run(_f32(dy), pos, cos, sin, ctx.heads, -1.0), None, None, None, None


class CrossEntropyRowsFn(torch.autograd.Function):
    """Per-row -log softmax(logits)[label]; label < 0 = ignored (loss 0, zero gradient).  V4:337-341."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = _f32(logits, "logits")
        loss = ops.cross_entropy_rows(logits, labels)
        ctx.save_for_backward(logits, labels)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, labels = ctx.saved_tensors
        dloss = _f32(dloss)
        d = torch.empty_like(logits)
        lib, c, st = _env(logits)
        check(lib.psg_train_ce_bwd(c, logits.data_ptr(), logits.shape[0], logits.shape[1], labels.data_ptr(),
                                   dloss.data_ptr(), d.data_ptr(), st), "psg_train_ce_bwd")
        return d, None


class BceFn(torch.autograd.Function):
    """mean BCE-with-logits x weight (V4:463-482, binary case)."""

    @staticmethod
    def forward(ctx, logit, label, weight):
        logit, label = _f32(logit, "logit"), _f32(label)
        ctx.save_for_backward(logit, label)
        ctx.weight = float(weight)
        return ops.bce_with_logits(logit, label, weight)

    @staticmethod
    def backward(ctx, dloss):
        logit, label = ctx.saved_tensors
        dloss = _f32(dloss).reshape(1)
        d = torch.empty_like(logit)
        lib, c, st = _env(logit)
        check(lib.psg_train_bce_bwd(c, logit.data_ptr(), label.data_ptr(), logit.numel(), ctx.weight, dloss.data_ptr(),
                                    d.data_ptr(), st), "psg_train_bce_bwd")
        return d, None, None


class PatchEmbedFn(torch.autograd.Function):
    """timm PatchEmbed (V4:410): forward = the exact-fp32 matrix-core kernel of the inference path; backward = the weight
    gradient as one GEMM over the unfolded feature map (the features themselves come from the frozen segmenter)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, patch):
        feat = _f32(feat, "mask_features")
        ctx.save_for_backward(feat)
        ctx.patch, ctx.wshape = patch, weight.shape
        if patch == 16 and feat.shape[-1] % 4 == 0 and weight.shape[0] % 128 == 0:
            return ops.patch_embed(feat, _f32(weight), _f32(bias), 16)
        return F.conv2d(feat, weight, bias, stride=patch).flatten(2).transpose(1, 2)[0].contiguous()

    @staticmethod
    def backward(ctx, dp):
        feat, = ctx.saved_tensors
        cols = F.unfold(feat, ctx.patch, stride=ctx.patch)[0]               # [C * patch * patch, L]
        dp = _f32(dp)                                                        # [L, Cout]
        dw = (dp.t() @ cols.t()).view(ctx.wshape)
        return None, dw, dp.sum(0), None


def layer_norm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps)


class Dropout:
    """The Q-Former's dropouts as the reference trains it: V4:78-84 builds InstructBlipQFormerConfig with its defaults
    (hidden_dropout_prob = attention_probs_dropout_prob = 0.1) and tools/train.py puts the model in train() mode, so HF
    applies dropout after the embedding LayerNorm (HF-IB:728-757), on the attention probabilities (HF-IB:176-196) and on
    every dense output before its residual LayerNorm (HF-IB:519-530, 579-596).  Masks are drawn with torch's generator
    of `device` (or the given one; a CPU generator + upload lets a test replay the same masks in the CPU oracle), in the
    order the layers run."""

    def __init__(self, p_hidden=0.1, p_attn=0.1, generator=None):
        self.p_hidden, self.p_attn, self.generator = float(p_hidden), float(p_attn), generator

    def _keep(self, shape, p, device):
        gdev = device if self.generator is None else self.generator.device
        return (torch.rand(shape, device=gdev, generator=self.generator) >= p).to(device)

    def hidden(self, x):
        if self.p_hidden <= 0:
            return x
        return x * (self._keep(x.shape, self.p_hidden, x.device).to(x.dtype) / (1.0 - self.p_hidden))

    def attn(self, B, heads, Sq, Sk, device):
        if self.p_attn <= 0:
            return None
        return self._keep((B, heads, Sq, Sk), self.p_attn, device).to(torch.uint8), 1.0 / (1.0 - self.p_attn)


def qformer_pairs(P, cfg, patches, ids, text_mask, pair_keep, dropout: Dropout | None = None):
    """The relation Q-Former (HF-IB:446-757 as driven by V4:179-185) over B pairs, all rows of all layers (training
    keeps the text rows of the last layer out of the loss, V4:185, but computes them like the reference).
    P: parameters by reference name; patches [L, C]; ids int64 [B, T]; text_mask [B, T]; pair_keep uint8 [B, L].
    dropout: None = off (the oracle comparison), or a `Dropout` plan.  Returns the last hidden state [B, 33 + T, 768]."""
    q = cfg.qformer
    nq, H, heads = q.q_rows, q.hidden, q.heads
    B, T = ids.shape
    pre = "relation_qformer.embeddings."
    query = torch.cat([P["rel_cls_query"][0], P["relation_query"][0]], dim=0)                      # V4:155-157
    emb = P[pre + "word_embeddings.weight"][ids] + P[pre + "position_embeddings.weight"][:T][None]
    h = layer_norm(torch.cat([query[None].expand(B, -1, -1), emb], dim=1), P[pre + "layernorm.weight"],
                   P[pre + "layernorm.bias"], q.ln_eps)
    dev = h.device
    dh = (lambda x: x) if dropout is None else dropout.hidden                                    # noqa: E731
    da = (lambda *a: None) if dropout is None else dropout.attn                                  # noqa: E731
    S, L = nq + T, patches.shape[0]
    h = dh(h)
    self_keep = torch.cat([torch.ones((B, nq), dtype=torch.uint8, device=dev), text_mask.to(torch.uint8)], dim=1)[:, None, :]
    cross_keep = pair_keep.to(torch.uint8)[:, None, :]
    scale = (H // heads) ** -0.5
    lin = lambda pfx, x: F.linear(x, P[pfx + ".weight"], P[pfx + ".bias"])  # noqa: E731
    for l in range(q.layers):
        p = f"relation_qformer.encoder.layer.{l}."
        a = AttnFn.apply(lin(p + "attention.attention.query", h), lin(p + "attention.attention.key", h),
                         lin(p + "attention.attention.value", h), self_keep, heads, scale, da(B, heads, S, S, dev))
        a = layer_norm(dh(lin(p + "attention.output.dense", a)) + h, P[p + "attention.output.LayerNorm.weight"],
                       P[p + "attention.output.LayerNorm.bias"], q.ln_eps)
        q33 = a[:, :nq]
        kx = lin(p + "crossattention.attention.key", patches)[None]                           # shared by every pair
        vx = lin(p + "crossattention.attention.value", patches)[None]
        c = AttnFn.apply(lin(p + "crossattention.attention.query", q33), kx, vx, cross_keep, heads, scale,
                         da(B, heads, nq, L, dev))
        c = layer_norm(dh(lin(p + "crossattention.output.dense", c)) + q33, P[p + "crossattention.output.LayerNorm.weight"],
                       P[p + "crossattention.output.LayerNorm.bias"], q.ln_eps)
        hq = layer_norm(dh(lin(p + "output_query.dense", GeluFn.apply(lin(p + "intermediate_query.dense", c)))) + c,
                        P[p + "output_query.LayerNorm.weight"], P[p + "output_query.LayerNorm.bias"], q.ln_eps)
        at = a[:, nq:]
        ht = layer_norm(dh(lin(p + "output.dense", GeluFn.apply(lin(p + "intermediate.dense", at)))) + at,
                        P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], q.ln_eps)
        h = torch.cat([hq, ht], dim=1)
    return h


def llama_teacher_forcing(engine, cfg, X, seq_len, rope_pos, rows):
    """Plain `language_model(inputs_embeds, attention_mask)` forward (V4:327-336) through the FROZEN Llama, on compact
    sequences X [K, S, D] (valid tokens first), rope_pos int32 [K, S] = each token's position in the reference's padded
    sequence (-1 behind a sequence's end), rows int64: flat row indices whose logits are wanted.  Returns fp32 logits."""
    m = cfg.llm
    K, S, D = X.shape
    dev = X.device
    valid = torch.arange(S, device=dev)[None, :] < seq_len[:, None]                            # [K, S]
    causal = torch.tril(torch.ones((S, S), dtype=torch.bool, device=dev))
    keep = (causal[None] & valid[:, None, :])
    keep = keep | (~valid)[:, :, None] & torch.eye(S, dtype=torch.bool, device=dev)[None]     # pad rows attend to themselves
    keep = keep.to(torch.uint8).contiguous()
    pos = rope_pos.clamp(min=0).reshape(-1).to(torch.int32).contiguous()
    cos, sin = engine.rope
    x = X
    scale = m.head_dim ** -0.5
    for L in engine.layers:
        n1 = RMSNormFn.apply(x, L["ln1"], m.rms_eps)
        qkv = F.linear(n1, L["wqkv"])
        qh = RopeFn.apply(qkv[..., :D].reshape(K * S, D), pos, cos, sin, m.heads).view(K, S, D)
        kh = RopeFn.apply(qkv[..., D:2 * D].reshape(K * S, D), pos, cos, sin, m.heads).view(K, S, D)
        att = AttnFn.apply(qh, kh, qkv[..., 2 * D:].contiguous(), keep, m.heads, scale)
        x = x + F.linear(att, L["wo"])
        n2 = RMSNormFn.apply(x, L["ln2"], m.rms_eps)
        act = SiluMulFn.apply(F.linear(n2, L["wgu"]).view(K * S, -1)).view(K, S, -1)
        x = x + F.linear(act, L["wdown"])
    hfin = RMSNormFn.apply(x, engine.final_norm, m.rms_eps).reshape(K * S, D)
    return F.linear(hfin.index_select(0, rows), engine.lm_head)
