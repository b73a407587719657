"""LMM relation decode engine: batched greedy Llama decode for the K selected pairs.

Replaces the serial loop at relation_transformer_head_v4.py:293-312 (one HF `generate` per pair,
batch 1, fp32 weights re-streamed ~17 times per pair) with ONE batched prefill + (max_new-1)
batched decode steps, so the LLM weights stream from HBM once per step for all K pairs.

Layout decisions (all per-pair results are independent of K and of the other pairs):
  * the reference left-pads prompts, which puts pad tokens in the MIDDLE of the sequence (after the
    32 visual rows, V4:296-301) and relies on HF's additive mask + cumsum positions.  Here every
    pair's sequence is COMPACTED: slots [0, 32+n_k) hold its tokens, position == slot
    (== cumsum(mask)-1, probe-verified in SURVEY Appendix B), so the only mask is causal;
  * prefill runs on a fixed [K, 32+T_p] token grid (rows beyond a pair's length carry pos = -1 and
    are skipped by the kernels), decode on [K] rows: static shapes, no host sync, graph-capturable;
  * KV cache [layers][K, heads, ctx, 128] in the activation dtype;
  * the greedy step (argmax, EOS bookkeeping, next ids, positions) is a device kernel.

Dense projections go through torch (`F.linear` -> hipBLASLt) or, for the decode steps, through the
hand-written weight-streaming kernel in libpsg_hip.so; RMSNorm, rotary+KV write, attention, SwiGLU
gate and the greedy step are HIP kernels.
"""
from __future__ import annotations

import collections
import os
import sys

import torch
import torch.nn.functional as F

from . import ops
from ._lib import PsgHipError
from .config import PSGConfig


# How the library runs one split-fp16 product [rows, K'] x [N, K']^T -> fp32: whole, or cut into column / row parts written
# in place, or as a batched product over K segments.  hipBLASLt's pick for a shape is a heuristic, and at K' = 3K it is erratic
# (Llama-2-7B gate|up at 960 rows: 694 us whole, 480 us as two column halves; tools/split_gemm_bench.py), so a cut pays.
# The forms are NOT numerically equivalent (K segments change the fp32 summation order, parts can make the library pick
# another kernel), so the choice must not depend on anything a process measures: it comes from the FIXED table below - the
# measured winners for the Llama-2-7B shapes on gfx950 (gpurun_out/plans_run*.txt of round 6: two runs of the old per-process
# timing disagreed on 3 of 12 shapes at near-ties, which is exactly what the table retires).  Every rank, every box and every
# process then runs the same arithmetic on the same input (tests/test_gpu_plans.py: two fresh processes, identical logits
# and tokens).  `PSG_PLAN=measure` brings the timing back as a TOOL: it times the candidates and prints table lines.
#   key: (N, K', result is fp32, the reader sums [P, rows, N] slices)  ->  [(rows from, rows below, plan), ...]; else whole
_SPLIT_PLAN_TABLE = {
    # fp32s prompt pass, generic fp32 weights: [xh | xh | xl] . [wh | wl | wh]^T over K' = 3K (960 rows at BASELINE C3)
    (22016, 12288, True, False): [(512, 2048, ("cols", 2))],      # gate|up: 702-709 us whole, 466-494 as two column halves
    (4096, 33024, True, True): [(512, 2048, ("kseg", 6))],        # down: 335-347 us whole, 283-293 as six K segments
    # fp32s prompt pass, fp16-valued weights: the two-plane operand [xh; xl] (2 x rows) against the fp16 weight over K
    (22016, 4096, True, False): [(1024, 6144, ("cols", 2))],      # gate|up: 375 us whole, 326-332 in halves (1920 rows)
    # 16-bit prompt pass of several images (forward_batch: 1920 / 3840 rows; 7680: whole)
    (22016, 4096, False, False): [(1024, 6144, ("cols", 2))],     # 598-604 us whole, 534-539 in halves (3840 rows)
}
_SPLIT_PLANS: dict = {}                                            # PSG_PLAN=measure only: what this process measured


def _plan_measuring() -> bool:
    return os.environ.get("PSG_PLAN", "") == "measure"


def _split_mm(a3, w3, plan=None, out_dtype=torch.float32):
    """a3 [rows, K'] . w3 [N, K']^T -> [rows, N] through the library, whole or in the parts `plan` names.
    out_dtype fp32: the split-fp16 products of the fp32s mode; None: the operands' 16-bit type (the 16-bit prompt pass)."""
    kw = {} if out_dtype is None else {"out_dtype": out_dtype}
    if plan is None or plan[0] == "whole":
        return torch.mm(a3, w3.t(), **kw)
    if plan[0] == "kseg":                                      # P segments of K' as ONE batched product: [P, rows, N] slices
        rows, N, P = a3.shape[0], w3.shape[0], plan[1]
        K = a3.shape[1] // P
        return torch.bmm(a3.view(rows, P, K).permute(1, 0, 2), w3.view(N, P, K).permute(1, 2, 0), **kw)
    rows, N = a3.shape[0], w3.shape[0]
    y = torch.empty((rows, N), device=a3.device, dtype=out_dtype or a3.dtype)
    kind, parts = plan
    if kind == "cols":                                         # ldc = N: the parts land where the whole product puts them
        h = N // parts
        for p_ in range(parts):
            torch.mm(a3, w3[p_ * h:(p_ + 1) * h].t(), out=y[:, p_ * h:(p_ + 1) * h], **kw)
    else:
        h = -(-rows // (16 * parts)) * 16
        for r0 in range(0, rows, h):
            torch.mm(a3[r0:r0 + h], w3.t(), out=y[r0:r0 + h], **kw)
    return y


def _plan_split_mm(rows, w3, out_dtype=torch.float32, k3=False, pool=None):
    """The form the library product of this shape runs in: a pure function of the shape (`_SPLIT_PLAN_TABLE`; 'whole' for
    every shape the table does not name), unless PSG_PLAN=measure."""
    if _plan_measuring():
        return _measure_split_plan(rows, w3, out_dtype, k3, pool)
    N, K3 = w3.shape
    for lo, hi, plan in _SPLIT_PLAN_TABLE.get((int(N), int(K3), out_dtype == torch.float32, bool(k3)), ()):
        if lo <= rows < hi:
            return plan
    return ("whole",)


def _measure_split_plan(rows, w3, out_dtype=torch.float32, k3=False, pool=None):
    """PSG_PLAN=measure (a tool, not the product path: its result depends on timing noise).
    The fastest of {whole, 2 / 3 column parts, 2 / 3 / 4 row parts} for this shape, timed on the device (7 runs each,
    minimum); a cut must win by 5 % to be taken.  Cached per process; prints the line `_SPLIT_PLAN_TABLE` would carry.
    k3: the caller's reader can sum slices (psg_rmsnorm_split) - 3 / 6 / 12 segments of K' as ONE batched product with a
    [P, rows, N] result are candidates too (P x the tiles: down at 960 rows 331 us whole, 303 as 3, 247 as 6 slices; o
    105 -> 98); the reader's extra slice reads are charged at 4 TB/s.
    pool: the other weights of this shape (the layers'): timed in rotation, so that a weight small enough for the
    Infinity Cache is as cold as it is in the pass itself."""
    N, K3 = w3.shape
    key = (int(rows), int(N), int(K3), w3.device.index or 0, str(out_dtype), bool(k3), str(w3.dtype))
    plan = _SPLIT_PLANS.get(key)
    if plan is not None:
        return plan
    if rows < 128 or torch.cuda.is_current_stream_capturing():
        return ("whole",)                                      # (not cached while capturing: planned at the next eager use)
    cands = [("whole",)]
    cands += [("cols", p_) for p_ in (2, 3) if N % (256 * p_) == 0 and N // p_ >= 2048]
    cands += [("rows", p_) for p_ in (2, 3, 4) if rows >= 128 * p_]
    if k3 and out_dtype == torch.float32:
        cands += [("kseg", p_) for p_ in (3, 6, 12) if K3 % (64 * p_) == 0]
    # (drain the device first: the timing loop is EAGER library work, and an eager library GEMM beside one inside another
    # slot's replaying graph is the combination that hung the GPU in round 4, tools/inflight_stress.py)
    torch.cuda.synchronize(w3.device)
    a3 = torch.randn((rows, K3), device=w3.device, generator=torch.Generator(device=w3.device).manual_seed(0)).to(w3.dtype)
    best, best_t, whole_t = cands[0], None, None
    pool = [w_ for w_ in (pool or [w3]) if w_.shape == w3.shape and w_.dtype == w3.dtype] or [w3]
    for c in cands:
        ts = []
        for i in range(9):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            _split_mm(a3, pool[i % len(pool)], c, out_dtype)
            e_.record()
            e_.synchronize()
            if i >= 2:
                ts.append(s_.elapsed_time(e_))
        t = min(ts)
        if c[0] == "kseg":
            t += (c[1] - 1) * rows * N * 4 / 4e9               # ms: the reader sums c[1] slices instead of reading one
        if c[0] == "whole":
            whole_t = best_t = t
        elif t < best_t and t < 0.95 * whole_t:
            best, best_t = c, t
    _SPLIT_PLANS[key] = best
    print(f"[psg plan] ({N}, {K3}, {out_dtype == torch.float32}, {bool(k3)}): rows={rows} -> {best}  # {best_t * 1e3:.0f} us "
          f"(whole {whole_t * 1e3:.0f} us)", file=sys.stderr, flush=True)
    return best


# Decode steps with 33..160 rows (several images' pairs decoded together, head.forward_batch): the library GEMM against the
# variants of psg_batch_gemm (slab height x range mode) per (rows, N, K).  'own' hands fp32 slices to the consumer where 'lib'
# hands a 16-bit-rounded result, so - like the split products above - the choice is a FIXED table of the measured winners at
# the Llama-2-7B shapes (COLD weights, the consumer's slice sums charged to the variants; profiles/r05_batch_gemm_ab.txt,
# gpurun_out/plans_run*.txt of round 6), never a per-process measurement.  At 160 rows the library streams o / down at
# 1.0-1.3 TB/s and wins on gate|up; at 40 rows psg_batch_gemm wins everywhere but the lm_head.
#   key: (N, K) -> [(rows from, rows below, plan), ...]; anything else: the library
_BATCH_PLAN_TABLE = {
    (12288, 4096): [(33, 49, ("own", 128, 2))],                                        # q|k|v: 80 / 160 rows library (32 / 40 us)
    (4096, 4096): [(33, 49, ("own", 128, 1)), (113, 161, ("own", 128, 1))],            # o: 80 rows library 22 us; 160: 29 vs 31
    (22016, 4096): [(33, 49, ("own", 128, 2))],                                        # gate|up
    (4096, 11008): [(33, 113, ("own", 128, 1)), (113, 161, ("own", 256, 1))],          # down: 160 rows 42 us vs library 72
}
_BATCH_PLANS: dict = {}                                            # PSG_PLAN=measure only


def _plan_batch_mm(x, pool):
    if _plan_measuring():
        return _measure_batch_plan(x, pool)
    w, rows = pool[0], int(x.shape[0])
    for lo, hi, plan in _BATCH_PLAN_TABLE.get((int(w.shape[0]), int(w.shape[1])), ()):
        if lo <= rows < hi:
            return plan
    return ("lib",)


def _measure_batch_plan(x, pool):
    """PSG_PLAN=measure (a tool): library vs the psg_batch_gemm variants, timed once per process on COLD weights (the layers'
    own weights of that shape in rotation) with a psg_reduce_partials pass charged to the variants."""
    w = pool[0]
    key = (int(x.shape[0]), int(w.shape[0]), int(w.shape[1]), str(x.dtype), str(w.dtype), w.device.index or 0)
    plan = _BATCH_PLANS.get(key)
    if plan is not None:
        return plan
    if torch.cuda.is_current_stream_capturing():
        return ("lib",)
    torch.cuda.synchronize(w.device)                           # (as in _measure_split_plan: no eager library work beside a replaying graph)
    cands = [("lib",)] + [("own", bn, mode) for bn in (256, 128) for mode in (1, 2)]
    best, best_t = cands[0], None
    for c in cands:
        if c[0] == "own":
            try:
                ops.batch_gemm(x, w, c[1], c[2])
            except PsgHipError:
                continue
        ts = []
        for i in range(8):                                     # a sample = 4 launches back to back on 4 different weights
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for j in range(4):
                wi = pool[(4 * i + j) % len(pool)]
                if c[0] == "lib":
                    F.linear(x, wi)
                else:
                    ops.batch_gemm(x, wi, c[1], c[2]).reduce(x.dtype)
            e_.record()
            e_.synchronize()
            if i >= 2:
                ts.append(s_.elapsed_time(e_) / 4)
        t = sorted(ts)[len(ts) // 2]
        if best_t is None or t < best_t * (0.97 if best[0] == "lib" else 1.0):
            best, best_t = c, t
    _BATCH_PLANS[key] = best
    print(f"[psg plan] decode projection ({key[1]}, {key[2]}): rows={key[0]} -> {best}  # {best_t * 1e3:.0f} us", file=sys.stderr,
          flush=True)
    return best


class LlamaDecodeEngine:
    def __init__(self, weights: dict, cfg: PSGConfig, device, dtype=torch.bfloat16, n_layers=None, resid_dtype=None,
                 prefill_split=False):
        """prefill_split (fp32 engines): the prompt pass's projections as split-fp16 products on the 16-bit matrix cores,
        `linear_split` below; the decode steps stay exact fp32 (psg_gemm_f32.hip).
        resid_dtype: storage type of the residual stream; None = `dtype` (what HF keeps for a model cast to 16
        bits), torch.float32 with a 16-bit `dtype` = mixed mode (16-bit GEMM operands, the residual stream - the sum
        of 2 x layers updates - never rounded to 16 bits; costs 160 KB more traffic per decode row kernel)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise PsgHipError(f"activation dtype must be float32, bfloat16 or float16, got {dtype}")
        self.resid_dtype = dtype if resid_dtype is None else resid_dtype
        if self.resid_dtype not in (dtype, torch.float32):
            raise PsgHipError(f"residual dtype must be {dtype} or float32, got {self.resid_dtype}")
        m = cfg.llm
        if m.head_dim != 128:
            raise PsgHipError(f"LLM head_dim {m.head_dim} unsupported (kernels are built for 128)")
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.n_layers = m.layers if n_layers is None else n_layers        # llm_truncate_num (V4:101-103)
        f32 = lambda k: weights[k].to(device=self.device, dtype=torch.float32).contiguous()   # noqa: E731
        act = lambda t: t.to(device=self.device, dtype=dtype).contiguous()                     # noqa: E731
        self.embed = act(weights["language_model.model.embed_tokens.weight"])
        self.lm_head = act(weights["language_model.lm_head.weight"])
        self.final_norm = f32("language_model.model.norm.weight")
        self.proj_w = act(weights["language_projection.weight"])
        self.proj_b = act(weights["language_projection.bias"])
        self.layers = []
        for l in range(self.n_layers):
            p = f"language_model.model.layers.{l}."
            self.layers.append(dict(
                wqkv=act(torch.cat([weights[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)),
                wo=act(weights[p + "self_attn.o_proj.weight"]),
                wgu=act(torch.cat([weights[p + "mlp.gate_proj.weight"], weights[p + "mlp.up_proj.weight"]], 0)),
                wdown=act(weights[p + "mlp.down_proj.weight"]),
                ln1=f32(p + "input_layernorm.weight"), ln2=f32(p + "post_attention_layernorm.weight")))
        self.use_skinny = True
        self.prefill_split = bool(prefill_split) and dtype == torch.float32
        # fp32 engines: are the projection weights fp16 VALUES?  The reference's LLM is the frozen Llama-2-7b-hf checkpoint
        # (fp16 on disk, configs/psg/baseline_v4_ov.py:61-65) upcast by from_pretrained (V4:99-100): if every element of
        # every projection round-trips through fp16, the decode steps stream an fp16 copy - half the HBM bytes - instead
        # of the fp32 tensor: exact mode 'fp32' on the same f32 matrix instructions (psg_skinny_gemm_w16, bit-identical),
        # mode 'fp32s' as two fp16 products of the split activations (psg_split_gemm_w16, 2^-22).  Anything trained in
        # fp32 fails the check and keeps the fp32 stream.
        self._w16 = {}
        self._skinny_ok = {}
        self._w16_all = False
        self._ones = {}
        from . import _lib as _lib0
        if dtype == torch.float32 and _lib0.get_option(self.device.index or 0, "llm_w16"):
            tensors = [L[k] for L in self.layers for k in ("wqkv", "wo", "wgu", "wdown")] + [self.lm_head]
            for t in tensors:                                   # per tensor: a fine-tuned lm_head keeps its fp32 stream alone
                h = t.half()
                if torch.equal(h.float(), t):
                    self._w16[t.data_ptr()] = h
            self._w16_all = len(self._w16) == len(tensors)      # the fused decode step / two-plane prompt pass need all of them
        if self.prefill_split:
            # [wh | wl | wh] fp16 + the per-row power of two that undoes the row scaling, per projection (3 x 2 bytes per
            # weight next to the fp32 copy the decode steps stream: 40 GB + 27 GB for Llama-2-7B, of 288 GB)
            for L in self.layers:
                for k in ("wqkv", "wo", "wgu", "wdown"):
                    L[k + "_s"] = ops.split_f16x3(L[k], weights=True)
        # row_invariant (fp32s engines): the prompt pass's projections and the language projection on psg_dense_gemm - one
        # k-ordered accumulation per output element whatever the row count of the call - instead of the library GEMM, whose
        # kernel choice follows the row count: a pair decoded in a batch of 3 (decodes DEALT over the ranks of a pair-sharded
        # job) then gives the bits it gives in the batch of 20.  ~1.3x the prompt pass's GEMM time: the sharded
        # pipeline switches it on, a single GPU does not need it
        self.row_invariant = False
        self.split_i2 = bool(_lib0.get_option(self.device.index or 0, "split_i2"))
        self._i2_w = {}
        self.proj_s = ops.split_f16x3(self.proj_w, weights=True) if self.prefill_split else None
        # greedy argmax over the fp32 split-K sums of the lm_head, NOT over their 16-bit rounding: HF computes the logits of
        # a model cast to 16 bits in 16 bits, but the reference runs the LLM in fp32 (V4:99-100), and a 16-bit logit has
        # an ulp of 0.008-0.016 (fp16) / 0.06 (bf16) at |x| ~ 8-16 - wider than many top-2 margins of a 32000-way
        # argmax, so rounding first turns near-ties into ties that the lower index wins
        self.exact_argmax = True
        self._mm_out_dtype = None        # torch.mm(..., out_dtype=fp32) available? (probed at first use)
        # row operations that run as the PROLOGUE of the projection that consumes them (one launch instead of two;
        # psg_skinny_gemm_fused).  Built for "rmsnorm", bit-identical, and OFF by default: measured 26.6 vs 25.6 us
        # per (RMSNorm + q/k/v projection), 75.2 vs 74.0 ms per image - the in-launch hand-off (write-through
        # publish, counter, poll, x staged after it) costs what the separate launch costs
        from . import _lib
        dev_i = self.device.index or 0
        self.fuse_rowops = frozenset({"rmsnorm"}) if _lib.get_option(dev_i, "llm_fuse_rmsnorm") else frozenset()
        self.prefill_attn_scalar = bool(_lib.get_option(dev_i, "prefill_attn_scalar"))
        # decode steps as ONE persistent launch per layer (psg_decode_layer; fp32 engines at Llama-2-7B width on a 256-CU
        # device, 13..24 rows): bit-identical to the launch chain, so every other shape - and the in-flight slots, since
        # only one persistent launch may run on a device at a time - simply keeps the chain
        self.persistent_layer = bool(_lib.get_option(dev_i, "decode_persistent"))
        # fp32s prompt pass: the split of a projection's operand and the un-scaling of its result inside the row kernels
        # next to it (psg_rmsnorm_split / psg_rope_kvwrite_scaled / psg_silu_mul_split; bit-identical, 7 launches per layer less)
        self.fuse_split = bool(_lib.get_option(dev_i, "llm_fuse_split"))
        # fp32s prompt pass: run each library product whole or in the column / row parts measured fastest (_plan_split_mm)
        self.plan_split = True
        # decode steps of 33..160 rows: psg_batch_gemm where it beats the library (option decode_batch_gemm)
        self.batch_gemm = bool(_lib.get_option(dev_i, "decode_batch_gemm"))
        self._w_pools = {}
        for L in self.layers:
            for k in ("wqkv", "wo", "wgu", "wdown"):
                self._w_pools.setdefault(tuple(L[k].shape), []).append(L[k])
        self._w_pools.setdefault(tuple(self.lm_head.shape), []).append(self.lm_head)
        self._dl_ws = {}
        self._dl_host_buf = None
        self.use_graph = True            # capture the batched decode in a HIP graph (per input shape)
        self.early_exit_chunk = 4        # natural-EOS decode: steps per graph between "all pairs done?" checks
        self.last_replays = 0
        self._graphs = collections.OrderedDict()   # input shape -> captured decode graphs (LRU, at most max_graphs)
        self.max_graphs = 8              # a graph owns its KV caches (~0.7 GB at K = 20 for Llama-2-7B); `forward` + two submit slots
        hd = m.head_dim
        # rotary tables as HF builds them (HF-LL:115-128): inv_freq and the outer product in fp32 on the
        # host, cos/sin per position; the kernels index them by position
        inv_freq = 1.0 / (m.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = torch.arange(4096, dtype=torch.float32)[:, None] * inv_freq[None, :]
        self.rope = (ang.cos().contiguous().to(self.device), ang.sin().contiguous().to(self.device))

    def linear_split(self, x, ws, w=None):
        """x [rows, K] fp32 @ w.T as ONE fp16 matrix-core GEMM over 3K: [xh | xh | xl] . [wh | wl | wh]^T with fp32
        accumulation, rows and columns rescaled by their powers of two afterwards (exact).  Products of fp16 values
        are exact in fp32, so what is lost against an fp32 GEMM is the xl.wl term and the split residuals: ~7e-7
        relative per product (fp32 rounds each product to 6e-8), at 3/16 of the fp32 matrix time."""
        if self.row_invariant and ws[0].shape[0] % 256 == 0 and ws[0].shape[1] % 64 == 0:
            if self.split_i2 and w is not None and w.shape[1] % 32 == 0:
                # round 6: interleaved hi / lo images, the three products from one staging (psg_dense_gemm_split)
                w2 = self._i2_weight(w)
                a2, inv_r = ops.split_f16i2(x)
                return ops.dense_gemm_split(a2, w2[0], None, inv_r, w2[1], tile="256x256")
            a3, inv_r = ops.split_f16x3(x)
            return ops.dense_gemm(a3, ws[0], None, out_dtype=torch.float32, row_scale=inv_r, col_scale=ws[1])
        a3, inv_r = ops.split_f16x3(x)
        y = _split_mm(a3, ws[0], _plan_split_mm(a3.shape[0], ws[0]) if self.plan_split else None)
        return ops.scale_rows_cols(y, inv_r, ws[1])

    def set_projection(self, weight, bias):
        """Replaces the packed copy of `language_projection` (a checkpoint loaded after the engine was built, an optimizer
        step in training mode) TOGETHER with everything derived from it: the fp32s split image and the interleaved image
        of the row-invariant path, whose cache is keyed by the tensor's address - which the allocator hands to the new
        tensor as soon as the old one is released."""
        self._i2_w.pop(self.proj_w.data_ptr(), None)
        self.proj_w = weight.to(device=self.device, dtype=self.dtype).contiguous()
        self.proj_b = bias.to(device=self.device, dtype=self.dtype).contiguous()
        self._i2_w.pop(self.proj_w.data_ptr(), None)
        self.proj_s = ops.split_f16x3(self.proj_w, weights=True) if self.prefill_split else None

    def _i2_weight(self, w):
        """The interleaved hi / lo image of a projection weight (+ its rows' inverse scales), made at first use: only the
        row-invariant path (the dealt decodes of a pair-sharded job) reads it - 4 bytes per weight next to the fp32 tensor."""
        ent = self._i2_w.get(w.data_ptr())
        if ent is None:
            ent = self._i2_w[w.data_ptr()] = ops.split_f16i2(w)
        return ent

    def _skinny_fits(self, x, w):
        """The weight-streaming kernel keeps the rows' K slice in LDS beside its weight rings: the fp32 kernel takes 32 rows up
        to K = 11776 and 20 rows up to K = 20480 (every Llama-2-7B / 13B shape at the reference's 20 selected pairs).  A wider
        model falls through to the library GEMM for that projection (exact fp32, not batch-invariant) instead of failing."""
        key = (x.shape[0], tuple(w.shape), x.dtype)
        ok = self._skinny_ok.get(key)
        if ok is None:
            try:
                ops.skinny_gemm_plan(x.shape[0], w.shape[0], w.shape[1], x.dtype, self.device)
                ok = True
            except PsgHipError:
                ok = False
            self._skinny_ok[key] = ok
        return ok

    def linear(self, x, w, ws=None, decode=False):
        """Bias-free projection.  Decode-step shapes (<= 32 rows) use the hand-written weight-streaming kernel - in
        the 16-bit modes and in the fp32 mode (the reference's own precision, V4:99-100) alike; the prompt pass goes
        through hipBLASLt; decode steps of 33..160 rows (several images' pairs, 16-bit modes) through whichever of
        psg_batch_gemm's variants and the library was measured fastest for the shape (_plan_batch_mm)."""
        if (self.use_skinny and x.shape[0] <= 32 and x.dtype == w.dtype and w.shape[0] % 16 == 0
                and w.shape[1] % 64 == 0 and w.shape[1] >= 256 and self._skinny_fits(x, w)):
            wh = self._w16.get(w.data_ptr()) if x.dtype == torch.float32 else None
            if wh is not None and self.prefill_split:         # fp32s: two fp16 products of the split rows, 2 bytes per weight
                x2, inv = ops.split_f16x2(x)
                return ops.split_gemm_w16(x2, inv, wh)
            if wh is not None:                                # exact fp32: the same f32 instructions on the widened weight
                return ops.skinny_gemm_w16(x, wh)
            return ops.skinny_gemm(x, w)          # fp32 split-K partials, reduced by the consumer kernel
        if (decode and self.batch_gemm and self.use_skinny and 32 < x.shape[0] <= 160 and x.dtype == w.dtype
                and x.dtype in (torch.bfloat16, torch.float16) and w.shape[0] % 16 == 0 and w.shape[1] % 64 == 0):
            plan = _plan_batch_mm(x, self._w_pools.get(tuple(w.shape), [w]))
            if plan[0] == "own":
                return ops.batch_gemm(x, w, plan[1], plan[2])
        if (self.plan_split and x.shape[0] >= 1024 and x.dtype == w.dtype and x.dtype in (torch.bfloat16, torch.float16)
                and x.dim() == 2 and x.is_contiguous()):
            # the 16-bit prompt pass of several images (forward_batch: 2 / 4 / 8 x 960 rows): the library's pick for the
            # whole product against its column / row parts (_plan_split_mm; 7-14 % per layer at 1920-7680 rows, nothing at 960)
            return _split_mm(x, w, _plan_split_mm(x.shape[0], w, None, pool=self._w_pools.get(tuple(w.shape))), None)
        if decode and x.dtype == torch.float32 and x.shape[0] > 32:
            # fp32 engines, decode steps of 33..160 rows (several images' pairs, head.forward_batch - BASELINE C5's batch of 8
            # at the reference's precision): the weight leaves HBM ONCE per step for all rows.  fp16-valued weights of an
            # fp32s engine: the two planes of the split rows [xh; xl] against the fp16 copy in ONE library product with an
            # fp32 result (2 bytes per weight, 2^-22 per product - the arithmetic of psg_split_gemm_w16); generic fp32
            # weights: the library SGEMM, exact (4 bytes per weight) - NOT the three-segment split product of the prompt
            # pass, which would stream 6 bytes per weight for flops a 160-row step does not need to save
            wh = self._w16.get(w.data_ptr()) if self.prefill_split else None
            if wh is not None and w.shape[1] % 8 == 0:
                a2, inv = ops.split_f16x2(x)
                y2 = torch.mm(a2.view(2 * x.shape[0], x.shape[1]), wh.t(), out_dtype=torch.float32)
                ones = self._ones.get(w.shape[0])
                if ones is None:
                    ones = self._ones[w.shape[0]] = torch.ones(w.shape[0], device=self.device, dtype=torch.float32)
                return ops.Scaled(y2.view(2, x.shape[0], w.shape[0]), inv, ones).dense()
            return F.linear(x, w)
        if ws is not None and x.dtype == torch.float32:
            return self.linear_split(x, ws, w)
        return F.linear(x, w)

    def logits(self, h):
        """lm_head.  <= 32 rows: the weight-streaming kernel (fp32 split-K partials, summed inside the greedy step);
        more rows (several images' pairs decoded together): the library GEMM with an fp32 result, so that the greedy
        argmax sees unrounded logits on this path too (exact_argmax)."""
        out = self.linear(h, self.lm_head, decode=True)
        if isinstance(out, ops.Partials) or not self.exact_argmax or h.dtype == torch.float32:
            return out
        if self._mm_out_dtype is None:
            try:
                torch.mm(h[:1], self.lm_head[:16].t(), out_dtype=torch.float32)
                self._mm_out_dtype = True
            except (TypeError, RuntimeError):
                self._mm_out_dtype = False
        if self._mm_out_dtype:
            return torch.mm(h, self.lm_head.t(), out_dtype=torch.float32)
        return out.float()

    # ---- one pass over `rows` token rows -------------------------------------------------------
    def _forward(self, resid, tok_pair, tok_pos, kc, vc, ctx_len, decode=False, prefill_shape=None, rope_pos=None,
                 keep_rows=None):
        """resid [rows, D] is updated in place (residual stream); returns final-norm hidden [rows, D].
        decode=True: every row is the newest token of its pair -> fused rotary + KV append + attention.
        prefill_shape=(pairs, rows_per_pair): the rows are a pair-major prompt batch -> matrix-core attention
        (bf16, <= 64 rows per pair); otherwise the scalar cache-attention kernel.
        rope_pos int32 [rows]: rotary positions when they differ from the cache slots `tok_pos` (training).
        keep_rows int32 [k]: the caller reads only these rows of the result (prompt pass: the last token of every
        pair).  The last layer then writes K/V for every row (the cache needs them) and runs everything behind the
        attention - output projection, norms, MLP - on those k rows only; returns [k, D]."""
        m = self.cfg.llm
        rows, D = resid.shape
        if (self.prefill_split and self.fuse_split and not self.row_invariant and not decode and prefill_shape is not None
                and rope_pos is None
                and rows > 32 and m.head_dim == 128 and prefill_shape[1] <= 64 and not self.prefill_attn_scalar
                and m.inter <= 16384 and D <= 8192):
            return self._forward_split(resid, tok_pair, tok_pos, kc, vc, ctx_len, prefill_shape, keep_rows)
        n = torch.empty((rows, D), device=self.device, dtype=self.dtype)
        ops.rmsnorm(resid, None, self.layers[0]["ln1"], m.rms_eps, n)
        q = torch.empty_like(n)
        att = torch.empty_like(n)
        act = torch.empty((rows, m.inter), device=self.device, dtype=self.dtype)
        mfma_prefill = (prefill_shape is not None and m.head_dim == 128 and prefill_shape[1] <= 64
                        and not self.prefill_attn_scalar)
        fused_rope = mfma_prefill and self.dtype in (torch.bfloat16, torch.float16)       # rotary + cache write in the launch
        for l, L in enumerate(self.layers):
            qkv = self.linear(n, L["wqkv"], L.get("wqkv_s"), decode=decode)
            if decode:
                ops.decode_attn(qkv, tok_pair, tok_pos, self.rope, m.heads, m.head_dim, ctx_len, kc[l], vc[l], att)
            elif fused_rope and isinstance(qkv, torch.Tensor) and rope_pos is None:
                ops.prefill_attn_rope(qkv, tok_pos, self.rope, prefill_shape[0], prefill_shape[1], m.heads, m.head_dim,
                                      ctx_len, kc[l], vc[l], att)
            else:
                ops.rope_kvwrite(qkv, tok_pair, tok_pos, self.rope, m.heads, m.head_dim, ctx_len, q, kc[l], vc[l],
                                 rope_pos=rope_pos)
                if mfma_prefill:
                    ops.prefill_attn(q, kc[l], vc[l], tok_pos, prefill_shape[0], prefill_shape[1], m.heads, m.head_dim,
                                     ctx_len, att)
                else:
                    ops.llm_attn(q, kc[l], vc[l], tok_pair, tok_pos, m.heads, m.head_dim, ctx_len, att)
            if keep_rows is not None and l == len(self.layers) - 1:
                k = keep_rows.numel()
                att_k, resid_k = torch.empty((k, D), device=self.device, dtype=self.dtype), torch.empty(
                    (k, D), device=self.device, dtype=resid.dtype)
                ops.gather_rows(att, keep_rows, att_k)
                ops.gather_rows(resid, keep_rows, resid_k)
                att, resid = att_k, resid_k
                n = torch.empty_like(att)
                act = torch.empty((k, m.inter), device=self.device, dtype=self.dtype)
            o = self.linear(att, L["wo"], L.get("wo_s"), decode=decode)
            ops.rmsnorm(resid, o, L["ln2"], m.rms_eps, n)                      # resid += o ; n = norm(resid)
            gu = self.linear(n, L["wgu"], L.get("wgu_s"), decode=decode)
            ops.silu_mul(gu, act)
            d = self.linear(act, L["wdown"], L.get("wdown_s"), decode=decode)
            nxt = self.layers[l + 1]["ln1"] if l + 1 < len(self.layers) else self.final_norm
            ops.rmsnorm(resid, d, nxt, m.rms_eps, n)                           # resid += d ; n = norm(resid)
        return n

    def _forward_split(self, resid, tok_pair, tok_pos, kc, vc, ctx_len, prefill_shape, keep_rows):
        """`_forward` for the prompt pass of the fp32s mode with the operand splits and result scalings fused into the row
        kernels: RMSNorm and SwiGLU write [hi | hi | lo] fp16 segments, every projection result stays raw (`ops.Scaled`)
        until its reader applies the scales while loading.  Same arithmetic as `_forward` + linear_split, bit for bit."""
        m = self.cfg.llm
        rows, D = resid.shape
        plan = _plan_split_mm if self.plan_split else (lambda r, w, k3=False: None)
        if self._w16_all:
            return self._forward_split_w16(resid, tok_pair, tok_pos, kc, vc, ctx_len, prefill_shape, keep_rows, plan)
        mm = lambda a3, ws, k3=False: _split_mm(a3, ws[0], plan(a3.shape[0], ws[0], k3=k3))     # noqa: E731
        a3, inv_r = ops.rmsnorm_split(resid, None, self.layers[0]["ln1"], m.rms_eps)
        q = torch.empty((rows, D), device=self.device, dtype=torch.float32)
        att = torch.empty_like(q)
        n = None
        for l, L in enumerate(self.layers):
            qkv = ops.Scaled(mm(a3, L["wqkv_s"]), inv_r, L["wqkv_s"][1])
            ops.rope_kvwrite_scaled(qkv, tok_pair, tok_pos, self.rope, m.heads, m.head_dim, ctx_len, q, kc[l], vc[l])
            ops.prefill_attn(q, kc[l], vc[l], tok_pos, prefill_shape[0], prefill_shape[1], m.heads, m.head_dim, ctx_len, att)
            last = l == len(self.layers) - 1
            if keep_rows is not None and last:
                k = keep_rows.numel()
                att_k = torch.empty((k, D), device=self.device, dtype=torch.float32)
                resid_k = torch.empty((k, D), device=self.device, dtype=torch.float32)
                ops.gather_rows(att, keep_rows, att_k)
                ops.gather_rows(resid, keep_rows, resid_k)
                att, resid = att_k, resid_k
            a3o, inv_o = ops.split_f16x3(att)                   # a row's maximum spans all heads: stays a kernel of its own
            o = ops.Scaled(mm(a3o, L["wo_s"], k3=True), inv_o, L["wo_s"][1])          # read by rmsnorm_split: may be K slices
            a3, inv_r = ops.rmsnorm_split(resid, o, L["ln2"], m.rms_eps)
            gu = ops.Scaled(mm(a3, L["wgu_s"]), inv_r, L["wgu_s"][1])
            a3a, inv_a = ops.silu_mul_split(gu, m.inter)
            d = ops.Scaled(mm(a3a, L["wdown_s"], k3=True), inv_a, L["wdown_s"][1])
            if last:                                            # the lm_head reads fp32 rows
                n = torch.empty_like(resid)
                ops.rmsnorm(resid, d.dense(), self.final_norm, m.rms_eps, n)
            else:
                a3, inv_r = ops.rmsnorm_split(resid, d, self.layers[l + 1]["ln1"], m.rms_eps)
        return n

    def _forward_split_w16(self, resid, tok_pair, tok_pos, kc, vc, ctx_len, prefill_shape, keep_rows, plan):
        """`_forward_split` when the LLM's matrices are fp16 values (`self._w16`): a weight has no low part, so a product is
        ONE library GEMM of the two-plane operand [xh; xl] (2 x rows rows) against the fp16 weight over K - two thirds of
        the three-segment form's flops, and no split copy of the weights at all; its [2, rows, N] result is summed by the
        reader (`slices`), the only scale left is the row's."""
        m = self.cfg.llm
        rows, D = resid.shape
        wh = self._w16
        ones = self._ones

        def one(n):
            t = ones.get(n)
            if t is None:
                t = ones[n] = torch.ones(n, device=self.device, dtype=torch.float32)
            return t

        def mm(a2, w, k3=False):                               # a2 [2, r, K] -> raw product slices [S, r, N]
            r = a2.shape[1]
            w16 = wh[w.data_ptr()]
            y = _split_mm(a2.view(2 * r, a2.shape[2]), w16, plan(2 * r, w16, k3=k3))
            return y.view(-1, r, w16.shape[0])

        a2, inv_r = ops.rmsnorm_split(resid, None, self.layers[0]["ln1"], m.rms_eps, planes=2)
        q = torch.empty((rows, D), device=self.device, dtype=torch.float32)
        att = torch.empty_like(q)
        n = None
        for l, L in enumerate(self.layers):
            qkv = ops.Scaled(mm(a2, L["wqkv"]), inv_r, one(3 * D))
            ops.rope_kvwrite_scaled(qkv, tok_pair, tok_pos, self.rope, m.heads, m.head_dim, ctx_len, q, kc[l], vc[l])
            ops.prefill_attn(q, kc[l], vc[l], tok_pos, prefill_shape[0], prefill_shape[1], m.heads, m.head_dim, ctx_len, att)
            last = l == len(self.layers) - 1
            if keep_rows is not None and last:
                k = keep_rows.numel()
                att_k = torch.empty((k, D), device=self.device, dtype=torch.float32)
                resid_k = torch.empty((k, D), device=self.device, dtype=torch.float32)
                ops.gather_rows(att, keep_rows, att_k)
                ops.gather_rows(resid, keep_rows, resid_k)
                att, resid = att_k, resid_k
            a2o, inv_o = ops.split_f16x2(att)
            o = ops.Scaled(mm(a2o, L["wo"], k3=True), inv_o, one(D))
            a2, inv_r = ops.rmsnorm_split(resid, o, L["ln2"], m.rms_eps, planes=2)
            gu = ops.Scaled(mm(a2, L["wgu"]), inv_r, one(2 * m.inter))
            a2a, inv_a = ops.silu_mul_split(gu, m.inter, planes=2)
            d = ops.Scaled(mm(a2a, L["wdown"], k3=True), inv_a, one(D))
            if last:
                n = torch.empty_like(resid)
                ops.rmsnorm(resid, d.dense(), self.final_norm, m.rms_eps, n)
            else:
                a2, inv_r = ops.rmsnorm_split(resid, d, self.layers[l + 1]["ln1"], m.rms_eps, planes=2)
        return n

    def _can_persist(self, rows, slot):
        m = self.cfg.llm
        return (self.persistent_layer and self.use_skinny and slot == 0 and self.dtype == torch.float32
                and ops.decode_layer_supported(rows, m.hidden, m.inter, m.heads, self.dtype, self.device))

    def _decode_step_persistent(self, st, counters):
        """One decode step on psg_decode_layers: ONE launch for the whole stack of decoder layers (chained inside the
        launch), then the final RMSNorm and the lm_head as in the chain.
        counters: int32 [layers * ops.decode_layer_counters()], zeroed."""
        m = self.cfg.llm
        x = st["x"]
        K = x.shape[0]
        ent = self._dl_ws.get(K)
        if ent is None:
            ws, _ = ops.decode_layer_workspace(K, m.hidden, m.inter, self.device)
            ent = self._dl_ws[K] = (ws, torch.empty((2, 16, K, m.hidden), device=self.device, dtype=torch.float32))
        ws, dparts = ent
        if "dl_table" not in st:                               # per decode state: the table holds its KV-cache pointers
            # (pinned staging + an async copy: capturable as a memcpy node, replayed with the graph)
            rows = ops.decode_layer_table(self.layers, st["kc"], st["vc"], device="cpu")
            host = self._dl_host_buf if self._dl_host_buf is not None else torch.empty_like(rows).pin_memory()
            self._dl_host_buf = None                           # (allocated by generate() BEFORE the capture began)
            host.copy_(rows)
            st["dl_table_host"] = host
            st["dl_table"] = torch.empty(host.shape, dtype=torch.int64, device=self.device)
            st["dl_table"].copy_(host, non_blocking=True)
        delta = ops.decode_layers(x, None, st["dl_table"], len(self.layers), st["dec_pair"], st["dec_pos"], self.rope, m.heads,
                                  st["ctx_len"], m.rms_eps, m.inter, ws, counters, dparts)
        n = torch.empty_like(x)
        ops.rmsnorm(x, delta, self.final_norm, m.rms_eps, n)
        return self.logits(n)

    def _can_w16(self, rows):
        """fp32s decode steps over fp16-valued weights: every projection as psg_split_gemm_w16, the row kernels writing
        its two-plane operand directly (`_decode_step_w16`)."""
        m = self.cfg.llm
        return (self._w16_all and self.prefill_split and self.use_skinny and self.fuse_split and rows <= 32
                and self.dtype == torch.float32 and m.hidden % 64 == 0 and m.inter % 64 == 0 and m.inter <= 16384
                and m.hidden <= 8192 and m.vocab % 16 == 0)

    def _decode_step_w16(self, st):
        """One decode step of the fp32s mode when the LLM's matrices are fp16 values (`self._w16`): 2 bytes per weight
        from HBM, two fp16 products per projection (high and low part of the fp32 rows) on the 16-bit matrix cores."""
        m = self.cfg.llm
        x, wh = st["x"], self._w16
        K, D = x.shape
        att = torch.empty((K, D), device=self.device, dtype=torch.float32)
        a2, inv = ops.rmsnorm_split2(x, None, self.layers[0]["ln1"], m.rms_eps)
        for l, L in enumerate(self.layers):
            qkv = ops.split_gemm_w16(a2, inv, wh[L["wqkv"].data_ptr()])
            ops.decode_attn(qkv, st["dec_pair"], st["dec_pos"], self.rope, m.heads, m.head_dim, st["ctx_len"], st["kc"][l],
                            st["vc"][l], att)
            # a row's maximum spans all heads / 11 008 columns: a split launch of its own behind attention and SwiGLU.  Folding
            # it into the producers needs a rendezvous of the row's 32 / 11 workgroups - built and measured in round 6
            # (profiles/r06_split2_rendezvous_ab.txt): inside a graph the launch costs 2.6 / 3.5 us, the rendezvous 9.3 / 3.4
            a2o, invo = ops.split_f16x2(att)
            o = ops.split_gemm_w16(a2o, invo, wh[L["wo"].data_ptr()])
            a2, inv = ops.rmsnorm_split2(x, o, L["ln2"], m.rms_eps)
            gu = ops.split_gemm_w16(a2, inv, wh[L["wgu"].data_ptr()])
            act = torch.empty((K, m.inter), device=self.device, dtype=torch.float32)
            ops.silu_mul(gu, act)
            a2a, inva = ops.split_f16x2(act)
            d = ops.split_gemm_w16(a2a, inva, wh[L["wdown"].data_ptr()])
            nxt = self.layers[l + 1]["ln1"] if l + 1 < len(self.layers) else self.final_norm
            a2, inv = ops.rmsnorm_split2(x, d, nxt, m.rms_eps)
        return ops.split_gemm_w16(a2, inv, wh[self.lm_head.data_ptr()])

    def _can_fuse(self, rows):
        m = self.cfg.llm
        D = m.hidden
        return (bool(self.fuse_rowops) and self.use_skinny and self.dtype in (torch.bfloat16, torch.float16)
                and self.resid_dtype == self.dtype and rows <= 32 and D in (1024, 4096) and m.inter >= 1024 and m.inter % 64 == 0
                and m.vocab >= 1024 and m.vocab % 16 == 0)

    def _decode_step_fused(self, st, sync):
        """One decode step with the row operations named in `fuse_rowops` folded into the projection that consumes
        them (same arithmetic, same order: bit-identical to the separate kernels).  Returns the lm_head partials.
        sync: int32 [>= 2 * (4 * layers + 1)], zeroed."""
        m = self.cfg.llm
        x = st["x"]                                            # residual stream [K, D], embedding of the new tokens
        K = x.shape[0]
        n = torch.empty_like(x)
        att = torch.empty_like(x)
        act = torch.empty((K, m.inter), device=self.device, dtype=self.dtype)
        f = self.fuse_rowops
        slot = [0]

        def words():
            w = sync[2 * slot[0]:2 * slot[0] + 2]
            slot[0] += 1
            return w

        def norm_proj(delta, ln, w):                           # x += delta ; n = norm(x) ; n @ w.T
            if "rmsnorm" in f:
                return ops.skinny_gemm_fused(ops.PSG_PRO_RMSNORM, n, w, words(), inp=delta, resid=x, norm_w=ln,
                                             eps=m.rms_eps)
            ops.rmsnorm(x, delta, ln, m.rms_eps, n)
            return ops.skinny_gemm(n, w)

        delta = None
        for l, L in enumerate(self.layers):
            qkv = norm_proj(delta, L["ln1"], L["wqkv"])
            ops.decode_attn(qkv, st["dec_pair"], st["dec_pos"], self.rope, m.heads, m.head_dim, st["ctx_len"],
                            st["kc"][l], st["vc"][l], att)
            o = ops.skinny_gemm(att, L["wo"])
            gu = norm_proj(o, L["ln2"], L["wgu"])
            ops.silu_mul(gu, act)
            delta = ops.skinny_gemm(act, L["wdown"])
        return norm_proj(delta, self.final_norm, self.lm_head)

    def build_inputs(self, pair_feature_rows, prompt_ids, prompt_len):
        """V4:294-301 for all K pairs.  pair_feature_rows [K*32, 768] (activation dtype),
        prompt_ids int32 [K, Tp] COMPACT (valid ids first, -1 after), prompt_len int32 [K].
        Returns X [K, 32+Tp, D]."""
        m = self.cfg.llm
        K, Tp = prompt_ids.shape
        nv = self.cfg.qformer.num_query
        X = torch.empty((K, nv + Tp, m.hidden), device=self.device, dtype=self.dtype)
        if self.row_invariant and self.proj_s is not None and m.hidden % 256 == 0 and pair_feature_rows.shape[1] % 64 == 0:
            if self.split_i2 and pair_feature_rows.shape[1] % 32 == 0:
                w2 = self._i2_weight(self.proj_w)
                a2, inv_r = ops.split_f16i2(pair_feature_rows.contiguous())
                vis = ops.dense_gemm_split(a2, w2[0], self.proj_b, inv_r, w2[1], tile="256x256").view(K, nv, m.hidden)
            else:
                a3, inv_r = ops.split_f16x3(pair_feature_rows.contiguous())
                vis = ops.dense_gemm(a3, self.proj_s[0], self.proj_b, out_dtype=torch.float32, row_scale=inv_r,
                                     col_scale=self.proj_s[1]).view(K, nv, m.hidden)
        else:
            vis = F.linear(pair_feature_rows, self.proj_w, self.proj_b).view(K, nv, m.hidden)
        X[:, :nv] = vis
        tok = torch.empty((K * Tp, m.hidden), device=self.device, dtype=self.dtype)
        ops.gather_rows(self.embed, prompt_ids.reshape(-1).contiguous(), tok)
        X[:, nv:] = tok.view(K, Tp, m.hidden)
        return X

    @torch.no_grad()
    def teacher_forcing_logits(self, X, seq_len, rope_pos, rows):
        """Training forward (V4:327-336: a plain `language_model(inputs_embeds, attention_mask)` call).
        X [K, S, D] COMPACT sequences (pads removed), seq_len int32 [K], rope_pos int32 [K*S] = each row's position
        in the reference's PADDED sequence (HF numbers positions 0..T-1 over pads in a plain forward), rows int32:
        flat row indices whose logits are wanted.  Returns logits [len(rows), vocab] in the activation dtype."""
        m = self.cfg.llm
        K, S, D = X.shape
        dev = self.device
        t = torch.arange(S, device=dev, dtype=torch.int32)[None, :].expand(K, -1)
        tok_pos = torch.where(t < seq_len[:, None].to(torch.int32), t, torch.full_like(t, -1)).reshape(-1).contiguous()
        tok_pair = torch.arange(K, device=dev, dtype=torch.int32)[:, None].expand(-1, S).reshape(-1).contiguous()
        kc = [torch.empty((K, m.heads, S, m.head_dim), device=dev, dtype=self.dtype) for _ in self.layers]
        vc = [torch.empty((K, m.heads, S, m.head_dim), device=dev, dtype=self.dtype) for _ in self.layers]
        resid = X.reshape(K * S, D).to(self.resid_dtype, copy=True)
        h = self._forward(resid, tok_pair, tok_pos, kc, vc, S, rope_pos=rope_pos.contiguous())
        h_rows = torch.empty((rows.numel(), D), device=dev, dtype=self.dtype)
        ops.gather_rows(h, rows.to(torch.int32).contiguous(), h_rows)
        return F.linear(h_rows, self.lm_head)

    @torch.no_grad()
    def generate(self, X, prompt_len, max_new_tokens=None, suppress_eos=False, return_first_logits=False, slot=0,
                 gate=None, defer=False):
        """Batched greedy decode.  X [K, 32+Tp, D]; prompt_len int32 [K] (# of prompt tokens).
        Returns tokens int32 [K, max_new] (device; -1 after a pair's EOS) and optionally the
        first-step logits [K, vocab].

        The decode is captured in HIP graphs per input shape and replayed: nothing in it depends on
        the host (selection, argmax, EOS flags and positions all live on the device).
          * suppress_eos (benchmark worst case): ONE graph = prefill + max_new-1 steps (~5000 launches);
          * natural EOS: the reference's per-pair `generate` stops at EOS (V4:305-312), and a relation string
            is a handful of tokens, so the steps are cut into graphs of `early_exit_chunk` steps and the
            replay stops as soon as every pair has emitted EOS (one 4-byte read-back per chunk).
        slot: graphs (with their KV caches and static buffers) are kept per slot, so that two generations - of two
        images on two HIP streams, `head.submit` - can be in flight at once; a slot is used by one stream at a time.
        gate: callable run between the prompt pass (+ first token) and the decode steps, which are then separate graphs -
        the caller makes the stream wait there for the previous image's decode (the prompt pass of image k+1 is
        matrix-core work that fits beside image k's HBM-bound decode steps; two decodes side by side only share the HBM).
        defer (natural EOS only): enqueue the prompt pass and the first chunk of steps WITHOUT a host wait and return a
        callable instead of the results; calling it replays the remaining chunks (with their "all done?" read-backs) and
        returns what the direct call returns.  `head.submit` hands it to the pending result, so that `result()` stays
        the only host wait of an image in flight."""
        max_new = self.cfg.max_new_tokens if max_new_tokens is None else max_new_tokens
        if not self.use_graph:
            outs = self._finish(self._generate_eager(X, prompt_len, max_new, suppress_eos, return_first_logits, slot=slot),
                                return_first_logits)
            return (lambda: outs) if defer else outs
        chunk = 0 if suppress_eos else int(self.early_exit_chunk)
        split = gate is not None and max_new > 1
        key = (tuple(X.shape), max_new, bool(suppress_eos), bool(return_first_logits), chunk, int(slot), split,
               bool(self.row_invariant))
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs.move_to_end(key)
        if ent is None:
            while len(self._graphs) >= self.max_graphs:        # least recently used shape: frees its graph pool
                # the evicted graph (or its KV pool) may still be replaying on another slot's stream
                torch.cuda.synchronize(self.device)
                self._graphs.popitem(last=False)
            Xs, ps = X.clone(), prompt_len.to(torch.int32).clone()
            if self.persistent_layer:                          # pinned staging of the layer table: not allocatable while capturing
                self._dl_host_buf = None
                pinned = torch.empty((len(self.layers), 8), dtype=torch.int64).pin_memory()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up: lazy library init must not be captured
                self._generate_eager(Xs, ps, max_new, suppress_eos, return_first_logits, slot=slot)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            if self.persistent_layer:
                self._dl_host_buf = pinned
            bounds = [max_new] if chunk <= 0 else list(range(chunk, max_new, chunk)) + [max_new]
            if split:                                          # the gate sits right behind the prompt pass + first token
                bounds = [1] + bounds
            bounds = sorted(set(bounds))
            graphs, st, lo = [], None, 0
            for hi in bounds:                                  # graph i runs steps [lo, hi); step 0 includes the prefill
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=graphs[0][0].pool() if graphs else None):
                    if st is None:
                        st = self._prefill(Xs, ps, max_new, suppress_eos, return_first_logits, slot=slot)
                        self._steps(st, 1, hi)
                    else:
                        self._steps(st, lo, hi)
                    all_done = st["done"].min() if (chunk > 0 and hi >= chunk) else None
                graphs.append((g, hi, all_done))
                lo = hi
            ent = self._graphs[key] = (graphs, Xs, ps, st)
        graphs, Xs, ps, st = ent
        Xs.copy_(X)
        ps.copy_(prompt_len)
        self.last_replays = 0
        cursor = [0]

        def advance(budget):
            """Replays graphs from the cursor on; budget = how many graphs may be enqueued without a read-back (None: run
            to the end, reading the all-done flag after every chunk)."""
            while cursor[0] < len(graphs):
                g, hi, all_done = graphs[cursor[0]]
                if budget is not None and cursor[0] >= budget:
                    return
                if cursor[0] == 1 and gate is not None:
                    gate()
                g.replay()
                cursor[0] += 1
                self.last_replays += 1
                if budget is None and hi < max_new and all_done is not None and int(all_done.item()) != 0:
                    break                                       # every pair has emitted EOS: the rest would be -1
            cursor[0] = len(graphs)

        def finish():
            advance(None)
            self._check_persistent(st)
            # the graph's static buffers are overwritten by the next replay: hand out copies
            fl = st["first_logits"]
            return self._finish((st["tokens"].clone(), None if fl is None else fl.clone()), return_first_logits)
        if defer and chunk > 0:
            advance(2 if split else 1)                          # prompt pass (+ gate) + the first chunk of steps: no host wait
            return finish
        outs = finish()
        return (lambda: outs) if defer else outs

    @staticmethod
    def _finish(outs, want_first):
        return outs if want_first else outs[0]

    def _check_persistent(self, st):
        """psg_decode_layer bounds every hand-off poll and reports a producer that never arrived through word
        PSG_DL_TIMEOUT of the launch's counter block (`ops.DL_TIMEOUT_WORD`).  Raises if any launch of this
        generation set it - its tokens are not to be trusted; the caller can switch `persistent_layer` off (the launch
        chain computes the same bits)."""
        for sync in st.get("dl_counters", ()):
            ncnt = ops.decode_layer_counters(self.device)
            if bool((sync.view(-1, ncnt)[:, ops.DL_TIMEOUT_WORD] != 0).any().item()):
                raise PsgHipError("psg_decode_layer: a hand-off poll timed out (cnt[PSG_DL_TIMEOUT] set); the tokens of this "
                                  "generation are invalid - disable option decode_persistent")

    def _generate_eager(self, X, prompt_len, max_new, suppress_eos, return_first_logits, slot=0):
        st = self._prefill(X, prompt_len, max_new, suppress_eos, return_first_logits, slot=slot)
        self._steps(st, 1, max_new)
        if not torch.cuda.is_current_stream_capturing():
            self._check_persistent(st)
        return st["tokens"], st["first_logits"]

    def _prefill(self, X, prompt_len, max_new, suppress_eos, return_first_logits, slot=0):
        """Prompt pass + first greedy token.  Returns the decode state (KV caches, token / flag buffers)."""
        m = self.cfg.llm
        K, maxlen, D = X.shape
        nv = self.cfg.qformer.num_query
        ctx_len = maxlen + max_new
        dev = self.device
        seq_len = (prompt_len.to(torch.int32) + nv)                                   # valid tokens per pair
        t = torch.arange(maxlen, device=dev, dtype=torch.int32)[None, :].expand(K, -1)
        tok_pos = torch.where(t < seq_len[:, None], t, torch.full_like(t, -1)).reshape(-1).contiguous()
        tok_pair = torch.arange(K, device=dev, dtype=torch.int32)[:, None].expand(-1, maxlen).reshape(-1).contiguous()
        # no zero fill (650 MB of stores per image for Llama-2-7B): every kernel reads only cache rows that were written
        kc = [torch.empty((K, m.heads, ctx_len, m.head_dim), device=dev, dtype=self.dtype) for _ in self.layers]
        vc = [torch.empty((K, m.heads, ctx_len, m.head_dim), device=dev, dtype=self.dtype) for _ in self.layers]
        resid = X.reshape(K * maxlen, D).to(self.resid_dtype, copy=True)
        last_rows = (torch.arange(K, device=dev, dtype=torch.int32) * maxlen + seq_len - 1).contiguous()
        h_last = self._forward(resid, tok_pair, tok_pos, kc, vc, ctx_len, prefill_shape=(K, maxlen), keep_rows=last_rows)
        logits = self.logits(h_last)
        first_logits = None
        if return_first_logits:
            first_logits = logits.reduce(self.dtype) if isinstance(logits, ops.Partials) else logits.to(self.dtype)
        tokens = torch.full((K, max_new), -1, device=dev, dtype=torch.int32)
        done = torch.zeros(K, device=dev, dtype=torch.int32)
        next_ids = torch.zeros(K, device=dev, dtype=torch.int32)
        dec_pos = (seq_len - 1).contiguous()                                           # greedy_step does += 1
        dec_pair = torch.arange(K, device=dev, dtype=torch.int32)
        sup = m.eos if suppress_eos else -1
        x = torch.empty((K, D), device=dev, dtype=self.resid_dtype)   # residual stream of the decode rows
        # the greedy step also writes the chosen token's embedding row into x: the next step's input, no gather launch
        ops.greedy_step(logits, 0, max_new, m.eos, sup, tokens, done, next_ids, dec_pos,
                        dtype=torch.float32 if self.exact_argmax else self.dtype, embed=self.embed, x_out=x)
        return dict(kc=kc, vc=vc, ctx_len=ctx_len, tokens=tokens, done=done, next_ids=next_ids, dec_pos=dec_pos,
                    dec_pair=dec_pair, sup=sup, x=x, max_new=max_new, first_logits=first_logits, slot=int(slot))

    def _steps(self, st, lo, hi):
        """Decode steps lo .. hi-1 (step s writes tokens[:, s])."""
        m = self.cfg.llm
        persist = hi > lo and self._can_persist(st["x"].shape[0], st.get("slot", 0))
        w16 = not persist and self._can_w16(st["x"].shape[0])
        fused = not persist and not w16 and self._can_fuse(st["x"].shape[0]) and hi > lo
        if persist:                                            # one counter block per layer launch, zeroed once per call
            per_step = ops.decode_layer_counters(self.device) * len(self.layers)
            sync = torch.zeros((hi - lo) * per_step, device=self.device, dtype=torch.int32)
            # kept in the decode state: a hand-off poll that gave up reports through the block's time-out word, which
            # `generate` reads back with the tokens (`_check_persistent`) - a dead producer must not yield silent garbage
            st.setdefault("dl_counters", []).append(sync)
        if fused:                                              # two counter words per fused launch, zeroed once per call
            per_step = 2 * (4 * len(self.layers) + 1)
            sync = torch.zeros((hi - lo) * per_step, device=self.device, dtype=torch.int32)
        for step in range(lo, hi):
            if persist:
                logits = self._decode_step_persistent(st, sync[(step - lo) * per_step:(step - lo + 1) * per_step])
            elif w16:
                logits = self._decode_step_w16(st)
            elif fused:
                logits = self._decode_step_fused(st, sync[(step - lo) * per_step:(step - lo + 1) * per_step])
            else:
                h = self._forward(st["x"], st["dec_pair"], st["dec_pos"], st["kc"], st["vc"], st["ctx_len"], decode=True)
                logits = self.logits(h)
            ops.greedy_step(logits, step, st["max_new"], m.eos, st["sup"], st["tokens"], st["done"], st["next_ids"],
                            st["dec_pos"], dtype=torch.float32 if self.exact_argmax else self.dtype, embed=self.embed,
                            x_out=st["x"])
