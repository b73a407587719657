"""Headline benchmark: object-pairs/sec of the relation-query + LMM-decode path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one synthetic image per rank: BASELINE.json config C3
(1024x1024, 50 masks = 2450 ordered pairs, relation Q-Former + existence head + top-20 selection +
batched greedy Llama-2-7B-shaped decode of 16 tokens), random-init weights, inputs already resident in HBM;
one `head(inputs)` call per step, its result awaited (wall time of ONE head call).

PRECISION OF THE HEADLINE: the reference runs this path in fp32 (V4:99-100 loads the LLM without a dtype) and the
north star's tolerance (logits 1e-3, argmax-exact) is an fp32 statement, so `value` is measured in the `fp32s` mode:
fp32 weights, activations, KV cache, norms, softmaxes and decode-step projections (26.4 GB of fp32 weights streamed
per decode step); the prompt pass's and the Q-Former's projections are split-fp16 products (x.w = xh.wh + xh.wl +
xl.wh on the 16-bit matrix cores, fp32 accumulation: 3e-7 relative, the library SGEMM's own error class).  The line's
`parity` block re-checks that mode against the CPU oracle in the same run.  The 16-bit figures of earlier rounds
(`mixed`: fp16 operands, outside the tolerance) and the exact-fp32 figures are sub-objects, as is BASELINE config C2
(`c2`).  `--workload rq` times config C2 alone.

N > 1 (one rank per GPU, RCCL; launched by torch.distributed.run - by the driver, or by this script itself
when `--gpus N` is given without a rendezvous in the environment): WEAK scaling.  The job is N images per step;
EVERY image's pairs are sharded over all N ranks, existence probabilities are all-gathered, every rank runs the
same deterministic top-K, the selected pair features are reduce-scattered to the image's decoding rank
(openpsg_amd/dist.py).  Per-rank work is that of one image, so `value` grows with N by construction; what the
multi-GPU line measures is the cost of the four collectives per step.  The line therefore also carries
`strong_scaling`: ONE BASELINE-C4 image (100 masks, 9900 pairs) for all N ranks, pairs sharded, the K decodes
dealt round-robin - its latency is bounded below by the 16 weight passes of the decode (no tensor parallelism in
scope), which is stated next to the number.

Round 6 additions to the line: `frozen_fp16_checkpoint` is timed exactly like the headline (same steps / warm-up, its
own roofline.image terms): the same mode over LLM matrices that hold fp16 VALUES, as the reference's frozen checkpoint does;
`c4_single_gpu` = BASELINE C4 (100 masks) on this one GPU; `batched_decode` = head.forward_batch at 4 / 8 images per step in
the headline mode (BASELINE C5's batch at the reference's precision); `parity.modes.*.decode_7b_32_layers` = the oracle's
UN-TRUNCATED 32-layer fp32 decode of a pair (the one cpu_baseline times) against the fp32s head on the same weights, both
weight kinds; multi-GPU lines add `strong_scaling_rq` (C4's sharded part alone) beside `strong_scaling`.

The JSON line also carries
  roofline     - the dominant kernel (skinny_gemm_f32_kernel: HBM stream of the fp32 LLM weights in the decode
                 steps), timed with HIP events on the launch stream right after the timed region with the
                 same weights and shapes (inside the region the decode is ONE graph replay, which has no
                 per-kernel events); profiles/ holds the rocprofv3 summary of the same command; `roofline.image` =
                 the whole image against its combined floor (15 weight passes at the HBM peak + the prompt pass's and
                 the relation query's FLOPs at the matrix peak of the dtype they run in);
  cpu_baseline - the CPU oracle (oracle/psg_oracle.py, a restatement of the reference's PyTorch path)
                 timed on this box's host cores on a bounded sample, rank 0, N = 1 only;
  parity       - the same scene through the fp32 verification mode (ms/step, pairs/s: the mode the 1e-3 claim is
                 made in); per 16-bit mode (bf16, fp16, mixed = fp16 operands + fp32 residual stream): ms/step of the
                 full path, deviation of the existence logits from the CPU oracle on ALL pairs of the scene, top-20
                 overlap, and the DECODE leg at Llama-2-7B width (2 full-width layers, the oracle's pair features
                 injected): first-step logit deviation and token-exact sequences against the oracle's greedy decode.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "object-pairs/sec (relation-query + LMM decode), 1024² img × 50 masks"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["full", "rq"], default="full")
    ap.add_argument("--objects", type=int, default=50)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--categories", type=int, default=133,
                    help="classes the synthetic objects draw from (default: all 133, SURVEY 8d; real images repeat "
                         "classes - fewer distinct prompts - which the relation query exploits)")
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--images-per-step", type=int, default=1,
                    help="single-GPU throughput mode: images per step whose selected pairs are decoded together")
    ap.add_argument("--no-batched", action="store_true", help="skip the secondary 4-images-per-step measurement")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="images in flight of the timed step (head.submit slots).  Default 1: one head call per step, its "
                         "result awaited (SURVEY 8d).  The two-in-flight figure is always reported as `two_in_flight`")
    ap.add_argument("--slot-priorities", default="0,-1",
                    help="HIP stream priorities of the in-flight slots (A/B: 0,0 = both slots in the normal queue pool)")
    ap.add_argument("--serialize-decodes", action="store_true",
                    help="pipelined step A/B: image k+1's decode steps wait for image k's (measured: no gain over --serial)")
    ap.add_argument("--serial", action="store_true", help="same as --in-flight 1 (kept for the profile scripts)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the 16-bit (mixed) sub-object")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 sub-object")
    ap.add_argument("--no-frozen16", action="store_true", help="skip the frozen-fp16-checkpoint sub-object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", choices=["fp32s", "fp32", "bf16", "fp16", "mixed"], default=None,
                    help="precision mode of the measured path.  fp32s (default of the full path): the reference's fp32 "
                         "arithmetic with split-fp16 products in the prompt pass and the Q-Former - inside the north "
                         "star's tolerance; fp32: exact everywhere; mixed = fp16 GEMM operands and KV cache, fp32 "
                         "accumulation, fp32 Llama residual stream (rounds 3-4's headline, outside the tolerance); bf16 "
                         "(default of --workload rq, BASELINE config 2 names it); fp16 (BASELINE config 5)")
    ap.add_argument("--no-untruncated", action="store_true",
                    help="cpu_baseline: skip the un-truncated 32-layer fp32 decode of one pair (27 GB of host memory)")
    ap.add_argument("--llm-values", choices=["fp32", "fp16"], default="fp32",
                    help="fp16: the LLM's fp32 matrices hold fp16 VALUES (a frozen fp16 checkpoint upcast on load, as the "
                         "reference's Llama-2-7b-hf is: V4:99-100, configs/psg/baseline_v4_ov.py:61-65); the engine then "
                         "streams them as fp16 in the decode steps")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--one-phase", action="store_true",
                    help="last Q-Former layer for all 33 rows of every pair (A/B against the default cls-first path)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the one-image strong-scaling figures")
    ap.add_argument("--no-c4", action="store_true", help="N = 1: skip the C4 (100 masks) single-GPU point")
    ap.add_argument("--pair-chunk", type=int, default=0, help="pairs per Q-Former pass (0: the head's default)")
    a = ap.parse_args()
    if a.dtype is None:
        a.dtype = "fp32s" if a.workload == "full" else "bf16"
    if a.serial:
        a.in_flight = 1
    a.serial = a.in_flight <= 1
    return a


def self_launch(a):
    """`python bench.py --gpus N` without a rendezvous in the environment: re-exec under torch.distributed.run,
    one rank per GPU.  Fails loudly when the node has fewer GPUs than asked for."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but this node exposes {n_dev} GPU(s); nothing was measured")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def setup_head(a, dev):
    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import make_weights_device
    cfg = PSGConfig(qformer=QFormerConfig(), llm=LlamaConfig(layers=a.llm_layers), max_object_num=a.objects)
    dtype = getattr(a, "dtype_override", None) or a.dtype
    tdt = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp32s": torch.float32}.get(dtype, torch.float16)
    vals = torch.float16 if (getattr(a, "llm_values", "fp32") == "fp16" and tdt == torch.float32) else None
    w = make_weights_device(cfg, 0, dev, llm_dtype=tdt, with_llm=a.workload == "full", llm_values=vals)
    head = RelationTransformerHeadV4(dtype=dtype, device=str(dev), tokenizers="word", max_object_num=a.objects,
                                     llm_config=cfg.llm, on_parse_error="skip", suppress_eos=True,
                                     cls_first=not a.one_phase,
                                     slot_priorities=tuple(int(p) for p in getattr(a, "slot_priorities", "0,-1").split(",")))
    head.load_weights(w)
    if getattr(a, "pair_chunk", 0) > 0:
        head.pair_chunk = a.pair_chunk
    del w
    torch.cuda.empty_cache()
    return head


def measure_decode_gemm(head, K):
    """Dominant kernel: one decode step's 129 weight-streaming launches (4 per layer + lm_head) with the
    engine's real weights; HIP events on the launch stream.  Returns (bytes/launch, seconds/launch, n)."""
    from openpsg_amd import ops
    eng = head.llm_engine
    m = head.cfg.llm
    dev = eng.device
    x_d = torch.randn(K, m.hidden, device=dev).to(eng.dtype)
    x_i = torch.randn(K, m.inter, device=dev).to(eng.dtype)
    mats = []
    for L in eng.layers:
        mats += [(x_d, L["wqkv"]), (x_d, L["wo"]), (x_d, L["wgu"]), (x_i, L["wdown"])]
    mats.append((x_d, eng.lm_head))
    nbytes = sum(w.numel() * w.element_size() for _, w in mats)

    def run():
        for x, w in mats:
            ops.skinny_gemm(x, w)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 1e3)
    ts.sort()
    return nbytes / len(mats), ts[len(ts) // 2] / len(mats), len(mats)      # median pass, averaged over its launches


def measure_decode_gemm_w16(head, K):
    """`measure_decode_gemm` for an fp32s engine whose LLM matrices are fp16 values: psg_split_gemm_w16 over the fp16 copies."""
    from openpsg_amd import ops
    eng = head.llm_engine
    m = head.cfg.llm
    dev = eng.device
    xd = ops.split_f16x2(torch.randn(K, m.hidden, device=dev))
    xi = ops.split_f16x2(torch.randn(K, m.inter, device=dev))
    wh = eng._w16
    mats = []
    for L in eng.layers:
        mats += [(xd, wh[L["wqkv"].data_ptr()]), (xd, wh[L["wo"].data_ptr()]), (xd, wh[L["wgu"].data_ptr()]),
                 (xi, wh[L["wdown"].data_ptr()])]
    mats.append((xd, wh[eng.lm_head.data_ptr()]))
    nbytes = sum(w.numel() * w.element_size() for _, w in mats)

    def run():
        for (x2, inv), w in mats:
            ops.split_gemm_w16(x2, inv, w)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 1e3)
    ts.sort()
    return nbytes / len(mats), ts[len(ts) // 2] / len(mats), len(mats)


def relation_query_flops(N, L, T, cls_first=False, selected=20):
    """FLOPs the relation-query stage has to execute for one image: SURVEY 8d's per-layer formula with the work
    whose result is never read or is identical for every pair left out (V4:185 slices the output to the 33 query
    rows, so in the last layer the text rows need K/V only; the Q/K/V projection of the 33 query rows entering
    layer 0 is the same for all pairs and is computed once).  Plus the per-image patch embedding and shared K/V.
    Counted PER PAIR: the head additionally computes the prompt-only share once per distinct prompt, which this
    count does not subtract - the figure is algorithmic work per second, not executed instructions."""
    S, H, F = 33 + T, 768, 3072
    cross = 2 * 33 * H * H + 4 * 33 * L * H + 2 * 33 * H * H
    first = 2 * T * H * 3 * H + 4 * S * S * H + 2 * S * H * H + cross + 4 * 33 * H * F + 4 * T * H * F
    last = 2 * S * H * 2 * H + 2 * 33 * H * H + 4 * 33 * S * H + 2 * 33 * H * H + cross + 4 * 33 * H * F
    per_image = 2 * 33 * H * 3 * H + 2 * L * 65536 * 256 + 2 * 2 * 2 * L * 256 * H
    if cls_first:
        # last layer: K/V of every row, then ONLY the cls row of every pair (existence head, V4:206-209) and the 33
        # rows of the selected pairs (V4:215, 235-237)
        # selection phase in the input space (psg_qformer_cls_attn_input): no K | V projection of all rows; per pair
        # the cls query (2 H H), its projection back through W_k (2 H H), 12 heads x S keys x 768 features for the
        # scores and as many for the weighted row means (4 * 12 * S * H), the means through W_v (2 H H)
        last_cls = (2 * H * H + 2 * H * H + 4 * 12 * S * H + 2 * H * H) + 2 * H * H + (
            2 * H * H + 4 * L * H + 2 * H * H) + 4 * H * F
        return N * N * (first + last_cls) + selected * last + per_image
    return N * N * (first + last) + per_image


def relation_query_flops_executed(N, L, T, U, selected=20):
    """What the cls-first, prompt-deduplicated relation query actually executes for one image (dense-equivalent for the
    cross-attention contraction, whose masked key tiles are skipped): layer 0's self-attention block, text FFN and
    cross-attention query projection once per DISTINCT prompt (U of them), the rest of layer 0 and the selection phase of
    the last layer per pair, the full last layer for the selected pairs."""
    S, H, F = 33 + T, 768, 3072
    per_prompt = 2 * T * H * 3 * H + 4 * S * S * H + 2 * S * H * H + 4 * T * H * F + 2 * 33 * H * H
    per_pair0 = 4 * 33 * L * H + 2 * 33 * H * H + 4 * 33 * H * F
    last_cls = (2 * H * H + 2 * H * H + 4 * 12 * S * H + 2 * H * H) + 2 * H * H + (2 * H * H + 4 * L * H + 2 * H * H) + 4 * H * F
    last_full = 2 * S * H * 3 * H + 4 * S * S * H + 2 * 33 * H * H + (2 * 33 * H * H + 4 * 33 * L * H + 2 * 33 * H * H) + 4 * 33 * H * F
    per_image = 2 * 33 * H * 3 * H + 2 * L * 65536 * 256 + 2 * 2 * 2 * L * 256 * H
    return U * per_prompt + N * N * (per_pair0 + last_cls) + selected * last_full + per_image


def host_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(a, scene_cpu):
    """The CPU oracle on this box's host cores: the relation query over ALL pairs of the scene (3 repetitions, median),
    the LMM decode on a bounded sample (serial batch-1 decodes as V4:293-312; 2 and 4 full-width fp32 layers
    extrapolated linearly to 32, plus ONE un-truncated 32-layer decode when the host has the memory).  Returns the
    JSON object and what the parity block reuses (weights, cfg, oracle logits / features / greedy decodes)."""
    from openpsg_amd.config import LlamaConfig, PSGConfig, QFormerConfig
    from openpsg_amd.weights import _std_for, llm_shapes, make_weights_numpy
    from oracle import psg_oracle as O
    from tests import helpers as H
    N = a.objects
    B = N * N
    lay = (2, 4)                                                    # llm_truncate_num in {2, 4}
    cfg = PSGConfig(qformer=QFormerConfig(), llm=LlamaConfig(layers=max(lay)), max_object_num=N)
    w = make_weights_numpy(cfg, seed=1, with_llm=a.workload == "full")
    ids, tmask = H.qformer_prompts(scene_cpu)
    host_cores, host_model = host_info()
    n_dec = 4                                                       # pairs decoded by the oracle at 2 layers (parity)
    with torch.no_grad():
        # torch's CPU kernels oversubscribe badly on a many-core host (256 threads were 20x slower
        # than 8 on these shapes): pick the fastest thread count on a small probe, report it as `cores`.
        probe_patches = torch.randn(256, 256)
        best = None
        for th in [t for t in (8, 16, 32, 64, 128, 256) if t <= host_cores] or [1]:
            torch.set_num_threads(th)
            O.qformer_forward(w, cfg, ids[:16], tmask[:16], probe_patches, torch.ones(16, 256, dtype=torch.bool))
            t0 = time.time()
            O.qformer_forward(w, cfg, ids[:32], tmask[:32], probe_patches, torch.ones(32, 256, dtype=torch.bool))
            dt = time.time() - t0
            if best is None or dt < best[1]:
                best = (th, dt)
        cores = best[0]
        torch.set_num_threads(cores)
        t0 = time.time()
        patches = O.patch_embed(w, scene_cpu["mask_features"], 16)[0]
        fh, fw = scene_cpu["mask_features"].shape[-2:]
        grid = O.mask_grid(scene_cpu["pan_results"], scene_cpu["img_meta"]["img_shape"],
                           scene_cpu["img_meta"]["pad_shape"], (fh // 16, fw // 16))
        pm = O.pair_masks(O.object_masks(grid, [int(i) for i in scene_cpu["object_id_list"]]))
        t_prep = time.time() - t0
        reps = []
        for _ in range(3):
            t0 = time.time()
            out = O.qformer_forward(w, cfg, ids, tmask, patches, pm, chunk=64)
            logit, prob = O.existence_head(w, out)
            reps.append(time.time() - t0)
        t_rq = sorted(reps)[1]
        rq_rate = B / t_rq
        t_image = t_prep + t_rq
        sel = O.select_topk(prob, 20)
        sample = (f"relation-query: patch-embed + all {B} pairs through the fp32 oracle, 3 repetitions "
                  f"({', '.join(f'{r:.1f}s' for r in reps)}; median {rq_rate:.0f} pairs/s)")
        decodes, decodes16, w16n, deep = None, [], None, None
        if a.workload == "full":
            pids, pmask = H.llm_prompts(scene_cpu, sel)
            ts = {}
            decodes = []
            for i in range(n_dec):                                  # the oracle's own top-20 order, first n_dec pairs
                x, mask = O.llm_inputs(w, out[sel[i], 1:], pids[i], pmask[i])
                t0 = time.time()
                toks, lgs = O.llm_generate(w, cfg, x, mask, n_layers=lay[0], suppress_eos=True)
                if i == 0:
                    ts[lay[0]] = time.time() - t0
                decodes.append((toks, lgs[0]))
            # the same decodes over the LLM's matrices rounded through fp16: the values the reference's frozen fp16
            # checkpoint has after from_pretrained upcast it (parity of the `frozen_fp16_checkpoint` object)
            w16n = {k_: (v_.half().float() if k_.startswith("language_model.") and v_.dim() >= 2 else v_) for k_, v_ in w.items()}
            for i in range(n_dec):
                x, mask = O.llm_inputs(w16n, out[sel[i], 1:], pids[i], pmask[i])
                toks, lgs = O.llm_generate(w16n, cfg, x, mask, n_layers=lay[0], suppress_eos=True)
                decodes16.append((toks, lgs[0]))
            x, mask = O.llm_inputs(w, out[sel[0], 1:], pids[0], pmask[0])
            t0 = time.time()
            O.llm_generate(w, cfg, x, mask, n_layers=lay[1], suppress_eos=True)
            ts[lay[1]] = time.time() - t0
            per_layer = max((ts[lay[1]] - ts[lay[0]]) / (lay[1] - lay[0]), 1e-6)
            t_pair = ts[lay[0]] + (a.llm_layers - lay[0]) * per_layer
            sample += (f"; LMM decode: 1 of 20 selected pairs, 16 new tokens, {lay[0]} and {lay[1]} full-width fp32 "
                       f"layers timed ({ts[lay[0]]:.2f}s, {ts[lay[1]]:.2f}s) and extrapolated linearly to "
                       f"{a.llm_layers} layers ({t_pair:.1f}s per pair, serial batch-1 as V4:293-312)")
            untrunc = None
            if not a.no_untruncated and a.llm_layers == 32:
                try:
                    import psutil
                    avail = psutil.virtual_memory().available
                except Exception:  # noqa: BLE001
                    avail = 0
                if avail >= 64 << 30:
                    # ONE un-truncated decode: 32 fp32 layers (27 GB of seeded numpy PCG64 weights, one generator per
                    # tensor on a thread pool).  Its tokens and first-step logits are KEPT: the parity block decodes the
                    # same pair on the GPU over the same 32-layer weights (`decode_7b_32_layers`)
                    from openpsg_amd.weights import extend_llm_weights_numpy
                    cfg32 = PSGConfig(qformer=QFormerConfig(), llm=LlamaConfig(layers=32), max_object_num=N)
                    w32 = extend_llm_weights_numpy(w, cfg32, threads=min(32, host_cores))
                    t0 = time.time()
                    toks32, lgs32 = O.llm_generate(w32, cfg32, x, mask, n_layers=32, suppress_eos=True)
                    untrunc = time.time() - t0
                    deep = dict(w=w32, cfg=cfg32, decodes=[(toks32, lgs32[0])], shared=set(w))
                    sample += f"; ONE un-truncated 32-layer fp32 decode of a pair measured: {untrunc:.1f}s (used)"
                    t_pair = untrunc
                else:
                    sample += "; un-truncated run skipped (host has < 64 GB available)"
            t_image += 20 * t_pair
    obj = dict(value=round(N * (N - 1) / t_image, 3), unit="pairs/s", cores=cores, kind="port", sample=sample,
               host_cores=host_cores, host_cpu=host_model)
    return obj, dict(w=w, cfg=cfg, n=B, logit=logit, hidden=out, selected=sel, decodes=decodes, n_layers=lay[0],
                     w16=w16n, decodes16=decodes16 or None, deep=deep, pids=pids if a.workload == "full" else None,
                     pmask=pmask if a.workload == "full" else None)


def decode_parity(h, oracle_part, scene, names, eos, batch=0):
    """Decode leg against the oracle at Llama-2-7B width: the oracle's pair features of its first selected pairs are
    injected, the head (LLM truncated to the oracle's layer count) decodes them in one batch.
    batch > the number of oracle decodes: the oracle's further selected pairs ride along (their features injected, their
    tokens unchecked), so that the GPU runs the BENCHMARKED decode shape - 20 rows, the planned prompt-pass products."""
    dev = h.device
    dec = oracle_part["decodes"]
    sel = oracle_part["selected"][:max(len(dec), batch)]
    feats = torch.cat([oracle_part["hidden"][p, 1:] for p in sel]).to(dev, h.act_dtype).contiguous()
    rq = dict(num_objects=len(names))
    out = h.decode_selected(rq, names, selected=torch.tensor(sel, dtype=torch.int32, device=dev), pair_features=feats)
    fl = out["first_logits"].float().cpu()
    err, exact, matched, total = 0.0, 0, 0, 0
    for i, (toks, lg0) in enumerate(dec):
        o = lg0.clone()
        g_ = fl[i].clone()
        o[eos] = 0.0
        g_[eos] = 0.0                                            # suppress_eos writes -inf there
        err = max(err, float((o - g_).abs().max()))
        got = [int(t) for t in out["tokens_host"][i] if t >= 0]
        exact += got == toks
        matched += next((s_ for s_ in range(min(len(got), len(toks))) if got[s_] != toks[s_]), min(len(got), len(toks)))
        total += len(toks)
    return dict(first_step_logit_err=float(f"{err:.3e}"), exact_sequences=f"{exact}/{len(dec)}",
                tokens_before_first_divergence=f"{matched}/{total}", decode_rows=len(sel))


def deep_decode_parity(a, dev, scene, oracle_part, names, modes):
    """The decode at the depth that is benchmarked (V4:99-103 without `llm_truncate_num`, V4:305-312): the oracle's
    un-truncated 32-layer fp32 greedy decode of its first selected pair (cpu_baseline ran and timed it) against the
    fp32s head over the SAME 32-layer weights (uploaded from the oracle's own dict), decoding the oracle's 20 selected
    pairs in one batch - the benchmarked shape.  Then the same on fp16-VALUED matrices (the reference's frozen fp16
    checkpoint): the weights are rounded in their storage, the oracle decodes the pair again, the head reloads them and
    streams them as fp16.  Adds `decode_7b_32_layers` to modes['fp32s'] / modes['fp32s_frozen_fp16_checkpoint']."""
    from openpsg_amd.head import RelationTransformerHeadV4
    from openpsg_amd.weights import llm_matrices_as_fp16_values
    from oracle import psg_oracle as O
    deep = oracle_part["deep"]
    cfg32, N = deep["cfg"], a.objects

    def gpu_leg(w, decodes):
        h = RelationTransformerHeadV4(dtype="fp32s", device=str(dev), tokenizers="word", max_object_num=N,
                                      on_parse_error="skip", llm_config=cfg32.llm, suppress_eos=True)
        h.load_weights(w)
        part = dict(oracle_part, decodes=decodes)
        r = decode_parity(h, part, scene, names, cfg32.llm.eos, batch=min(20, len(oracle_part["selected"])))
        r["weights_streamed_as_fp16"] = bool(h.llm_engine._w16_all)
        r["oracle"] = "un-truncated 32-layer fp32 greedy decode of the pair on the host (oracle/psg_oracle.py)"
        del h
        torch.cuda.empty_cache()
        return r
    modes["fp32s"]["decode_7b_32_layers"] = gpu_leg(deep["w"], deep["decodes"])
    w32h = llm_matrices_as_fp16_values(deep["w"], in_place=set(deep["w"]) - deep["shared"])
    sel, hid = oracle_part["selected"], oracle_part["hidden"]
    dec16 = []
    with torch.no_grad():
        for i in range(len(deep["decodes"])):
            x, mask = O.llm_inputs(w32h, hid[sel[i], 1:], oracle_part["pids"][i], oracle_part["pmask"][i])
            toks, lgs = O.llm_generate(w32h, cfg32, x, mask, n_layers=32, suppress_eos=True)
            dec16.append((toks, lgs[0]))
    modes.setdefault("fp32s_frozen_fp16_checkpoint", {})["decode_7b_32_layers"] = gpu_leg(w32h, dec16)


def parity_block(a, dev, scene, oracle_part):
    """Per precision mode: deviation from the CPU oracle on the bench scene - the relation query on ALL pairs (existence
    logits, top-20) and the decode leg at Llama-2-7B width with 2 layers (the oracle's pair features injected).  The
    headline mode's numbers are copied to the top of the block."""
    from openpsg_amd.categories import INSTANCE_OFFSET, object_categories
    from openpsg_amd.head import RelationTransformerHeadV4
    N = a.objects
    out = {}
    ids_ = [int(i) for i in scene["object_id_list"]]
    names_ = [object_categories[i % INSTANCE_OFFSET] for i in ids_]
    modes = {}
    w, n = oracle_part["w"], oracle_part["n"]
    full = a.workload == "full" and oracle_part["decodes"] is not None
    ocfg = oracle_part["cfg"]
    k = min(20, n)
    for dt in ("fp32s", "fp32", "bf16", "fp16", "mixed"):
        kw = dict(llm_config=ocfg.llm, llm_truncate_num=oracle_part["n_layers"], suppress_eos=True) if full else {}
        h = RelationTransformerHeadV4(dtype=dt, device=str(dev), tokenizers="word", max_object_num=N,
                                      on_parse_error="skip", **kw)
        h.load_weights(w if full else {k_: v for k_, v in w.items() if not k_.startswith("language_model.")})
        # the whole image through the benchmarked path (cls-first last layer) against the oracle's logits of ALL pairs
        rq = h.run_relation_query(scene["mask_features"], scene["img_meta"], ids_, names_, scene["pan_results"])
        m = dict(max_logit_err_vs_oracle=float(f"{(rq['exist_logit'][:n].cpu() - oracle_part['logit']).abs().max().item():.3e}"),
                 top20_overlap=f"{len(set(rq['selected'].cpu().tolist()) & set(oracle_part['selected']))}/{k}")
        if full:
            m["decode_7b_width_2_layers"] = decode_parity(h, oracle_part, scene, names_, ocfg.llm.eos)
        modes[dt] = m
        del h, rq
        torch.cuda.empty_cache()
    if full and oracle_part.get("decodes16"):
        # the fp32s head over fp16-VALUED LLM weights (streamed as fp16: psg_split_gemm_w16 + the two-plane prompt pass)
        # against the oracle run on the same values
        h = RelationTransformerHeadV4(dtype="fp32s", device=str(dev), tokenizers="word", max_object_num=N,
                                      on_parse_error="skip", llm_config=ocfg.llm, llm_truncate_num=oracle_part["n_layers"],
                                      suppress_eos=True)
        h.load_weights(oracle_part["w16"])
        part16 = dict(oracle_part, decodes=oracle_part["decodes16"])
        modes["fp32s_frozen_fp16_checkpoint"] = dict(
            weights_streamed_as_fp16=bool(h.llm_engine._w16_all),
            decode_7b_width_2_layers=decode_parity(h, part16, scene, names_, ocfg.llm.eos))
        del h
        torch.cuda.empty_cache()
    if full and oracle_part.get("deep") is not None:
        try:
            deep_decode_parity(a, dev, scene, oracle_part, names_, modes)
        except Exception as exc:                                   # never lose the line
            modes["fp32s"]["decode_7b_32_layers"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    out["pairs_checked"] = n
    out["tolerance"] = "existence logits within 1e-3 of the fp32 oracle, greedy tokens identical (BASELINE.json north_star)"
    hm = modes.get(a.dtype, {})
    out["headline_mode"] = a.dtype
    out["headline_max_logit_err_vs_oracle"] = hm.get("max_logit_err_vs_oracle")
    def leg_ok(leg):                                               # every checked sequence token-exact, logits within 1e-3
        if leg is None:
            return True
        if "error" in leg:
            return False
        e_, n_ = leg["exact_sequences"].split("/")
        return e_ == n_ and leg["first_step_logit_err"] < 1e-3
    out["headline_within_tolerance"] = bool(
        hm.get("max_logit_err_vs_oracle", 1.0) < 1e-3 and hm.get("top20_overlap") == f"{k}/{k}" and (
            not full or (leg_ok(hm["decode_7b_width_2_layers"]) and leg_ok(hm.get("decode_7b_32_layers")))))
    d32 = hm.get("decode_7b_32_layers")
    out["headline_decode_32_layers"] = d32 if d32 is not None else (
        "not run (needs >= 64 GB of free host memory for the oracle's 27 GB of fp32 weights, and --llm-layers 32)")
    out["fp32_max_logit_err_vs_oracle"] = modes["fp32"]["max_logit_err_vs_oracle"]
    out["fp32s_max_logit_err_vs_oracle"] = modes["fp32s"]["max_logit_err_vs_oracle"]
    out["modes"] = modes
    return out


PRECISION = {
    "fp32s": "fp32 (the reference's arithmetic, V4:99-100): fp32 weights, activations, KV cache, norms, softmaxes and "
             "decode-step projections; the prompt pass's and the Q-Former's projections are split-fp16 products "
             "(x.w = xh.wh + xh.wl + xl.wh on the 16-bit matrix cores, fp32 accumulation; ~3e-7 relative per product, "
             "the error class of an fp32 GEMM) - head dtype 'fp32s'",
    "fp32": "fp32 everywhere, exact (fp32-input matrix instructions / library SGEMM) - head dtype 'fp32'",
    "mixed": "mixed: fp16 GEMM operands / KV cache, fp32 accumulation, fp32 Llama residual stream (outside the north "
             "star's 1e-3 tolerance: existence logits ~0.025 off the fp32 oracle)",
    "fp16": "fp16 operands and residual stream, fp32 accumulation",
    "bf16": "bf16 operands and residual stream, fp32 accumulation"}
DTYPE_LABEL = {"fp32s": "fp32", "fp32": "fp32", "mixed": "fp16", "fp16": "fp16", "bf16": "bf16"}
# matrix peak the mode's dense projections run against, and how many matrix products one algorithmic product costs
MATRIX_PEAK = {"fp32s": (2.5e15, 3), "fp32": (157.3e12, 1), "mixed": (2.5e15, 1), "fp16": (2.5e15, 1), "bf16": (2.5e15, 1)}


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def pmc_traffic_checked(pmc_file):
    """(HBM bytes per launch, source note) from the committed PMC summary of a kernel - counters need their own rocprofv3
    passes, so the figure is READ, not re-measured in the run (profiles/README.md).  The summary records the hashes of the
    kernel's source files at collection time (`kernel_sources`, tools/pmc_summary.py): if any of them has changed since,
    the figure no longer describes the kernel that was timed and `traffic` is reported as null with the reason."""
    pmc = os.path.join(REPO, "profiles", pmc_file)
    if not os.path.exists(pmc):
        return None, f"profiles/{pmc_file} is missing"
    d = json.load(open(pmc))
    src = d.get("kernel_sources") or {}
    stale = [f for f, h in src.items() if not os.path.exists(os.path.join(REPO, f)) or _sha16(os.path.join(REPO, f)) != h]
    if stale:
        return None, (f"profiles/{pmc_file} is STALE: {', '.join(stale)} changed since its rocprofv3 --pmc passes "
                      "(re-run tools/collect_profiles_r06.sh)")
    note = (f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel; not re-measured in this "
            "run" + ("; the kernel's sources are unchanged since: hashes checked" if src else "") + ")")
    return d.get("hbm_bytes_per_launch"), note


def pmc_traffic(pmc_file):
    return pmc_traffic_checked(pmc_file)[0]


def cross_attention_figures(dev, N, size):
    """North star: ">= 40 % MFMA utilisation on relation-query cross-attention".  The kernel (psg_qformer_cross_attn, bf16:
    cross_attn_dma_kernel) on the bench scene's own object masks, N * N pairs x 33 query rows x 12 heads, timed LIVE with
    events around REP back-to-back launches on torch's current stream (the stream the op launches on).  `frac` is the
    DENSE-EQUIVALENT figure (4 P 33 L 768 flops: key tiles nobody attends to are skipped, exactly, so the matrix pipe
    executes a fraction of it); what the matrix pipe itself was busy, and the kernel's duration inside the relation-query
    pass, come from the committed rocprofv3 summaries of the same tree and are labelled as such."""
    from openpsg_amd import ops
    from openpsg_amd.synthetic import make_scene
    L = (size // 64) ** 2
    P = N * N
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.randn(P * 33, 768, device=dev, generator=g).bfloat16()
    k = torch.randn(L, 768, device=dev, generator=g).bfloat16()
    v = torch.randn(L, 768, device=dev, generator=g).bfloat16()
    sc = make_scene((size, size), N, seed=0, device=str(dev), features=False)
    grid = ops.mask_grid(sc["pan_results"], (size, size), (size, size), (size // 64, size // 64))
    bits = ops.object_bitmasks(grid, torch.tensor([int(i) for i in sc["object_id_list"]], dtype=torch.int32, device=dev))
    pidx = torch.arange(P, device=dev, dtype=torch.int32)
    out = torch.empty_like(q)
    REP = 20

    def run():
        for _ in range(REP):
            ops.qformer_cross_attn(q, k, v, bits, pidx, N, 33, 12, out=out)
    for _ in range(3):
        run()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    us = sorted(ts)[len(ts) // 2]
    flops = 4.0 * P * 33 * L * 768
    nbytes = 2.0 * P * 33 * 768 * 2                                    # Q in + context out
    fig = {"kernel": "cross_attn_dma_kernel<EBf16> (psg_qformer_cross_attn: HF-IB:464-496 driven by V4:168-185)",
           "pairs": P, "query_rows_per_pair": 33, "keys": L, "heads": 12, "dtype": "bf16",
           "us_per_launch": round(us, 2), "timing": f"live: HIP events around {REP} back-to-back launches, median of 7",
           "bound": "mfma", "achieved": round(flops / us / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s",
           "frac": round(flops / us / 1e6 / 2500.0, 4), "frac_kind": "dense-equivalent (masked key tiles are skipped, exactly)",
           "target_frac": 0.40, "flops_per_launch": int(flops),
           "hbm": {"bytes_per_launch": int(nbytes), "achieved": round(nbytes / us / 1e3, 1), "unit": "GB/s",
                   "frac": round(nbytes / us / 1e3 / 8000.0, 4)}}
    # committed counter / trace summaries of the same tree (tools/collect_profiles_r06.sh), NOT re-measured in this run
    try:
        txt = open(os.path.join(REPO, "profiles", "r06_xattn_pmc_n50.txt")).read()
        import re
        m = re.search(r"= ([0-9.]+) % busy", txt)
        t = re.search(r"HBM traffic: read .*? = ([0-9.]+) MB, written ([0-9.]+) MB", txt)
        if m:
            fig["matrix_pipe_busy"] = {"value": round(float(m.group(1)) / 100.0, 3),
                                       "source": "profiles/r06_xattn_pmc_n50.txt: SQ_VALU_MFMA_BUSY_CYCLES / (32 x GRBM_GUI_ACTIVE)"}
        if t:
            fig["hbm"]["traffic"] = int((float(t.group(1)) + float(t.group(2))) * 1e6)
            fig["hbm"]["traffic_source"] = "profiles/r06_xattn_pmc_n50.txt (2 x FETCH_SIZE + WRITE_SIZE passes)"
        for row in open(os.path.join(REPO, "profiles", "r06_per_step_rq_kernels.csv")):
            if row.startswith("void cross_attn_dma_kernel<EBf16; 2; 10>"):
                us_situ = float(row.rstrip().split(",")[-2])
                fig["in_situ"] = {"us_per_launch": us_situ, "frac": round(flops / us_situ / 1e6 / 2500.0, 4),
                                  "source": "profiles/r06_per_step_rq_kernels.csv (kernel trace of the C2 pass: the launch "
                                            "between its projection GEMMs, de-duplicated query blocks)"}
    except Exception:                                                  # the live figures stand without the files
        pass
    return fig


def decode_roofline(head, N, pmc_file, kernel):
    bpl, spl, n = measure_decode_gemm(head, min(20, N * N))
    ach = bpl / spl / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": int(bpl),
         "us_per_launch": round(spl * 1e6, 2), "launches_per_decode_step": n, "bytes_per_decode_step": int(bpl * n)}
    r["traffic"], r["traffic_source"] = pmc_traffic_checked(pmc_file)
    return r


def image_floor(head, a, dtype, bytes_per_step, ms_measured, prompt_mult=None):
    """The whole image against its combined floor: (max_new - 1) weight passes at the HBM peak + the prompt pass's and the
    relation query's algorithmic FLOPs at the dense matrix peak of the dtype they run in (fp32s: three fp16 products per
    algorithmic product; prompt_mult = 2 for the prompt pass over fp16-valued weights, which have no low part)."""
    m = head.cfg.llm
    peak, mult = MATRIX_PEAK[dtype]
    pmult = mult if prompt_mult is None else prompt_mult
    steps = head.cfg.max_new_tokens - 1
    X = head.last.get("llm_inputs") if head.last is not None else None
    rows = int(X.shape[0] * X.shape[1]) if X is not None else head.cfg.num_selected * 49
    per_layer = 4 * m.hidden * m.hidden + 3 * m.hidden * m.inter
    fl_prompt = 2.0 * rows * per_layer * a.llm_layers
    T = max(v[2].shape[1] for k_, v in head._table_cache.items() if k_[0] == "q")
    fl_rq = relation_query_flops(a.objects, (a.size // 64) ** 2, T, head.cls_first, head.cfg.num_selected)
    t_dec = steps * bytes_per_step / (HBM_PEAK_GBS * 1e9)
    # the decode attention reads every cached K and V row of every pair once per step (~5 % of a step that the
    # weight-only floor left out): pairs x layers x context rows x hidden x 2 (K, V) x bytes of the cache's type; the
    # context of step s is the pair's prompt (32 visual rows + its tokens) + s
    plen = head.last.get("prompt_len") if head.last is not None else None
    kv_item = 4 if dtype in ("fp32", "fp32s") else 2
    if plen is not None:
        ctx0 = plen.to(torch.float64).cpu() + head.cfg.qformer.num_query
        ctx_rows = float(sum((ctx0 + s_).sum() for s_ in range(1, steps + 1)))
    else:
        ctx_rows = float(head.cfg.num_selected * sum(48 + s_ for s_ in range(1, steps + 1)))
    kv_bytes = ctx_rows * a.llm_layers * m.hidden * 2 * kv_item
    t_kv = kv_bytes / (HBM_PEAK_GBS * 1e9)
    t_pp = fl_prompt * pmult / peak
    t_rq = fl_rq * mult / peak
    floor = (t_dec + t_kv + t_pp + t_rq) * 1e3
    return {"floor_ms": round(floor, 2), "measured_ms": round(ms_measured, 2), "frac": round(floor / ms_measured, 4),
            "terms_ms": {"decode_weight_passes": round(t_dec * 1e3, 2), "decode_kv_reads": round(t_kv * 1e3, 2),
                         "prompt_pass": round(t_pp * 1e3, 2), "relation_query": round(t_rq * 1e3, 2)},
            "assumptions": f"{steps} decode steps x {bytes_per_step / 1e9:.2f} GB of weights + {kv_bytes / 1e9:.1f} GB of cached "
                           f"K / V rows over the steps, at {HBM_PEAK_GBS / 1e3:.0f} TB/s; prompt pass "
                           f"{fl_prompt / 1e12:.1f} TFLOP over {rows} rows x{pmult} matrix products and relation query "
                           f"{fl_rq / 1e12:.2f} TFLOP x{mult}, at {peak / 1e12:.0f} TFLOP/s"}


def rq_stage(head, a, scene, pairs_per_image, steps=10):
    from openpsg_amd.categories import INSTANCE_OFFSET, object_categories
    N = a.objects
    ids_ = [int(i) for i in scene["object_id_list"]]
    names_ = [object_categories[i % INSTANCE_OFFSET] for i in ids_]

    def rq_step():
        rq = head.run_relation_query(scene["mask_features"], scene["img_meta"], ids_, names_, scene["pan_results"])
        head.selected_pair_features(rq)
        return rq["selected"].cpu()
    el = time_steps(rq_step, 2, steps) / steps
    T = max(v[2].shape[1] for k_, v in head._table_cache.items() if k_[0] == "q")
    fl = relation_query_flops(N, (a.size // 64) ** 2, T, head.cls_first, head.cfg.num_selected)
    U = max([v[3][0].shape[0] for v in head._gather_cache.values() if len(v) > 3] or [N * N])
    fx = relation_query_flops_executed(N, (a.size // 64) ** 2, T, U, head.cfg.num_selected) if head.cls_first else fl
    return el, fl, fx, U, T


def scene_inputs(scene):
    return dict(mask_features=scene["mask_features"], img_metas=[scene["img_meta"]],
                object_info=[dict(object_id_list=scene["object_id_list"], pan_results=scene["pan_results"])])


def time_in_flight(head, inputs, warmup, steps, slots=2):
    """`steps` images through head.submit with `slots` in flight; every result taken inside the timed region."""
    import collections
    pend, n = collections.deque(), [0]

    def run(k):
        for _ in range(k):
            pend.append(head.submit(inputs, slot=n[0] % slots))
            n[0] += 1
            if len(pend) >= slots:
                pend.popleft().result()
        while pend:
            pend.popleft().result()
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def batched_decode_figures(head, a, dev, pairs_per_image, steps):
    """head.forward_batch on 4 and on 8 images per step (BASELINE C5's batch): the images' 80 / 160 selected pairs decoded
    together, so that the Llama weights stream once per decode step for all of them."""
    from openpsg_amd.synthetic import make_scene
    N = a.objects
    try:
        B3 = 4
        batch = [scene_inputs(make_scene((a.size, a.size), N, seed=m, device=str(dev))) for m in range(B3)]
        k = max(2, min(steps, 5))
        elb = time_steps(lambda: head.forward_batch(batch), 1, k)
        out = {"images_per_step": B3, "value": round(B3 * k * pairs_per_image / elb, 1), "unit": "pairs/s",
               "ms_per_step": round(elb / k * 1e3, 3), "steps": k}
        B8 = 8                                                         # BASELINE C5's batch: 160 decode rows per step
        batch += [scene_inputs(make_scene((a.size, a.size), N, seed=m, device=str(dev))) for m in range(B3, B8)]
        el8 = time_steps(lambda: head.forward_batch(batch), 1, 2)
        out["eight_images"] = {"value": round(B8 * 2 * pairs_per_image / el8, 1), "ms_per_step": round(el8 / 2 * 1e3, 3),
                               "steps": 2}
        return out
    except Exception as exc:                                           # never lose the line
        return {"error": f"{type(exc).__name__}: {exc}"[:200]}


def median_step(step, warmup, steps):
    """Median seconds of `steps` individually timed calls (a one-off stall - an allocator refill, a library kernel
    selection - does not move it)."""
    for _ in range(warmup):
        step()
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def time_steps(step, warmup, steps):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                                                 # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PSG_BENCH_FORCE_DIST=1: run the pair-sharded pipeline (RCCL collectives included) even at world size 1
    force_dist = os.environ.get("PSG_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        # RCCL initialises lazily (no device_id): an eager communicator creates its HSA queues before the head's streams
        # and the two decode streams of a rank then interfere (140 instead of 117 ms per two images, measured at world 1)
        dist.init_process_group("nccl")
        world = dist.get_world_size()                                  # the ranks RCCL actually sees
        if world != a.gpus and rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the job has {world} rank(s); reporting n_gpus = {world}",
                  file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from openpsg_amd.synthetic import make_scene
    head = setup_head(a, dev)
    N = a.objects
    pairs_per_image = N * (N - 1)
    single = world == 1 and not force_dist

    drain = lambda: None  # noqa: E731  (a pipelined step replaces it)
    per_rank = 1
    if single:
        scene = make_scene((a.size, a.size), N, seed=0, device=str(dev), num_categories=a.categories)
        inputs = scene_inputs(scene)
        if a.workload == "full" and a.images_per_step > 1:
            batch = [scene_inputs(make_scene((a.size, a.size), N, seed=m, device=str(dev)))
                     for m in range(a.images_per_step)]

            def step():
                return head.forward_batch(batch)
        elif a.workload == "full" and a.in_flight > 1:
            # --in-flight k (not the default): image j+1 is enqueued on another HIP stream before image j's result is awaited
            import collections
            pending, issued = collections.deque(), [0]
            head.serialize_decodes = a.serialize_decodes

            def step():
                pending.append(head.submit(inputs, slot=issued[0] % a.in_flight))
                issued[0] += 1
                return pending.popleft().result() if len(pending) >= a.in_flight else None

            def drain():
                while pending:
                    pending.popleft().result()
        elif a.workload == "full":
            def step():                                                # ONE head call, its result awaited
                return head(inputs)
        else:
            from openpsg_amd.categories import INSTANCE_OFFSET, object_categories
            obj_ids = [int(i) for i in scene["object_id_list"]]
            names = [object_categories[i % INSTANCE_OFFSET] for i in obj_ids]

            def step():
                rq = head.run_relation_query(scene["mask_features"], scene["img_meta"], obj_ids, names,
                                             scene["pan_results"])
                head.selected_pair_features(rq)                 # pair_feature of the selected pairs is part of the stage
                return rq["selected"].cpu()
        barrier = lambda: None  # noqa: E731
    else:
        import torch.distributed as dist
        from openpsg_amd.dist import PairShardedPipeline
        # a step is `per_rank` images per rank (--in-flight; default 1: a rank decodes one image of the step); every
        # image's pairs are sharded over all ranks
        per_rank = 1 if a.workload != "full" else max(1, a.in_flight)
        scenes = [make_scene((a.size, a.size), N, seed=m, device=str(dev)) for m in range(world * per_rank)]
        pipe = PairShardedPipeline(head, dist.group.WORLD, decode=a.workload == "full")
        barrier = lambda: dist.barrier(device_ids=[local])  # noqa: E731

        def step():
            return pipe.step(scenes)

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                                            # every submitted image's result is taken
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if not single:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    plain = single and a.workload == "full" and a.images_per_step == 1

    strong = None
    if not single and not a.no_strong:
        # STRONG scaling: one BASELINE-C4 image (100 masks) for all ranks; max over ranks like the headline
        import torch.distributed as dist
        from openpsg_amd.dist import PairShardedPipeline
        n4 = 100
        head.max_object_num = max(head.max_object_num, n4)
        scene4 = make_scene((a.size, a.size), n4, seed=4, device=str(dev))
        pipe1 = PairShardedPipeline(head, dist.group.WORLD, decode=a.workload == "full")
        ks = max(3, min(a.steps, 10))
        for _ in range(2):
            pipe1.step_one_image(scene4)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(ks):
            pipe1.step_one_image(scene4)
        torch.cuda.synchronize()
        barrier()
        t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el1 = float(t.item()) / ks
        # the single-GPU reference of the same image, measured in this run (every rank runs it at the same time;
        # rank 0's figure is reported)
        inputs4 = scene_inputs(scene4)
        ref1 = time_steps(lambda: head(inputs4) if a.workload == "full" else head.run_relation_query(
            scene4["mask_features"], scene4["img_meta"], [int(i) for i in scene4["object_id_list"]],
            pipe1.be._names(scene4), scene4["pan_results"]), 2, 5) / 5
        barrier()
        wbytes = 26.4 if a.dtype in ("fp32", "fp32s") else 13.2
        strong = {"workload": f"ONE {a.size}x{a.size} image, {n4} masks ({n4 * (n4 - 1)} pairs), pairs sharded over "
                              f"{world} rank(s), top-20 decodes dealt round-robin",
                  "ms_per_image": round(el1 * 1e3, 3), "value": round(n4 * (n4 - 1) / el1, 1), "unit": "pairs/s",
                  "steps": ks, "single_gpu_reference_ms": round(ref1 * 1e3, 3),
                  "speedup_vs_single_gpu": round(ref1 / el1, 3), "dtype": DTYPE_LABEL[a.dtype], "mode": a.dtype,
                  "bound": f"16 passes over the {wbytes} GB of Llama weights per image on every decoding rank (a decode "
                           "step streams all weights for 1 row as for 20; no tensor parallelism, SURVEY 8e): only the "
                           "relation query and the compute-bound prompt pass shrink with N"}

    strong_rq = None
    if not single and not a.no_strong:
        # the part of C4 that actually shards: relation query + existence head + all-gather + top-20 + the
        # selected pairs' features, no decode - the curve north_star asks for at 1/2/4/8 GPUs
        import torch.distributed as dist
        from openpsg_amd.dist import PairShardedPipeline
        pipe_q = PairShardedPipeline(head, dist.group.WORLD, decode=False)
        for _ in range(2):
            pipe_q.step_one_image(scene4)
        torch.cuda.synchronize()
        barrier()
        kq = max(5, min(a.steps, 20))
        t1 = time.perf_counter()
        for _ in range(kq):
            pipe_q.step_one_image(scene4)
        torch.cuda.synchronize()
        barrier()
        t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elq = float(t.item()) / kq
        ids4 = [int(i) for i in scene4["object_id_list"]]
        names4 = pipe_q.be._names(scene4)

        def rq4():                                                     # the same stage on one GPU: up to the selection
            rq = head.run_relation_query(scene4["mask_features"], scene4["img_meta"], ids4, names4, scene4["pan_results"])
            return rq["selected"].cpu()
        refq = time_steps(rq4, 2, 5) / 5
        barrier()
        strong_rq = {"workload": f"ONE {a.size}x{a.size} image, {n4} masks ({n4 * (n4 - 1)} pairs), relation query only "
                                 f"(C4's sharded part), pairs sharded over {world} rank(s): broadcast of the image constants, "
                                 "pair shards, all-gather of the probabilities, the same top-20 on every rank",
                     "ms_per_image": round(elq * 1e3, 3), "value": round(n4 * (n4 - 1) / elq, 1), "unit": "pairs/s", "steps": kq,
                     "single_gpu_reference_ms": round(refq * 1e3, 3), "speedup_vs_single_gpu": round(refq / elq, 3),
                     "dtype": DTYPE_LABEL[a.dtype], "mode": a.dtype}

    if rank == 0:
        ips = a.images_per_step if (single and a.workload == "full") else world * per_rank
        images = ips * a.steps
        wl = ("C3: 1024x1024, 50 masks, full path incl. LMM autoregressive relation decode (Llama-2-7B shape, top-20 "
              "pairs, 16 new tokens each, EOS suppressed)") if a.workload == "full" else \
             "C2: 1024x1024, 50 masks, relation-query transformer only"
        if (a.size, a.objects, a.categories) != (1024, 50, 133):
            wl = f"custom: {a.size}x{a.size}, {a.objects} masks from {a.categories} classes, {a.workload}"
        if not single:
            par = (f"{ips} images per step, pairs of every image sharded over {world} rank(s) (all-gather of patches / "
                   f"probabilities / token ids, reduce-scatter of the selected pair features); rank r decodes images r, "
                   f"r + {world}, ... of the step" + (" side by side on two HIP streams" if per_rank > 1 else ""))
        elif plain and a.in_flight > 1:
            par = (f"single GPU, {a.in_flight} images in flight (head.submit: image j+1 enqueued on another HIP stream before "
                   "image j's result is awaited; every image processed in full inside the timed region)")
        else:
            par = "single GPU, one head call per step, its result awaited before the next image is submitted"
        ms_step = elapsed / a.steps * 1e3
        line = {
            "metric": METRIC, "value": round(images * pairs_per_image / elapsed, 1), "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[a.dtype], "data": "synthetic",
            "config": {"workload": wl, "objects": N, "pairs_per_image": pairs_per_image, "images_per_step": ips,
                       "patches": (a.size // 64) ** 2, "llm_layers": a.llm_layers if a.workload == "full" else 0,
                       "mode": a.dtype, "precision": PRECISION[a.dtype], "parallelism": par},
        }
        f32 = a.dtype in ("fp32", "fp32s")
        k_name = ("skinny_gemm_f32_kernel (psg_skinny_gemm with PSG_F32: v_mfma_f32_16x16x1_4b + v_mfma_f32_4x4x1_16b on "
                  "LDS-DMA rings; fp32 weights)") if f32 else \
                 "skinny_gemm_dma_kernel (psg_skinny_gemm; 8-wave slabs, 11-wave for gate/up, 12-wave for q/k/v)"
        pmc_file = "pmc_skinny_gemm_f32.json" if f32 else "pmc_skinny_gemm.json"
        if not a.no_roofline and a.workload == "full":
            line["roofline"] = decode_roofline(head, N, pmc_file, k_name)
            if plain and a.in_flight <= 1:
                line["roofline"]["image"] = image_floor(head, a, a.dtype, line["roofline"]["bytes_per_decode_step"], ms_step)
        if a.workload == "rq" and not a.no_roofline and single:
            T = max(v[2].shape[1] for k_, v in head._table_cache.items() if k_[0] == "q")
            fl = relation_query_flops(N, (a.size // 64) ** 2, T, head.cls_first, head.cfg.num_selected)
            peak, mult = MATRIX_PEAK[a.dtype]
            ach = fl / (elapsed / a.steps) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "relation-query stage (GEMMs + cross-attention + self-attention + "
                                "qformer_cls_attn_input_kernel)", "achieved": round(ach, 1), "peak": peak / mult / 1e12,
                                "unit": "TFLOP/s", "frac": round(ach * 1e12 * mult / peak, 4), "traffic": None,
                                "flops_per_step": int(fl)}
        if plain:
            # the same mode with two images in flight (fixed form, no best-of selection): head.submit on two slot streams
            try:
                k2 = max(6, a.steps // 2)
                el_p = time_in_flight(head, inputs, 4, k2) / k2
                line["two_in_flight"] = {"ms_per_image": round(el_p * 1e3, 3), "value": round(pairs_per_image / el_p, 1),
                                         "unit": "pairs/s", "steps": k2, "mode": a.dtype,
                                         "note": "head.submit, image j+1 enqueued on a second HIP stream before image j's "
                                                 "result is awaited; whether the slots overlap depends on the HIP runtime's "
                                                 "stream -> hardware-queue binding (DESIGN 4.5)"}
            except Exception as exc:                                   # never lose the headline line
                line["two_in_flight"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            # stage split (not part of the contract): relation-query stage alone in the headline mode
            el, fl, fx, U, T = rq_stage(head, a, scene, pairs_per_image)
            peak, mult = MATRIX_PEAK[a.dtype]
            line["stages"] = {"relation_query_ms": round(el * 1e3, 3),
                              "relation_query_pairs_per_s": round(pairs_per_image / el, 1),
                              "relation_query_tflops": round(fl / el / 1e12, 1),
                              "relation_query_matrix_frac": round(fl * mult / el / peak, 4),
                              "relation_query_executed_tflops": round(fx / el / 1e12, 1),
                              "distinct_prompts": int(U), "prompt_tokens": T}
            # the timed region repeats ONE scene (hot per-names caches); what an image with a class list never seen
            # before costs: six other scenes, each run once (head.warm_prompts() as a deployment does at start-up)
            try:
                head.warm_prompts()
                fresh = [make_scene((a.size, a.size), N, seed=100 + m, device=str(dev), num_categories=a.categories)
                         for m in range(6)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for sc in fresh:
                    head(scene_inputs(sc))
                torch.cuda.synchronize()
                line["stages"]["new_scene_ms_per_image"] = round((time.perf_counter() - t0) / len(fresh) * 1e3, 3)
                del fresh
            except Exception as e:  # noqa: BLE001
                line["stages"]["new_scene_ms_per_image"] = f"failed: {e}"
            # BASELINE C4 (100 masks = 9900 pairs) on ONE GPU, headline mode: the N = 1 point of the curve SURVEY 8d asks for
            # (the sharded N > 1 points are `strong_scaling` / `strong_scaling_rq` of a multi-GPU launch)
            if (a.size, a.objects) == (1024, 50) and not a.no_c4:
                try:
                    n4 = 100
                    head.max_object_num = max(head.max_object_num, n4)
                    scene4 = make_scene((a.size, a.size), n4, seed=4, device=str(dev))
                    inputs4 = scene_inputs(scene4)
                    k4 = max(5, a.steps // 4)
                    el4 = time_steps(lambda: head(inputs4), 2, k4) / k4
                    ids4 = [int(i) for i in scene4["object_id_list"]]
                    from openpsg_amd.categories import INSTANCE_OFFSET, object_categories
                    names4 = [object_categories[i % INSTANCE_OFFSET] for i in ids4]

                    def rq4():
                        rq = head.run_relation_query(scene4["mask_features"], scene4["img_meta"], ids4, names4,
                                                     scene4["pan_results"])
                        head.selected_pair_features(rq)
                        return rq["selected"].cpu()
                    el4q = time_steps(rq4, 2, k4) / k4
                    p4 = n4 * (n4 - 1)
                    line["c4_single_gpu"] = {
                        "workload": f"C4 on one GPU: {a.size}x{a.size}, {n4} masks ({p4} pairs), full path incl. the top-20 decode",
                        "mode": a.dtype, "dtype": DTYPE_LABEL[a.dtype], "ms_per_step": round(el4 * 1e3, 3),
                        "value": round(p4 / el4, 1), "unit": "pairs/s", "steps": k4,
                        "relation_query_ms": round(el4q * 1e3, 3), "relation_query_pairs_per_s": round(p4 / el4q, 1)}
                    del scene4, inputs4
                except Exception as exc:                               # never lose the headline line
                    line["c4_single_gpu"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            # throughput mode at the headline's precision (round 6): several images per step, decode rows 80 / 160 on the
            # library SGEMM (generic fp32 weights: 4 bytes per weight once per step for all rows)
            if a.dtype in ("fp32s", "fp32") and not a.no_batched:
                bd = batched_decode_figures(head, a, dev, pairs_per_image, a.steps)
                bd["mode"] = a.dtype
                bd["decode_projections"] = "library SGEMM over the fp32 weights (exact), 33..160 rows"
                line["batched_decode"] = bd
        # ---- sub-objects measured with their own heads: the headline head is released first -------------------------
        if plain:
            import copy
            del head
            torch.cuda.empty_cache()
            if a.dtype != "fp32" and not a.no_exact:
                try:                                                   # exact fp32 everywhere
                    b = copy.copy(a)
                    b.dtype_override = "fp32"
                    h = setup_head(b, dev)
                    ke = max(5, a.steps // 4)
                    el = time_steps(lambda: h(inputs), 2, ke) / ke
                    line["exact_fp32"] = {"mode": "fp32", "precision": PRECISION["fp32"], "ms_per_step": round(el * 1e3, 3),
                                          "value": round(pairs_per_image / el, 1), "unit": "pairs/s", "steps": ke}
                    if not a.no_roofline:
                        r = decode_roofline(h, N, "pmc_skinny_gemm_f32.json", "skinny_gemm_f32_kernel")
                        r["image"] = image_floor(h, a, "fp32", r["bytes_per_decode_step"], el * 1e3)
                        line["exact_fp32"]["roofline"] = r
                    del h
                    torch.cuda.empty_cache()
                except Exception as exc:
                    line["exact_fp32"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            if a.dtype == "fp32s" and a.llm_values == "fp32" and not a.no_frozen16:
                # The reference's LLM is the FROZEN Llama-2-7b-hf checkpoint (configs/psg/baseline_v4_ov.py:61-65), fp16 on
                # disk and upcast by from_pretrained (V4:99-100): its fp32 matrices hold fp16 values.  Same mode, same
                # arithmetic class, weights drawn as such values: the engine verifies the round trip per tensor and streams
                # fp16 copies (decode: 2 bytes per weight; prompt pass: no low part of the weight = 2 products, not 3)
                try:
                    b = copy.copy(a)
                    b.llm_values = "fp16"
                    h = setup_head(b, dev)
                    kf = a.steps                                           # timed like the headline: same steps, same warm-up
                    el = time_steps(lambda: h(inputs), a.warmup, kf) / kf
                    kp = max(6, a.steps // 2)
                    el_p = time_in_flight(h, inputs, 4, kp) / kp
                    fz = {"mode": "fp32s", "dtype": "fp32", "standing": "equal to the headline: the same mode, arithmetic class, "
                          "parity gate, steps and warm-up; `value` stays on generic fp32 weight values because the drop-in head "
                          "takes any checkpoint (VERDICT r5) - this object is the figure for the reference's shipped configuration",
                          "weights": "random-init fp32 LLM matrices holding fp16 VALUES (what from_pretrained makes of the "
                                     "frozen fp16 Llama-2-7b-hf checkpoint the reference loads: V4:99-100, "
                                     "configs/psg/baseline_v4_ov.py:61-65); every tensor verified to round-trip at load",
                          "weights_streamed_as_fp16": bool(h.llm_engine._w16_all),
                          "precision": "as the headline (fp32-grade): decode projections = two fp16 products of the split "
                                       "fp32 rows against the fp16-valued weight (2^-22 relative), prompt pass = ONE library "
                                       "GEMM of the two-plane operand per projection",
                          "ms_per_step": round(el * 1e3, 3), "value": round(pairs_per_image / el, 1), "unit": "pairs/s",
                          "steps": kf, "warmup": a.warmup,
                          "one_image_at_a_time": {"ms_per_image": round(el * 1e3, 3), "value": round(pairs_per_image / el, 1),
                                                  "unit": "pairs/s", "steps": kf},
                          "two_in_flight": {"ms_per_image": round(el_p * 1e3, 3), "value": round(pairs_per_image / el_p, 1),
                                            "unit": "pairs/s", "steps": kp}}
                    if not a.no_roofline and h.llm_engine._w16_all:
                        bpl, spl, nl = measure_decode_gemm_w16(h, min(20, N * N))
                        ach = bpl / spl / 1e9
                        fz["roofline"] = {"bound": "hbm", "kernel": "batch_gemm_kernel<EF16, 2, 1, 1, PAIR> (psg_split_gemm_w16)",
                                          "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_checked("pmc_split_gemm_w16.json")[0],
                                          "bytes_per_launch": int(bpl),
                                          "us_per_launch": round(spl * 1e6, 2), "launches_per_decode_step": nl,
                                          "bytes_per_decode_step": int(bpl * nl),
                                          "traffic_source": pmc_traffic_checked("pmc_split_gemm_w16.json")[1],
                                          # the same three terms as the headline's roofline.image, from this head's own shapes
                                          "image": image_floor(h, a, "fp32s", int(bpl * nl), el * 1e3, prompt_mult=2)}
                    if not a.no_batched:
                        bd = batched_decode_figures(h, a, dev, pairs_per_image, a.steps)
                        bd["decode_projections"] = ("one library product of the two fp16 planes of the fp32 rows against the fp16 "
                                                    "weight per projection (2 bytes per weight, fp32 result), 33..160 rows")
                        fz["batched_decode"] = bd
                    line["frozen_fp16_checkpoint"] = fz
                    del h
                    torch.cuda.empty_cache()
                except Exception as exc:
                    line["frozen_fp16_checkpoint"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            if a.dtype != "mixed" and not a.no_mixed:
                try:                                                   # rounds 3-4's headline mode, outside the tolerance
                    b = copy.copy(a)
                    b.dtype_override = "mixed"
                    h = setup_head(b, dev)
                    km = max(10, a.steps // 2)
                    el = time_steps(lambda: h(inputs), 3, km) / km
                    el_p = time_in_flight(h, inputs, 4, km) / km
                    mx = {"mode": "mixed", "dtype": "fp16", "precision": PRECISION["mixed"],
                          "one_image_at_a_time": {"ms_per_image": round(el * 1e3, 3), "value": round(pairs_per_image / el, 1),
                                                  "unit": "pairs/s", "steps": km},
                          "two_in_flight": {"ms_per_image": round(el_p * 1e3, 3), "value": round(pairs_per_image / el_p, 1),
                                            "unit": "pairs/s", "steps": km}}
                    if not a.no_roofline:
                        r = decode_roofline(h, N, "pmc_skinny_gemm.json", "skinny_gemm_dma_kernel")
                        r["image"] = image_floor(h, a, "mixed", r["bytes_per_decode_step"], el * 1e3)
                        mx["roofline"] = r
                    if not a.no_batched:
                        mx["batched_decode"] = batched_decode_figures(h, a, dev, pairs_per_image, a.steps)
                    line["mixed"] = mx
                    del h
                    torch.cuda.empty_cache()
                except Exception as exc:
                    line["mixed"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            if not a.no_roofline:
                # BASELINE config C2 (relation-query transformer only, bf16) as an object of the SAME line, so that the
                # driver's default run records it: its own bf16 head (no LLM), the stage timed as `--workload rq` times it
                try:
                    b = copy.copy(a)
                    b.workload, b.dtype_override = "rq", "bf16"
                    h2 = setup_head(b, dev)
                    k2 = max(10, a.steps)
                    el2, fl2, fx2, U2, T2 = rq_stage(h2, b, scene, pairs_per_image, steps=k2)
                    line["c2"] = {"workload": "C2: 1024x1024, 50 masks, relation-query transformer only, 1xMI355X bf16",
                                  "value": round(pairs_per_image / el2, 1), "unit": "pairs/s", "ms_per_step": round(el2 * 1e3, 3),
                                  "steps": k2, "dtype": "bf16",
                                  "roofline": {"bound": "mfma", "achieved": round(fl2 / el2 / 1e12, 1), "peak": 2500.0,
                                               "unit": "TFLOP/s", "frac": round(fl2 / el2 / 2.5e15, 4),
                                               "executed_tflops": round(fx2 / el2 / 1e12, 1),
                                               "executed_frac": round(fx2 / el2 / 2.5e15, 4), "flops_per_step": int(fl2)}}
                    del h2
                    torch.cuda.empty_cache()
                    try:
                        line["c2"]["cross_attention"] = cross_attention_figures(dev, N, a.size)
                    except Exception as exc:
                        line["c2"]["cross_attention"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
                except Exception as exc:                               # never lose the headline line
                    line["c2"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        if strong is not None:
            line["strong_scaling"] = strong
        if strong_rq is not None:
            line["strong_scaling_rq"] = strong_rq
        oracle_part = None
        if not a.no_cpu_baseline and single:
            scene_cpu = make_scene((a.size, a.size), N, seed=0)
            line["cpu_baseline"], oracle_part = cpu_baseline(a, scene_cpu)
        if not a.no_parity and single and a.images_per_step == 1 and oracle_part is not None:
            try:
                # the GPU scene is generated on the device (another generator than the CPU scene's): the parity
                # legs run the CPU scene, uploaded, so that oracle and kernels see identical inputs
                sc = make_scene((a.size, a.size), N, seed=0)
                sc_dev = dict(sc, mask_features=sc["mask_features"].to(dev), pan_results=sc["pan_results"].to(dev))
                line["parity"] = parity_block(a, dev, sc_dev, oracle_part)
            except Exception as exc:                                   # never lose the headline line
                line["parity"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(line), flush=True)
    if not single:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
