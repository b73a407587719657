#!/bin/bash
# Round-5 evidence (run on the MI355X box from the repo root): the headline is the fp32s mode now.
#   tools/collect_profiles_r05.sh     -> gpurun_out/r05_prof/
set -u
TAG=r05
OUT=$PWD/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
run() { local d=$1; shift; rm -rf "$d"; rocprofv3 "$@" > "$d.log" 2>&1; }
# 1. the default command, as the driver runs it
python bench.py --steps 20 --warmup 5 2> "$OUT/bench_default.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_default_line.json"
# 2. per-kernel durations of the headline path (kernel trace + stats; no counters in this pass)
run /tmp/p_full --kernel-trace --stats -d /tmp/p_full -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-mixed --no-exact --no-frozen16
python tools/prof_summary.py "$(db /tmp/p_full)" "$OUT/${TAG}_bench_full_kernel_stats.csv" > /dev/null
grep '^{"metric"' /tmp/p_full.log | tail -1 > "$OUT/${TAG}_bench_full_line.json"
# 3. dominant kernel (fp32 decode GEMM): HBM traffic, two PMC passes
run /tmp/p_fetch32 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
run /tmp/p_write32 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
python tools/pmc_summary.py "$(db /tmp/p_fetch32)" "$(db /tmp/p_write32)" "$OUT/pmc_skinny_gemm_f32.json" 4 > /dev/null
# 3b. the same for the 2-byte stream of fp16-valued weights (psg_split_gemm_w16) + kernel stats of that configuration
run /tmp/p_fetchw --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetchw -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinnysplit
run /tmp/p_writew --pmc WRITE_SIZE --kernel-trace -d /tmp/p_writew -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinnysplit
python tools/pmc_summary.py "$(db /tmp/p_fetchw)" "$(db /tmp/p_writew)" "$OUT/pmc_split_gemm_w16.json" 2 batch_gemm > /dev/null
run /tmp/p_fullw --kernel-trace --stats -d /tmp/p_fullw -- python bench.py --steps 5 --warmup 2 --llm-values fp16 --no-cpu-baseline --no-parity --no-mixed --no-exact --no-frozen16
python tools/prof_summary.py "$(db /tmp/p_fullw)" "$OUT/${TAG}_bench_w16_kernel_stats.csv" > /dev/null
grep '^{"metric"' /tmp/p_fullw.log | tail -1 > "$OUT/${TAG}_bench_w16_line.json"
# 4. BASELINE C2 (bf16 relation query) per step, and the 16-bit cross-attention counters at C2
bash tools/per_image_profile.sh "$OUT/${TAG}_per_step_rq_kernels.csv" --workload rq
f="$OUT/${TAG}_xattn_pmc_n50.txt"
echo "# cross_attn_dma_kernel, N=50 objects, L=256, 12 heads, bf16; tools/bench_kernels.py xattn only50" > "$f"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  run /tmp/p_x50_$i --pmc $grp --kernel-trace -d /tmp/p_x50_$i -- python tools/bench_kernels.py xattn only50
  python tools/pmc_kernel.py cross_attn_dma "$(db /tmp/p_x50_$i)" >> "$f"
done
python tools/prof_summary.py "$(db /tmp/p_x50_1)" | grep cross_attn >> "$f"
python tools/xattn_derive.py "$f" 50 > /dev/null
# 5. the fp32 cross-attention inside the fp32s relation query: duration + matrix-pipe counters
f="$OUT/${TAG}_xattn_f32_pmc.txt"
echo "# cross_attn_f32_kernel inside bench.py --workload rq --dtype fp32s (C2 scene: 2500 pairs); averages per dispatch record" > "$f"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  run /tmp/p_xf_$i --pmc $grp --kernel-trace -d /tmp/p_xf_$i -- python bench.py --workload rq --dtype fp32s --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-roofline
  python tools/pmc_kernel.py cross_attn_f32 "$(db /tmp/p_xf_$i)" >> "$f"
done
python tools/prof_summary.py "$(db /tmp/p_xf_1)" | grep -E "cross_attn_f32|qformer_self_attn_f32" >> "$f"
ls -la "$OUT"
