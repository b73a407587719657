#!/bin/bash
# Round-6 evidence (run on the MI355X box from the repo root).  Both weight kinds of the fp32s mode at equal standing:
# generic fp32 values (the headline) and fp16-valued matrices (the reference's frozen checkpoint, `--llm-values fp16`).
#   tools/collect_profiles_r06.sh     -> gpurun_out/r06_prof/
set -u
TAG=r06
OUT=$PWD/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-parity --no-mixed --no-exact --no-frozen16 --no-batched --no-c4"
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
run() { local d=$1; shift; rm -rf "$d"; rocprofv3 "$@" > "$d.log" 2>&1; }
# 1. the default command, as the driver runs it
python bench.py --steps 20 --warmup 5 2> "$OUT/bench_default.err" | grep '^{"metric"' | tail -1 > "$OUT/${TAG}_bench_default_line.json"
# 2. per-kernel durations of the headline path (kernel trace + stats; no counters in this pass), both weight kinds
run /tmp/p_full --kernel-trace --stats -d /tmp/p_full -- python bench.py --steps 5 --warmup 2 $QUIET
python tools/prof_summary.py "$(db /tmp/p_full)" "$OUT/${TAG}_bench_full_kernel_stats.csv" > /dev/null
grep '^{"metric"' /tmp/p_full.log | tail -1 > "$OUT/${TAG}_bench_full_line.json"
run /tmp/p_fullw --kernel-trace --stats -d /tmp/p_fullw -- python bench.py --steps 5 --warmup 2 --llm-values fp16 $QUIET
python tools/prof_summary.py "$(db /tmp/p_fullw)" "$OUT/${TAG}_bench_w16_kernel_stats.csv" > /dev/null
grep '^{"metric"' /tmp/p_fullw.log | tail -1 > "$OUT/${TAG}_bench_w16_line.json"
# 3. what ONE image costs per kernel (difference of two traces), both weight kinds
bash tools/per_image_profile.sh "$OUT/${TAG}_fp32s_per_image_kernels.csv"
bash tools/per_image_profile.sh "$OUT/${TAG}_w16_per_image_kernels.csv" --llm-values fp16
# 4. dominant kernels: HBM traffic, two PMC passes each (fp32 weight stream; 2-byte stream of fp16-valued weights)
run /tmp/p_fetch32 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
run /tmp/p_write32 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
python tools/pmc_summary.py "$(db /tmp/p_fetch32)" "$(db /tmp/p_write32)" "$OUT/pmc_skinny_gemm_f32.json" 4 > /dev/null
run /tmp/p_fetchw --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetchw -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinnysplit
run /tmp/p_writew --pmc WRITE_SIZE --kernel-trace -d /tmp/p_writew -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinnysplit
python tools/pmc_summary.py "$(db /tmp/p_fetchw)" "$(db /tmp/p_writew)" "$OUT/pmc_split_gemm_w16.json" 2 batch_gemm > /dev/null
# 5. relation query per step: BASELINE C2 (bf16) and the headline mode's (fp32s)
bash tools/per_image_profile.sh "$OUT/${TAG}_per_step_rq_kernels.csv" --workload rq
bash tools/per_image_profile.sh "$OUT/${TAG}_fp32s_per_step_rq_kernels.csv" --workload rq --dtype fp32s
# 6. the 16-bit cross-attention counters at C2 (north star: MFMA utilisation of the relation-query cross-attention)
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
ls -la "$OUT"
