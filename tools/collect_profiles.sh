#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the MI355X box from the repo root):
#   tools/collect_profiles.sh <round tag, e.g. r02>
# Kernel traces and PMC counters are separate passes (never combined with other trace domains); every pass is
# summarised on the box (the sqlite databases are too large to travel) into gpurun_out/<tag>_prof/.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
run() { local d=$1; shift; rm -rf "$d"; rocprofv3 "$@" > "$d.log" 2>&1; }

# 1. headline benchmark, per-kernel durations
run /tmp/p_full --kernel-trace --stats -d /tmp/p_full -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-batched
python tools/prof_summary.py "$(db /tmp/p_full)" "$OUT/${TAG}_bench_full_kernel_stats.csv" > /dev/null
grep '^{"metric"' /tmp/p_full.log | tail -1 > "$OUT/${TAG}_bench_full_line.json"
# 2. relation-query-only workload (BASELINE C2)
run /tmp/p_rq --kernel-trace -d /tmp/p_rq -- python bench.py --workload rq --no-cpu-baseline --no-parity --steps 10 --warmup 2
python tools/prof_summary.py "$(db /tmp/p_rq)" "$OUT/${TAG}_bench_rq_kernel_stats.csv" > /dev/null
# 3. dominant kernel: HBM traffic (two PMC passes)
run /tmp/p_fetch --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch -- python tools/bench_kernels.py skinny
run /tmp/p_write --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write -- python tools/bench_kernels.py skinny
python tools/pmc_summary.py "$(db /tmp/p_fetch)" "$(db /tmp/p_write)" "$OUT/pmc_skinny_gemm.json" > /dev/null
# 3b. the fp32 instantiation (psg_gemm_f32.hip): traffic, wave-state counters, per-image tables of both fp32 modes
run /tmp/p_fetch32 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_fetch32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
run /tmp/p_write32 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_write32 -- env PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32
python tools/pmc_summary.py "$(db /tmp/p_fetch32)" "$(db /tmp/p_write32)" "$OUT/pmc_skinny_gemm_f32.json" 4 > /dev/null
PSG_BENCH_SHAPES=gate_up PSG_BENCH_NOLIB=1 bash tools/pmc_cmd.sh skinny_gemm_f32 "$OUT/${TAG}_f32_decode_gemm_pmc_gateup.txt" python tools/bench_kernels.py skinny32 > /dev/null
PSG_BENCH_NOLIB=1 python tools/bench_kernels.py skinny32 skinny32_m4 skinny32_m16 skinny32_m32 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_f32_decode_gemm_bench.txt"
bash tools/fp32_profile.sh "$OUT/${TAG}_fp32_per_image_kernels.csv" > /dev/null
PSG_MODE=fp32s bash tools/fp32_profile.sh "$OUT/${TAG}_fp32s_per_image_kernels.csv" > /dev/null
# 4. cross-attention (LDS-DMA kernel), N = 50 and N = 100: four PMC passes each
for n in 50 100; do
  f="$OUT/${TAG}_xattn_pmc_n$n.txt"
  echo "# cross_attn_dma_kernel, N=$n objects, L=256, 12 heads, bf16; tools/bench_kernels.py xattn only$n" > "$f"
  echo "# rocprofv3 --pmc <counters> --kernel-trace, one pass per group; averages per dispatch record (SQ_*: per XCD)" >> "$f"
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    run /tmp/p_x${n}_$i --pmc $grp --kernel-trace -d /tmp/p_x${n}_$i -- python tools/bench_kernels.py xattn only$n
    python tools/pmc_kernel.py cross_attn_dma "$(db /tmp/p_x${n}_$i)" >> "$f"
  done
  python tools/prof_summary.py "$(db /tmp/p_x${n}_1)" | grep cross_attn >> "$f"
  python tools/xattn_derive.py "$f" $n > /dev/null
done
# 5. what one image / one relation-query step costs per kernel (differential traces)
bash tools/per_image_profile.sh "$OUT/${TAG}_per_image_kernels.csv"
bash tools/per_image_profile.sh "$OUT/${TAG}_per_step_rq_kernels.csv" --workload rq
ls -la "$OUT"
