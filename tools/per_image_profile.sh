#!/bin/bash
# Per-step kernel time of a bench.py workload: two rocprofv3 kernel traces that differ only in the number of timed
# steps; the difference of the per-kernel totals divided by the extra steps is what ONE step (image) costs (setup,
# warm-up, graph capture and the roofline leg cancel).  Run on the MI355X box from the repo root:
#   tools/per_image_profile.sh [out.csv] [extra bench.py arguments, e.g. --workload rq]
# (Round 6: the library products of the prompt pass run in parts fixed per SHAPE - llm._SPLIT_PLAN_TABLE - so the two
# traces run the same kernels.)
set -u
OUT=${1:-gpurun_out/per_image_kernels.csv}
shift || true
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
for n in 2 10; do
  rm -rf /tmp/pi_$n
  rocprofv3 --kernel-trace -d /tmp/pi_$n -- python bench.py --steps $n --warmup 1 --no-cpu-baseline --no-parity --no-batched --no-mixed --no-exact --no-frozen16 --no-c4 "$@" > /tmp/pi_$n.log 2>&1
done
python tools/per_image_diff.py "$(db /tmp/pi_2)" "$(db /tmp/pi_10)" 8 "$OUT" > /dev/null
