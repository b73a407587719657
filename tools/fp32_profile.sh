#!/bin/bash
# Per-image kernel time of the fp32 (reference-precision) mode: two rocprofv3 kernel traces of tools/fp32_mode.py that
# differ only in the number of timed steps; PSG_MODE=fp32s profiles the split-fp16 variant (tools/per_image_diff.py divides the difference by the extra steps).
#   tools/fp32_profile.sh [out.csv]        (run on the MI355X box from the repo root)
set -u
OUT=${1:-gpurun_out/r04_fp32_per_image_kernels.csv}
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
for n in 1 5; do
  rm -rf /tmp/pf_$n
  rocprofv3 --kernel-trace -d /tmp/pf_$n -- python tools/fp32_mode.py $n 1 > /tmp/pf_$n.log 2>&1
  tail -1 /tmp/pf_$n.log
done
python tools/per_image_diff.py "$(db /tmp/pf_1)" "$(db /tmp/pf_5)" 4 "$OUT" > /dev/null
head -30 "$OUT"
