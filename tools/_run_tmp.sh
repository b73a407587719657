export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
for n in 1 5; do rm -rf /tmp/pf_$n; PSG_MODE=fp32s rocprofv3 --kernel-trace -d /tmp/pf_$n -- python tools/fp32_mode.py $n 1 > /tmp/pf_$n.log 2>&1; tail -1 /tmp/pf_$n.log; done
python tools/per_image_diff.py "$(db /tmp/pf_1)" "$(db /tmp/pf_5)" 4 gpurun_out/r04_fp32s_per_image_kernels.csv > /dev/null
head -24 gpurun_out/r04_fp32s_per_image_kernels.csv | cut -c1-150; tail -1 gpurun_out/r04_fp32s_per_image_kernels.csv
