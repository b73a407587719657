#!/bin/bash
# A/B of the decode GEMM's LDS footprint with two images in flight: can two streams' launches share a CU?
# Each line: the bench's headline (two in flight) under one kernel-plan setting.  Output: gpurun_out/colocate_ab.txt
out=gpurun_out/colocate_ab.txt
mkdir -p gpurun_out; : > $out
B="python bench.py --steps 16 --warmup 4 --no-batched --no-cpu-baseline --no-roofline --no-untruncated --no-parity"
run() { echo "== $1" >> $out; env $1 timeout 300 $B 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print(d['ms_per_step'], d['value'], (d.get('one_image_at_a_time') or {}))
except Exception as e:
    print('ERR', l[-300:])" >> $out; }
run "PSG_NOP=1"
run "PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0"
run "PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8"
run "PSG_SKINNY_DMA=416 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8"
run "PSG_SKINNY_DMA=424 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8"
run "PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=16"
run "PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8"
cat $out
