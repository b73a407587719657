#!/bin/bash
# A/B table for the two N = 4096 decode projections (o: K = 4096, down: K = 11008; 32 slabs of 128 rows for 256 CUs):
# split count x kernel geometry (<waves><K blocks per batch><ring slots>), 16-bit kernel and fp32 kernel.
#   tools/o_down_sweep.sh > profiles/r04_o_down_sweep.txt      (on the MI355X box)
export PSG_BENCH_SHAPES=o,down PSG_BENCH_NOLIB=1
echo "# tools/bench_kernels.py skinny (fp16... bf16 operands, M = 20), shapes o and down; default plan first"
python tools/bench_kernels.py skinny 2>&1 | grep "skinny "
for dma in 813 815 823 414 416 424; do
  for s in 4 8 16; do
    echo "## PSG_SKINNY_DMA=$dma PSG_SKINNY_SPLITS=$s"
    PSG_SKINNY_DMA=$dma PSG_SKINNY_SPLITS=$s python tools/bench_kernels.py skinny 2>&1 | grep "skinny \|Error\|failed" | head -3
  done
done
echo "# fp32 kernel (skinny32), default plan, then forced splits"
python tools/bench_kernels.py skinny32 2>&1 | grep "skinny "
for s in 4 8 16; do
  echo "## PSG_SKINNY_SPLITS=$s"
  PSG_SKINNY_SPLITS=$s python tools/bench_kernels.py skinny32 2>&1 | grep "skinny \|Error\|failed" | head -3
done
