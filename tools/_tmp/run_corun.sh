for t in lib 256x128 256x64 128x128 256x192; do
  timeout 200 python tools/corun_probe.py $t 2>&1 | tail -1
  PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=8 timeout 200 python tools/corun_probe.py $t 2>&1 | tail -1
done
PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=16 timeout 200 python tools/corun_probe.py 256x128 2>&1 | tail -1
PSG_SKINNY_DMA=414 PSG_SKINNY_WIDE=0 PSG_SKINNY_SPLITS=16 timeout 200 python tools/corun_probe.py 128x128 2>&1 | tail -1
