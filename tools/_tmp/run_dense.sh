set -x
timeout 600 python -m pytest tests/test_gpu_dense_tiles.py -x -q 2>&1 | tail -15
PSG_GEMM_VARIANTS=lib,own:256x256,own:256x192,own:256x128,own:256x64,own:128x128,own:auto,lib+silu,swiglu:256x192,swiglu:256x256,swiglu:256x128 timeout 600 python tools/gemm_shapes_bench.py 980 2>&1 | grep -v "^K'=3K" | tail -12
