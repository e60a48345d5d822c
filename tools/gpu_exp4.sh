#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/exp4.log
: > $L
echo "== mfma peak calibration" >> $L
[ -x tools/micro/mfma_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip
for bpc in 1 2 4; do ./tools/micro/mfma_peak $bpc 20000 0.01 >> $L 2>&1; done
./tools/micro/mfma_peak 2 20000 0 >> $L 2>&1
PC="15,17,19,20,22,23,55,57,60,75,76"
echo "== B=32 80x80 128->128" >> $L
python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $PC --iters 40 >> $L 2>&1
echo "== B=32 160x160 64->64" >> $L
python tools/conv_bench.py --shape 32,160,160,64,64,3,1 --cfgs 15,19,23,57,76,24 --iters 40 >> $L 2>&1
echo "== B=32 80x80 192->384 (heads)" >> $L
python tools/conv_bench.py --shape 32,80,80,192,384,3,1 --cfgs $PC,0 --iters 20 >> $L 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -k "conv" -p no:cacheprovider 2>&1 | tail -3 >> $L
cat $L
