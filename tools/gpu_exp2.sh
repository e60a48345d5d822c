#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/exp2.log
: > $L
PC="15,17,19,20,22,23,55,57,60,75,76"
echo "== B=32 80x80 128->128" >> $L
python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $PC --iters 40 >> $L 2>&1
for ab in 8 9; do
  echo "== ABLATE=$ab" >> $L
  VGH_CONV_ABLATE=$ab python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs 19,55,75 --iters 40 >> $L 2>&1
done
echo "== B=32 160x160 64->64" >> $L
python tools/conv_bench.py --shape 32,160,160,64,64,3,1 --cfgs 15,19,23,57,76,24 --iters 40 >> $L 2>&1
echo "== B=32 80x80 192->192 (heads)" >> $L
python tools/conv_bench.py --shape 32,80,80,192,384,3,1 --cfgs $PC,0,25,26,27,28,29,30 --iters 20 >> $L 2>&1
cat $L
