#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/exp8.log
: > $L
for cin in 32 64 128 256 512; do
echo "== cin=$cin cout=128 80x80 B=32" >> $L
python tools/conv_bench.py --shape 32,80,80,$cin,128,3,1 --cfgs 15,19,55,57,0,74 --iters 40 >> $L 2>&1
done
cat $L
