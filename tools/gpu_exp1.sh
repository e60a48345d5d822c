#!/bin/bash
# experiment: phase ablation of the patch kernel + batch scaling (does the kernel speed up with more rounds of blocks?)
set -u
mkdir -p gpurun_out
L=gpurun_out/exp1.log
: > $L
PC="15,16,17,18,19,20,21,22,23,24,54,55,56,57,58,59,60"
echo "== B=32 all patch cfgs" >> $L
python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $PC --iters 40 >> $L 2>&1
for ab in 1 2 3 8 9 10 11; do
  echo "== B=32 ABLATE=$ab" >> $L
  VGH_CONV_ABLATE=$ab python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs 19,21,56,58 --iters 40 >> $L 2>&1
done
echo "== B=128" >> $L
python tools/conv_bench.py --shape 128,80,80,128,128,3,1 --cfgs $PC --iters 20 >> $L 2>&1
echo "== B=8" >> $L
python tools/conv_bench.py --shape 8,80,80,128,128,3,1 --cfgs $PC --iters 40 >> $L 2>&1
cat $L
