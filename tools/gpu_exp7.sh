#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/exp7.log
: > $L
timeout 900 python -m pytest tests -m gpu -q -x -k "conv or network_every_op" -p no:cacheprovider 2>&1 | tail -3 >> $L
PC="15,17,19,20,22,23,55,57,60,75,76"
echo "== B=32 80x80 128->128" >> $L
python tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $PC --iters 40 >> $L 2>&1
echo "== B=32 160x160 64->64" >> $L
python tools/conv_bench.py --shape 32,160,160,64,64,3,1 --cfgs 15,19,23,57,76,24 --iters 40 >> $L 2>&1
echo "== B=32 80x80 192->384 (heads)" >> $L
python tools/conv_bench.py --shape 32,80,80,192,384,3,1 --cfgs 17,20,55,75,0 --iters 20 >> $L 2>&1
echo "== B=32 40x40 256->256" >> $L
python tools/conv_bench.py --shape 32,40,40,256,256,3,1 --cfgs 19,20,17,15,55,29,74 --iters 40 >> $L 2>&1
cat $L
