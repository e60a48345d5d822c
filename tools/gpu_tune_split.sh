#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --split 2 --report gpurun_out/tune_m32x2.json > gpurun_out/tune_split.log 2>&1
timeout 900 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --split 2 --report gpurun_out/tune_l64x2.json >> gpurun_out/tune_split.log 2>&1
tail -3 gpurun_out/tune_split.log
cp head_detector_amd/tuning/conv_cfg.json gpurun_out/conv_cfg.json
timeout 600 python -m pytest tests -m gpu -q -x -k "split or overlap" -p no:cacheprovider 2>&1 | tail -2
for sp in 1 2; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline --split $sp 2>&1 | grep "^{" | cut -c98-130; done
for sp in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant vgg_heads_l --batch 64 --split $sp 2>&1 | grep "^{" | cut -c98-130; done
