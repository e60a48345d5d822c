#!/bin/bash
set -u
mkdir -p gpurun_out
L=gpurun_out/exp6.log
: > $L
echo "== hbm calibration" >> $L
[ -x tools/micro/hbm_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_peak tools/micro/hbm_peak.hip
./tools/micro/hbm_peak >> $L 2>&1
echo "== flame sweep" >> $L
python tools/flame_sweep.py gpurun_out/flame_sweep.json >> $L 2>&1
echo "== bench graph" >> $L
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --graph 2>&1 | grep '^{' >> $L
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' >> $L
cat $L
