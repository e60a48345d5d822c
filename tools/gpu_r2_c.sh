#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
for ab in 9 25 41 57; do
  echo "#### ABLATE=$ab (9 = no loads, no epilogue; +16 no step barrier; +32 no fragment reads)" >> $O/ablate.log
  VGH_LIB_PATH=$EXP VGH_CONV_ABLATE=$ab timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 64,160,160,96,96,3,1 --cfgs p16x16x64_n4x1,p8x32x96_n4x1,p8x40x64_n5x1,p16x16x128_n4x2,p16x16x128_n4x1,p16x32x128_n4x2,p16x16x256_n4x4 --iters 30 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
echo done > $O/done
