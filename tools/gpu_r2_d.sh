#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
for mb in 0 32 64; do
for ab in 0 9 41; do
  echo "#### ABLATE=$ab max-blocks/xcd=$mb" >> $O/ablate.log
  VGH_LIB_PATH=$EXP VGH_CONV_ABLATE=$ab timeout 300 python tools/conv_bench.py --max-blocks $mb --shape 64,80,80,128,128,3,1 --cfgs p16x16x64_n4x1,p16x16x128_n4x2,p16x16x128_n4x1,p16x16x32_n4x1,p8x32x64_n4x2 --iters 30 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
done
echo done > $O/done
