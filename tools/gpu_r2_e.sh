#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv" 2>&1 | tail -25 ) > $O/pytest_conv.log
SHAPES="64,80,80,128,128,3,1 64,160,160,96,96,3,1 64,40,40,256,256,3,1 64,80,80,256,256,3,1 32,80,80,128,128,3,1"
CFGS="p16x16x64_n4x1,q16x16x64_n4x1,p16x16x128_n4x2,q16x16x128_n4x2,p8x32x96_n4x1,q8x32x96_n4x1,p16x16x96_n4x1,q16x16x96_n4x1,p8x40x64_n5x1,q8x40x64_n5x1,p8x40x128_n5x2,q8x40x128_n5x2,p16x16x256_n4x4,q16x16x256_n4x4,p8x32x64_n4x2,q8x32x64_n4x2,p8x32x64_n4x1,q8x32x64_n4x1"
timeout 300 python tools/conv_bench.py --shape $SHAPES --cfgs $CFGS --iters 30 2>&1 | grep -v amdgpu.ids > $O/conv_ab.log
echo "#### +res" >> $O/conv_ab.log
timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 64,160,160,96,96,3,1 --cfgs $CFGS --iters 30 --res 2>&1 | grep -v amdgpu.ids >> $O/conv_ab.log
for ab in 0 1 8 9 11; do
  echo "#### ABLATE=$ab" >> $O/ablate.log
  VGH_LIB_PATH=$EXP VGH_CONV_ABLATE=$ab timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 --cfgs p16x16x64_n4x1,q16x16x64_n4x1 --iters 30 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
echo done > $O/done
