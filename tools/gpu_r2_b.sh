#!/bin/bash
# ablation matrix of the patch kernels: bit0 = no LDS-DMA loads in the K loop, bit1 = no fragment reads / MFMAs, bit3 = no epilogue
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
for ab in 0 1 2 3 8 9 10 11; do
  echo "#### ABLATE=$ab" >> $O/ablate.log
  VGH_LIB_PATH=$EXP VGH_CONV_ABLATE=$ab timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 64,160,160,96,96,3,1 64,40,40,256,256,3,1 --cfgs p16x16x64_n4x1,q16x16x64_n4x1,p8x32x96_n4x1,q8x32x96_n4x1,p8x40x64_n5x1,256x128_w64x64_k1_r3,p16x16x128_n4x2 --iters 30 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
echo done > $O/done
