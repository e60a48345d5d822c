#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
for burst in 0 40 400; do
VGH_LIB_PATH=$EXP timeout 300 python tools/conv_trace.py --burst $burst --shape 64,80,80,128,128,3,1 --cfgs p16x16x64_n4x1,q16x16x64_n4x1 2>&1 | grep -v amdgpu.ids >> $O/trace2.log
done
rocm-smi --showclocks --showpower 2>&1 | head -30 >> $O/trace2.log
echo done > $O/done
