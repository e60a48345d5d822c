#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
for mt in 1 2; do
echo "### VGH_FLAME_MT=$mt" >> $O/sweep2.log
VGH_LIB_PATH=$EXP VGH_FLAME_MT=$mt timeout 600 python tools/flame_sweep.py 2>&1 | grep "n': 96\|n': 1024\|n': 8192" >> $O/sweep2.log
done
echo done > $O/done
