#!/bin/bash
# Ablation matrix of the ping-pong conv kernel (experiments build) + A/B of the residual L2 touches (variant build).
#   gpurun -- 'tools/pp_ablate.sh r4b'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$ROOT
SH=${SHAPE:-64,80,80,128,128,3,1}
CF=${CFGS:-g8x8x128_n8}
{
echo "# conv_bench --shape $SH --cfgs $CF, experiments build, VGH_CONV_ABLATE bits: 1 no LDS-DMA, 2 no MFMA, 8 no epilogue, 16 no barrier after the L phase, 32 no fragment reads, 64 no barrier after the M phase, 128 epilogue without its stores"
for ab in ${ABLATES:-0 1 32 33 2 3 34 35 8 41 43}; do
  echo "## ablate=$ab"
  VGH_EXPERIMENTS=1 VGH_CONV_ABLATE=$ab VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_exp.so timeout 120 python tools/conv_bench.py --shape $SH --cfgs $CF --iters 40 2>&1 | grep "cfg "
done
} > $O/pp_ablate.txt 2>&1
cat $O/pp_ablate.txt
[ "${SKIP_AB:-0}" = 1 ] && exit 0
{
for r in 1 2 3; do
  echo "## round $r: default (residual L2 touches on) / variant (off)"
  timeout 120 python tools/conv_bench.py --shape $SH --cfgs $CF,p16x16x64_n4x1 --iters 40 --res 2>&1 | grep "cfg "
  VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_var.so timeout 120 python tools/conv_bench.py --shape $SH --cfgs $CF --iters 40 --res 2>&1 | grep "cfg "
  timeout 120 python tools/conv_bench.py --shape $SH --cfgs $CF --iters 40 2>&1 | grep "cfg "
  VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_var.so timeout 120 python tools/conv_bench.py --shape $SH --cfgs $CF --iters 40 2>&1 | grep "cfg "
done
} > $O/pp_ab_res.txt 2>&1
cat $O/pp_ab_res.txt
