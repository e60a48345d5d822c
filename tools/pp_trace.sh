#!/bin/bash
# Cycle-accurate phase timeline of the ping-pong conv tiles (tools/conv_trace.py on the -DVGH_EXPERIMENTS build) with and without each phase's work:
# cycles per barrier slot, MFMA-busy share and the EFFECTIVE shader clock (cycles of the busiest blocks / wall time of the launch) per ablation.
#   gpurun -- 'tools/pp_trace.sh r4z'     (SHAPES / CFG / ABLATES / BURST to vary; needs head_detector_amd/libvgh_exp.so = build --experiments)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export PYTHONPATH=$ROOT TMPDIR=/tmp VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_exp.so VGH_EXPERIMENTS=1
O=$ROOT/gpurun_out/$1; mkdir -p $O
OUT=$O/${TAG:-r04}_pp_trace.txt
echo "# tools/pp_trace.sh: s_memtime marks of thread 0 per tile (tile top, first channel block done, K loop done, epilogue done), last launch of a burst of ${BURST:-300}; VGH_CONV_ABLATE bits: 1 no LDS-DMA, 2 no MFMA, 8 no epilogue, 32 no fragment reads" > $OUT
for sh in ${SHAPES:-64,80,80,128,128,3,1 64,80,80,256,512,3,1 64,160,160,96,96,3,1}; do
  for ab in ${ABLATES:-0 1 32 33 8 41 2}; do
    echo "## shape $sh ablate $ab" >> $OUT
    c=${CFG:-g8x8x128_n8}; [ "${sh##*,96,96,3,1}" != "$sh" ] && c=g8x8x96_n8
    VGH_CONV_ABLATE=$ab timeout 120 python tools/conv_trace.py --shape $sh --cfgs $c --burst ${BURST:-300} 2>&1 | grep -v amdgpu | grep "==\|busiest\|tile  1 " >> $OUT
  done
done
cat $OUT
