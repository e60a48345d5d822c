#!/bin/bash
# A/B of compile-time knobs of the ping-pong conv kernel: default build against variant builds (libvgh_<tag>.so), alternating, conv_bench shapes.
#   gpurun -- 'VARS="vA vB vC" tools/pp_ab.sh r4d'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$ROOT
CF=${CFGS:-g8x8x128_n8}
{
for r in 1 2 3; do
  for v in default ${VARS:-}; do
    L=$ROOT/head_detector_amd/libvgh.so; [ $v != default ] && L=$ROOT/head_detector_amd/libvgh_$v.so
    echo "## round $r lib $v"
    VGH_LIB_PATH=$L timeout 120 python tools/conv_bench.py --shape ${SHAPES:-64,80,80,128,128,3,1 64,40,40,256,256,3,1} --cfgs $CF --iters 40 2>&1 | grep "cfg \|shape"
    VGH_LIB_PATH=$L timeout 120 python tools/conv_bench.py --shape ${RES_SHAPES:-64,80,80,128,128,3,1} --cfgs $CF --iters 40 --res 2>&1 | grep "cfg \|shape"
  done
done
} > $O/pp_ab.txt 2>&1
python3 - $O/pp_ab.txt <<'PY'
import sys, re, collections
d = collections.defaultdict(list); lib = shape = None
for l in open(sys.argv[1]):
    if l.startswith('## round'): lib = l.split()[-1]
    elif l.startswith('== shape'): shape = l.split('shape')[1].strip()
    elif l.startswith('cfg'):
        p = l.split(); d[(shape, p[2], lib)].append(float(p[3]))
for k in sorted(d): print(f"{k[0]:34s} {k[1]:18s} {k[2]:8s} min {min(d[k])*1e3:7.1f} us  med {sorted(d[k])[len(d[k])//2]*1e3:7.1f} us")
PY
