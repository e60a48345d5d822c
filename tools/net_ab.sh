#!/bin/bash
# Whole-network A/B of library builds on ONE box, alternating (tools/net_probe.py: two-lane forward, table of the tree):
#   gpurun -- 'VARS="vOLD vH" ROUNDS=3 tools/net_ab.sh r4o'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$ROOT
: > $O/net_ab.txt
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in default ${VARS:-}; do
    L=$ROOT/head_detector_amd/libvgh.so; [ $v != default ] && L=$ROOT/head_detector_amd/libvgh_$v.so
    for w in ${WORKLOADS:-"vgg_heads_l 64" "vgg_heads_m 32"}; do
      echo "$v $(VGH_LIB_PATH=$L python tools/net_probe.py $w 2>&1 | grep -v amdgpu | tail -1)" >> $O/net_ab.txt
    done
  done
done
python3 - $O/net_ab.txt <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+) B=(\d+).*?: ([\d.]+) ms/forward", l)
    if m: d[(m.group(2), m.group(3), m.group(1))].append(float(m.group(4)))
for k in sorted(d): print(f"{k[0]} b{k[1]} {k[2]:8s} min {min(d[k]):.3f} med {sorted(d[k])[len(d[k])//2]:.3f} ms/forward  (n={len(d[k])})")
PY
