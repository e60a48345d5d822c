#!/bin/bash
# Shader clock and socket power while a conv shape runs in a loop: is the ~0.8-0.9 PFLOP/s plateau of the 128-cout layers a power / clock plateau?
#   tools/power_probe.sh  (GPU box; writes gpurun_out/power_probe.txt)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/power_probe.txt
: > $O
sample() {  # $1 = label; samples until the background job ends
  while kill -0 $2 2>/dev/null; do
    echo "$1 $(rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' ' | cut -c1-400)" >> $O
    sleep 0.3
  done
}
run() {  # label, conv_bench args
  python tools/conv_bench.py $2 --iters ${ITERS:-20000} > gpurun_out/power_$1.log 2>&1 &
  pid=$!
  sleep ${WARM:-12}
  sample $1 $pid
  wait $pid
  grep cfg gpurun_out/power_$1.log | grep -v amdgpu >> $O
}
ITERS=120000 run p16x16x64 "--shape 64,80,80,128,128,3,1 --cfgs p16x16x64_n4x1"
ITERS=80000 run big128 "--shape 64,80,80,128,128,3,1 --cfgs p16x32x128_n4x1"
ITERS=18000 run n512 "--shape 64,80,80,256,512,3,1 --cfgs p16x16x256_n4x4"
ITERS=60000 run one_by_one "--shape 64,160,160,96,192,1,1 --cfgs -1"
grep -c . $O
python - <<'PY'
import re, collections, os
rows = collections.defaultdict(list)
for ln in open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/power_probe.txt"):
    t = ln.split(None, 1)
    if len(t) == 2 and "card" in t[1]:
        rows[t[0]].append(t[1])
for k, v in rows.items():
    print(k, len(v), "samples; last:", v[-1][:300])
PY
grep cfg $O
