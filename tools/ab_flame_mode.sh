#!/bin/bash
# A/B of the FLAME vertex-kernel family INSIDE the benchmark step (the decode runs on the post-stage stream beside the next forward's convolutions): the default bench
# command with vgh_flame_set_matrix_path(MODE) applied first, modes alternating on one box.   usage: tools/ab_flame_mode.sh <outdir> "1 2 1 2"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/${1:-abflame}; mkdir -p $O
: > $O/ab_flame_mode.txt
for m in ${2:-1 2 1 2}; do
  python - $m > $O/b_$m.json 2> $O/b_$m.err <<'PY'
import runpy, sys
mode = int(sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-accuracy", "--no-secondary", "--traffic", "off"]
from head_detector_amd import _lib
_lib.check(_lib.load().vgh_flame_set_matrix_path(mode))
runpy.run_path("bench.py", run_name="__main__")
PY
  python - $m $O/b_$m.json >> $O/ab_flame_mode.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"flame mode {sys.argv[1]}: {d['value']:.1f} img/s, {d['config']['ms_per_forward']} ms per forward (net {d['config']['net_ms_per_forward']}), decode n=96 {d['config']['flame_decode_us_per_head_n96']} us/head")
PY
done
cat $O/ab_flame_mode.txt
