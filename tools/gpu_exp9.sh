#!/bin/bash
# exact kernel durations (rocprofv3 kernel trace) of the skeleton variants, persistent vs one tile per block
set -u
mkdir -p gpurun_out/exp9
export TMPDIR=/tmp
cd /tmp
for pe in 1 0; do for ab in 0 11 3; do
  VGH_PATCH_PERSIST=$pe VGH_CONV_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp9 -o p${pe}a$ab -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs 0,15,19,55 --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/exp9/p${pe}a$ab.log 2>&1
done; done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, re
for pe in (1,0):
  for ab in (0,11,3):
    f=glob.glob(f'gpurun_out/exp9/**/p{pe}a{ab}_kernel_stats.csv', recursive=True)
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'conv' in n:
            m=re.search(r'(conv\w+<[^>]*>)', n)
            print(f"persist {pe} ablate {ab:2d} {m.group(1):52s} avg {float(r['AverageNs'])/1e3:7.2f} us")
PY
