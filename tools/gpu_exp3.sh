#!/bin/bash
# exact kernel durations (rocprofv3 kernel trace) for ablation variants
set -u
mkdir -p gpurun_out/exp3
export TMPDIR=/tmp
cd /tmp
for ab in 0 8 9 11 1 2; do
  VGH_CONV_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp3 -o a$ab -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape ${SHAPE:-32,80,80,128,128,3,1} --cfgs ${CFGS:-0,19,55,57} --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/exp3/a$ab.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, re
for ab in (0,8,9,11,1,2):
    f=glob.glob(f'gpurun_out/exp3/**/a{ab}_kernel_stats.csv', recursive=True)
    if not f: print('missing',ab); continue
    for r in csv.DictReader(open(f[0])):
        n=r['Name']
        if 'conv' in n:
            m=re.search(r'(conv\w+<[^>]*>)', n)
            print(f"ablate {ab:2d} {m.group(1) if m else n[:60]:60s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
