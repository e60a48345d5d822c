#!/bin/bash
# HBM traffic of one whole forward (PMC FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only) + torchrun smoke.
set -u
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/traffic -o $c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/traffic/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
tot={}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    f=glob.glob(f'gpurun_out/traffic/{c}_counter_collection.csv')
    if not f: continue
    per=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        key='conv' if ('conv_igemm' in k or 'patch_kernel' in k) else 'stem' if 'stem_kernel' in k else 'other'
        per[key]+=float(r['Counter_Value']); n[key]+=1
    tot[c]=(dict(per), dict(n))
    print(c, {k: round(v/1024,1) for k,v in per.items()}, 'MiB total over the run;', dict(n), 'dispatches')
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
