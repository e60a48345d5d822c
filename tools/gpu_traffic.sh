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
import json
if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
    # one stem launch per lane per forward: the 3 bench forwards (warmup 1 + steps 2) run on 2 lanes, the untimed threshold
    # calibration before them on one -> forwards of work = stem launches - 3 * (lanes - 1)
    lanes, bench_forwards = 2, 3
    fw = tot['FETCH_SIZE'][1].get('stem', 1) - bench_forwards * (lanes - 1)
    rd = (tot['FETCH_SIZE'][0].get('conv', 0) + tot['FETCH_SIZE'][0].get('stem', 0)) * 1024 * 2 / fw   # KiB -> B; x2: gfx950 FETCH_SIZE counts half of a wide read stream (MI355X_MICROARCH.md)
    wr = (tot['WRITE_SIZE'][0].get('conv', 0) + tot['WRITE_SIZE'][0].get('stem', 0)) * 1024 / fw
    launches = (tot['FETCH_SIZE'][1].get('conv', 0) + tot['FETCH_SIZE'][1].get('stem', 0)) / fw
    d = dict(workload='vgg_heads_m bf16 batch 32 @ 640x640, two batch-split lanes', forwards_in_run=fw, conv_and_stem_launches_per_forward=launches,
             read_bytes_per_forward=rd, write_bytes_per_forward=wr, traffic_bytes_per_forward=rd + wr, traffic_bytes_per_launch=(rd + wr) / launches,
             method='rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes around `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`; counters in KiB summed over the conv + stem dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as is')
    json.dump(d, open('gpurun_out/traffic_m32.json', 'w'), indent=1)
    print(json.dumps(d, indent=1))
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
