set -u
O=$GRAFT_REPO_ROOT/gpurun_out/post
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-accuracy --no-secondary > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/post/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
stems=[i for i,r in enumerate(rows) if 'stem_kernel' in r['Kernel_Name']]
seg=rows[stems[-17]:stems[-1]]
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    n=r['Kernel_Name']
    if 'conv' in n or 'stem' in n or 'spp' in n: continue
    k=n.split('(anonymous namespace)::')[-1][:40]
    agg[k][0]+=1; agg[k][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"{v[0]:4d} calls avg {v[1]/v[0]/1e3:7.1f} us  {k}"); tot+=v[1]/8
print('post stages per step: %.1f us'%(tot/1e3))
# gaps between last conv of a step and first conv of next
PY
