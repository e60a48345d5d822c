#!/bin/bash
# MFMA-busy of the FLAME vertex kernels at n = 8192 (all coefficients): rocprofv3 --pmc, kernel-trace only
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/pmcf
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
FLAME_NS=8192 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -o f -- python $GRAFT_REPO_ROOT/tools/flame_sweep.py > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmcf'
f=glob.glob(O+'/f_counter_collection.csv')[0]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    per[(__import__('re').search(r'(flame_\w+(<[^>]*>)?)', r['Kernel_Name']) or [None, 'other'])[1]][r['Counter_Name']].append(float(r['Counter_Value']))
with open(O+'/r02_pmc_flame.txt','w') as out:
    out.write('# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE around `FLAME_NS=8192 python tools/flame_sweep.py` (three live-coefficient settings, 25 decodes each)\n')
    out.write('# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); v_mfma_f32_32x32x2_f32 = 64 busy cycles\n')
    for k,v in per.items():
        if 'flame' not in k: continue
        b=v['SQ_VALU_MFMA_BUSY_CYCLES']; a=v['GRBM_GUI_ACTIVE']
        fr=[x/(y/8*1024) for x,y in zip(b,a) if y>0]
        line='%-42s %4d dispatches  MFMA busy min %.1f %%  median %.1f %%  max %.1f %%'%(k,len(fr),100*min(fr),100*sorted(fr)[len(fr)//2],100*max(fr))
        print(line); out.write(line+'\n')
PY
