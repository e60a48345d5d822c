#!/bin/bash
# MFMA utilisation / LDS conflict counters of the patch kernels after the round-1 changes (separate --pmc passes, kernel-trace only)
set -u
mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; cfg=$2; shift 2; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o ${tag}_c$cfg -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $cfg --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc3/${tag}_c$cfg.log 2>&1; }
for cfg in 15 19 55 0; do
run s1 $cfg SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
run s2 $cfg SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run g1 $cfg GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
out=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc3/**/*_counter_collection.csv', recursive=True)):
    cfg=f.split('_c')[-1].split('_')[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'patch' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); out[cfg]['kernel']=r['Kernel_Name'][:90]
    for c,v in acc.items(): out[cfg][c]=sum(v)/len(v)
for cfg,d in out.items():
    print('cfg',cfg,d.get('kernel'))
    for c,v in sorted(d.items()):
        if c!='kernel': print('   %-34s %.4g'%(c,v))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'SQ_BUSY_CYCLES' in d: print('   MFMA busy / SQ busy (per-SE aggregation: x4 SIMDs) = %.3f'%(d['SQ_VALU_MFMA_BUSY_CYCLES']/d['SQ_BUSY_CYCLES']/4/ (1)))
    if 'SQ_LDS_BANK_CONFLICT' in d and 'SQ_LDS_IDX_ACTIVE' in d: print('   LDS conflict cycles / LDS active = %.3f'%(d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1)))
PY
