#!/bin/bash
# PMC counters for one conv shape (separate passes; never combined with sys/hip traces).
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
SHAPE=${SHAPE:-32,80,80,128,128,3,1}
CFG=${CFG:-0}
cd /tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o $tag -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape $SHAPE --cfgs $CFG --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log 2>&1; }
run sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(os.path.basename(f), k)
        for c,vals in v.items(): print('    %-28s mean %.4g  (n=%d)'%(c, sum(vals)/len(vals), len(vals)))
PY
