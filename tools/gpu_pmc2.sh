#!/bin/bash
# deeper PMC look at one conv config: LDS / VMEM / TA / TCP stall counters. ABL=ablate flags.
set -u
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
SHAPE=${SHAPE:-32,80,80,128,128,3,1}
CFG=${CFG:-0}
ABL=${ABL:-0}
cd /tmp
run() { tag=$1; shift; VGH_CONV_ABLATE=$ABL timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o ${tag}_a$ABL -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape $SHAPE --cfgs $CFG --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc2/${tag}_a$ABL.log 2>&1; }
run s1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD
run s2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run s3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
run t1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run t2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
run t3 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
run t4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run t5 TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum
run t6 TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum
run g1 GRBM_GUI_ACTIVE GRBM_TA_BUSY
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out={}
for f in sorted(glob.glob('gpurun_out/pmc2/*_a%s_counter_collection.csv' % os.environ.get('ABL','0'))):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'patch' in r['Kernel_Name']:
            out.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for c,v in sorted(out.items()): print('%-36s %.4g'%(c, sum(v)/len(v)))
PY
