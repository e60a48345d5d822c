#!/bin/bash
# PMC of the pure compute loop (VGH_CONV_ABLATE=9: no tile loads, no epilogue) vs the full kernel
set -u
mkdir -p gpurun_out/pmc4
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; cfg=$2; ab=$3; shift 3; VGH_CONV_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc4 -o ${tag}_c${cfg}_a$ab -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --shape 32,80,80,128,128,3,1 --cfgs $cfg --iters 5 > /dev/null 2>&1; }
for cfg in 15 55 0; do for ab in 9 0; do
run s1 $cfg $ab SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run s2 $cfg $ab SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run s3 $cfg $ab SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
done; done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, re
out=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc4/*_counter_collection.csv')):
    m=re.search(r'_c(\d+)_a(\d+)_counter', f); key=(int(m.group(1)),int(m.group(2)))
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name'] or 'patch' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in acc.items(): out[key][c]=sum(v)/len(v)
for key,d in sorted(out.items()):
    cyc=d.get('GRBM_GUI_ACTIVE',0)/8
    print('cfg %d ablate %d: cycles/XCD %.4g  MFMA util %.1f%%  wave-cycles %.3g  wait_any/wave %.2f  wait_lds/wave %.2f  active_any/wave %.2f valu %.2f lds %.2f misc %.2f' % (key[0],key[1],cyc,
        100*d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(cyc*1024,1), d.get('SQ_WAVE_CYCLES',0), d.get('SQ_WAIT_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1), d.get('SQ_WAIT_INST_LDS',0)/max(d.get('SQ_WAVE_CYCLES',1),1),
        d.get('SQ_ACTIVE_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1), d.get('SQ_ACTIVE_INST_VALU',0)/max(d.get('SQ_WAVE_CYCLES',1),1), d.get('SQ_ACTIVE_INST_LDS',0)/max(d.get('SQ_WAVE_CYCLES',1),1), d.get('SQ_ACTIVE_INST_MISC',0)/max(d.get('SQ_WAVE_CYCLES',1),1)))
PY
