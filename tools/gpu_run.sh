#!/bin/bash
# One parametrised GPU-box script (replaces the per-experiment gpu_*.sh pile):
#     tools/gpu_run.sh <out-subdir> <step> [<step> ...]        e.g.  gpurun -- 'tools/gpu_run.sh r3a tests bench prof pmc'
# steps:  tests      pytest -m gpu (PYTEST_K='expr' to narrow)             smoke      __graft_entry__.smoke()
#         bench      default bench line + per-layer table             prof       rocprofv3 --kernel-trace --stats of the bench command
#         pmc        HBM traffic + MFMA-busy PMC passes around tools/traffic_run.py, one and two lanes (separate --pmc passes, kernel-trace only)
#         tune       retune the L b64 / M b32 tile tables             tune1280   retune the L b16 @1280 bucket
#         tunesplit  tile table of the fp16x3 parity mode (L b32)
#         flame      FLAME decode sweep (tools/flame_sweep.py)        pmcflame   MFMA-busy of the FLAME kernels at n = 8192
#         probe      whole-net time of both benchmark buckets (two lanes)      py         python $PY_CMD (any tools/ script)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$ROOT
TAG=${TAG:-r03}
FW=${FORWARDS:-8}
for step in "$@"; do
  echo "=== $step"
  case $step in
    tests)
      if [ -n "${PYTEST_K:-}" ]; then
        timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} -k "$PYTEST_K" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
      else
        timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
      fi
      tail -${TAILN:-25} $O/pytest_gpu.log ;;
    smoke)
      timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log ;;
    bench)
      timeout 1200 python bench.py --per-layer $O/per_layer_l64.json ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err; cut -c1-900 $O/bench.json ;;
    prof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-accuracy --no-secondary --traffic off ${PROF_ARGS:-} > $O/prof.log 2>&1)
      f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_bench_${PROF_NAME:-l64}_kernel_stats.csv && head -14 "$f" | cut -c1-220
      tail -1 $O/prof.log | cut -c1-300 ;;
    pmc)
      PV=${PMC_VARIANT:-vgg_heads_l}; PB=${PMC_BATCH:-64}
      for L in ${PMC_LANES:-1 2}; do
        for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
          t=$(echo $c | cut -d' ' -f1)
          (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_${PV}_x$L -o $t -- python $ROOT/tools/traffic_run.py --variant $PV --batch $PB --forwards $FW --split $L > $O/pmc_${PV}_x${L}_$t.log 2>&1)
        done
        python tools/pmc_summary.py $O/pmc_${PV}_x$L $PV $PB $FW $L $O $TAG
      done ;;
    tune)
      timeout 1500 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --report $O/${TAG}_tune_l64.json > $O/tune.log 2>&1
      timeout 1500 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --report $O/${TAG}_tune_m32.json >> $O/tune.log 2>&1
      tail -3 $O/tune.log; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    tunesplit)
      timeout 1500 python tools/tune_conv.py --variant vgg_heads_l --batch 32 --precision fp16x3 --report $O/${TAG}_tune_fp16x3_l32.json > $O/tunesplit.log 2>&1
      tail -2 $O/tunesplit.log; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    tune1280)
      timeout 1500 python tools/tune_conv.py --variant vgg_heads_l --batch 16 --image-size 1280 --report $O/${TAG}_tune_l16_1280.json > $O/tune1280.log 2>&1
      tail -2 $O/tune1280.log; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    flame)
      timeout 900 python tools/flame_sweep.py $O/${TAG}_flame_sweep.json > $O/flame.log 2>&1; grep -v amdgpu $O/flame.log | tail -${TAILN:-30} ;;
    proflame)
      # per-kernel split of the FLAME decode at FLAME_NS heads (prologue vs vertex kernel)
      (cd /tmp && FLAME_NS=${FLAME_NS:-512} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proflame -o f -- python $ROOT/tools/flame_sweep.py > $O/proflame.log 2>&1)
      f=$(find $O/proflame -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-260 ;;
    pmcflame)
      (cd /tmp && FLAME_NS=8192 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmcf -o f -- python $ROOT/tools/flame_sweep.py > $O/pmcflame.log 2>&1)
      python tools/pmc_flame_summary.py $O/pmcf $O/${TAG}_pmc_flame.txt ;;
    pmcreq)
      # which request sizes the L2 -> fabric read counters distinguish on this box, then the request mix of one forward per op (calibrates FETCH_SIZE)
      rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_R[A-Z0-9_]*\|TCC_EA_R[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt
      cat $O/tcc_counters.txt | tr '\n' ' '; echo
      for c in ${REQ_COUNTERS:-"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum"}; do
        t=$(echo $c | cut -d' ' -f1)
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmcreq -o $t -- python $ROOT/tools/traffic_run.py --forwards 2 --split 1 > $O/pmcreq_$t.log 2>&1); tail -2 $O/pmcreq_$t.log | cut -c1-200
      done ;;
    pmcknob)
      # per-op FETCH_SIZE with a library knob set (compare with the pmc step's single-lane pass through tools/pmc_per_op.py)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_knob -o FETCH_SIZE -- python $ROOT/tools/traffic_run.py --forwards 2 --split 1 --knob ${KNOB:-vgh_conv_set_nt_store}=1 > $O/pmc_knob_F.log 2>&1)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_knob -o WRITE_SIZE -- python $ROOT/tools/traffic_run.py --forwards 2 --split 1 --knob ${KNOB:-vgh_conv_set_nt_store}=1 > $O/pmc_knob_W.log 2>&1)
      python tools/pmc_per_op.py $O/pmc_knob vgg_heads_l 64 2 $O/per_op_knob.txt; tail -1 $O/per_op_knob.txt ;;
    tunet)
      # the streaming 1x1 tiles against the current table, both benchmark buckets (report only unless TUNE_WRITE=1)
      W=$([ "${TUNE_WRITE:-0}" = 1 ] && echo "" || echo "--no-write")
      SP=${SPLIT:-1}
      timeout 900 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --only ${ONLY:-t} --split $SP $W --report $O/${TAG}_tune_${ONLY:-t}_l64x$SP.json > $O/tunet.log 2>&1
      timeout 900 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --only ${ONLY:-t} --split $SP $W --report $O/${TAG}_tune_${ONLY:-t}_m32x$SP.json >> $O/tunet.log 2>&1
      timeout 900 python tools/tune_conv.py --variant vgg_heads_l --batch 16 --image-size 1280 --only ${ONLY:-t} --split $SP $W --report $O/${TAG}_tune_${ONLY:-t}_l16_1280x$SP.json >> $O/tunet.log 2>&1
      grep -v amdgpu $O/tunet.log | tail -${TAILN:-70}; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    tunee2e)
      # per-op report of EVERY tile with two lanes (no table write), then coordinate descent on the whole two-lane forward from the current table
      V=${E2E_VARIANT:-vgg_heads_l}; B=${E2E_BATCH:-64}; S=${E2E_SIZE:-640}; PR=${E2E_PRECISION:-bf16}
      timeout 1200 python tools/tune_conv.py --variant $V --batch $B --image-size $S --split 2 --reps 3 --no-write --precision $PR --report $O/${TAG}_tune_${V}_b${B}x2.json > $O/tunee2e.log 2>&1
      timeout 1200 python tools/tune_e2e.py --variant $V --batch $B --image-size $S --split 2 --precision $PR --report $O/${TAG}_tune_${V}_b${B}x2.json --topk ${TOPK:-4} --log $O/${TAG}_tune_e2e_${V}_b${B}.json >> $O/tunee2e.log 2>&1
      grep -v amdgpu $O/tunee2e.log | tail -${TAILN:-30}; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    abtable)
      # the committed table before this step's retune (tools/_prev_table.json, untracked) against the current one, alternating on one engine
      timeout 900 python tools/ab_table.py ${PREV_TABLE:-tools/_prev_table.json} ${NEW_TABLES:-head_detector_amd/tuning/conv_cfg.json} --rounds ${ROUNDS:-4} > $O/abtable.log 2>&1; grep -v amdgpu $O/abtable.log | tail -8 ;;
    abknob)
      timeout 900 python tools/ab_knob.py ${KNOB:-vgh_conv_set_nt_store} --json $O/${TAG}_ab_${KNOB:-vgh_conv_set_nt_store}.json ${KNOB_ARGS:-} > $O/abknob.log 2>&1; grep -v amdgpu $O/abknob.log | tail -${TAILN:-60} ;;
    convbench)
      # single-shape A/B of tile configurations through vgh_conv2d (CB_SHAPES / CB_CFGS / CB_ARGS)
      timeout 900 python tools/conv_bench.py --shape ${CB_SHAPES:-64,80,80,128,128,3,1} --cfgs ${CB_CFGS:-all} --iters ${CB_ITERS:-30} ${CB_ARGS:-} > $O/convbench${CB_TAG:-}.log 2>&1; grep -v amdgpu $O/convbench${CB_TAG:-}.log | tail -${TAILN:-80} ;;
    py)
      # any tools/ script:  PY_CMD="tools/ab_stem_pitch.py --rounds 4"  (log named by PY_TAG)
      timeout ${PY_TIMEOUT:-900} python ${PY_CMD} > $O/py${PY_TAG:-}.log 2>&1; echo "py rc=$?" >> $O/py${PY_TAG:-}.log; grep -v amdgpu $O/py${PY_TAG:-}.log | tail -${TAILN:-40} ;;
    tune16)
      # tile table of the single-plane fp16 mode (keys "fp16:..."), two lanes, both benchmark buckets; the merged table comes back as $O/conv_cfg.json
      timeout 1200 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --precision fp16 --split 2 --reps 3 --report $O/${TAG}_tune_fp16_l64x2.json > $O/tune16.log 2>&1
      timeout 900 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --precision fp16 --split 2 --reps 3 --report $O/${TAG}_tune_fp16_m32x2.json >> $O/tune16.log 2>&1
      grep -v amdgpu $O/tune16.log | tail -4; cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json ;;
    benchm)
      # BASELINE configs[1] (VGGHeads_M b32) as the MAIN workload of bench.py: its own line, per-layer table and (PROF_ARGS / PMC_VARIANT) kernel stats / PMC passes
      timeout 900 python bench.py --variant vgg_heads_m --batch 32 --no-secondary --no-cpu-baseline --no-accuracy --per-layer $O/${TAG}_per_layer_m32.json ${BENCH_ARGS:-} > $O/bench_m32.json 2> $O/bench_m32.err; echo "benchm rc=$?"; tail -3 $O/bench_m32.err; cut -c1-700 $O/bench_m32.json ;;
    probe)
      python tools/net_probe.py vgg_heads_l 64 2>&1 | grep -v amdgpu; python tools/net_probe.py vgg_heads_m 32 2>&1 | grep -v amdgpu ;;
    *) echo "unknown step $step" ;;
  esac
done
