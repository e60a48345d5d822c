#!/bin/bash
# r02 evidence run: GPU tests, the default bench line, rocprofv3 kernel stats of the bench command, HBM traffic + MFMA-busy PMC passes.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r2l
mkdir -p $O
export TMPDIR=/tmp
if [ "${RUN_TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -${TAILN:-15} $O/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
fi
if [ "${RUN_BENCH:-1}" = "1" ]; then
  timeout 900 python bench.py --per-layer $O/per_layer_l64.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
fi
BARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-accuracy --no-secondary"
if [ "${RUN_PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py $BARGS > $O/prof.log 2>&1)
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r02_bench_l64_kernel_stats.csv && head -12 "$f" | cut -c1-200
  tail -1 $O/prof.log | cut -c1-300
fi
if [ "${RUN_PMC:-1}" = "1" ]; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-accuracy --no-secondary > $O/pmc_$tag.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $O/pmc vgg_heads_l 64 $O
fi
