#!/bin/bash
# One gpurun call for a tuning round: (optional) tests, tune tile configs, bench, rocprof kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${RUN_TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -${TAILN:-40} gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
fi
if [ "${RUN_TUNE:-1}" = "1" ]; then
  timeout 600 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --report gpurun_out/tune_m32.json > gpurun_out/tune.log 2>&1
  timeout 600 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --report gpurun_out/tune_l64.json >> gpurun_out/tune.log 2>&1
  timeout 300 python tools/tune_conv.py --variant vgg_heads_l --batch 1 --report gpurun_out/tune_l1.json >> gpurun_out/tune.log 2>&1
  tail -5 gpurun_out/tune.log
  cp head_detector_amd/tuning/conv_cfg.json gpurun_out/conv_cfg.json
fi
timeout 600 python bench.py --steps ${BENCH_STEPS:-20} --warmup 3 --per-layer gpurun_out/per_layer.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
timeout 300 python bench.py --steps 10 --warmup 3 --variant vgg_heads_l --batch 64 --no-cpu-baseline > gpurun_out/bench_l64.log 2>&1; tail -2 gpurun_out/bench_l64.log
if [ "${RUN_PROF:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1)
  find gpurun_out/prof -name "*stats*" | head; ls -la gpurun_out/prof | head
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
fi
