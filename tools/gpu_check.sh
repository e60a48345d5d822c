#!/bin/bash
# One gpurun call: GPU parity tests + smoke + short bench (+ per-layer table). Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" > gpurun_out/env.log; (rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; nproc; python -c "import torch;print(torch.__version__, torch.cuda.is_available())") >> gpurun_out/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --per-layer gpurun_out/per_layer.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log; tail -5 gpurun_out/bench.log
