#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flame or detect or facade or c_only or independence" 2>&1 | tail -15 ) > $O/pytest.log
timeout 600 python tools/flame_sweep.py $O/flame_sweep.json > $O/sweep.log 2>&1
echo done > $O/done
