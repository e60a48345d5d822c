#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
( timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 ) > $O/pytest_gpu.log
echo done > $O/done
