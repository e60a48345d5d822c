#!/bin/bash
# A/B of the product library against a variant build (python -m head_detector_amd.build -DNAME=v -> libvgh_var.so): alternates the two
# libraries on one box so that box-to-box and run-to-run drift cancels.   tools/ab_var.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in $(seq 1 ${1:-3}); do
  echo "base: $(PYTHONPATH=$ROOT python tools/order_probe.py split 2>&1 | grep 'split2' | tr '\n' ' ')"
  echo "var:  $(VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_var.so PYTHONPATH=$ROOT python tools/order_probe.py split 2>&1 | grep 'split2' | tr '\n' ' ')"
done
