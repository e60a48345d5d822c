for i in 1 2 3; do
  echo "base: $(PYTHONPATH=. python tools/order_probe.py split 2>&1 | grep 'split2' | tr '\n' ' ')"
  echo "var:  $(VGH_LIB_PATH=$GRAFT_REPO_ROOT/head_detector_amd/libvgh_var.so PYTHONPATH=. python tools/order_probe.py split 2>&1 | grep 'split2' | tr '\n' ' ')"
done
