#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks --showtemp 2>&1 | grep -v "^$" > $O/smi_idle.log
( timeout 120 python tools/net_probe.py vgg_heads_l 64 --split 2 --steps 600 > $O/net.log 2>&1 ) &
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|power\|mclk" >> $O/smi_load.log
  echo "--" >> $O/smi_load.log
  sleep 0.5
done
wait
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -20 > $O/dpm.log
rocm-smi --showperflevel --showpowerprofile 2>&1 | head -30 >> $O/dpm.log
echo done > $O/done
