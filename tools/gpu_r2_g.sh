#!/bin/bash
# re-tune the tile tables with the pipelined "q" tiles among the candidates, then measure the whole network
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg_before.json
timeout 900 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --reps 4 --report $O/tune_m32.json > $O/tune.log 2>&1
timeout 900 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --reps 4 --report $O/tune_l64.json >> $O/tune.log 2>&1
cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg_after.json
for i in 1 2; do
timeout 300 python tools/net_probe.py vgg_heads_l 64 --split 2 --tuning $O/conv_cfg_before.json >> $O/net.log 2>&1
timeout 300 python tools/net_probe.py vgg_heads_l 64 --split 2 --tuning $O/conv_cfg_after.json >> $O/net.log 2>&1
timeout 300 python tools/net_probe.py vgg_heads_m 32 --split 2 --tuning $O/conv_cfg_before.json >> $O/net.log 2>&1
timeout 300 python tools/net_probe.py vgg_heads_m 32 --split 2 --tuning $O/conv_cfg_after.json >> $O/net.log 2>&1
done
echo done > $O/done
