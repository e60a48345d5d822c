#!/bin/bash
# retune both benchmark buckets with the new high-occupancy tiles among the candidates; whole-net time before / after on the same box
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
probe() { python - <<'PY'
import torch, sys
from head_detector_amd.engine import VGHeadsEngine
dev=torch.device("cuda",0)
for v,b in (("vgg_heads_m",32),("vgg_heads_l",64)):
    eng=VGHeadsEngine(v,image_size=640,max_batch=b,seed=1); eng.set_split(2)
    x=torch.randint(0,256,(b,640,640,3),dtype=torch.uint8).to(dev)
    for _ in range(5): eng.forward_net(x)
    torch.cuda.synchronize()
    best=1e9
    for rep in range(3):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        for _ in range(40): eng.forward_net(x)
        e1.record(eng.stream); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/40)
    print(f"{v} b{b} two lanes: {best:.3f} ms", flush=True)
    eng.close()
PY
}
echo "--- before"; probe 2>&1 | grep -v amdgpu
timeout 900 python tools/tune_conv.py --variant vgg_heads_l --batch 64 --report $O/tune_l64.json > $O/tune.log 2>&1
timeout 900 python tools/tune_conv.py --variant vgg_heads_m --batch 32 --report $O/tune_m32.json >> $O/tune.log 2>&1
tail -3 $O/tune.log
cp head_detector_amd/tuning/conv_cfg.json $O/conv_cfg.json
echo "--- after"; probe 2>&1 | grep -v amdgpu
