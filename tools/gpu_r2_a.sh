#!/bin/bash
# Round-2 GPU call A: correctness of the cross-tile pipelined patch kernels ("q" tiles) + kernel-level and whole-net A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
EXP=$PWD/head_detector_amd/libvgh_exp.so
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv" 2>&1 | tail -25 ) > $O/pytest_conv.log
SHAPES="64,80,80,128,128,3,1 64,160,160,96,96,3,1 64,40,40,256,256,3,1 64,80,80,256,256,3,1 32,80,80,128,128,3,1 64,20,20,512,512,3,1"
CFGS="p16x16x64_n4x1,q16x16x64_n4x1,p16x16x128_n4x2,q16x16x128_n4x2,p8x32x96_n4x1,q8x32x96_n4x1,p16x16x96_n4x1,q16x16x96_n4x1,p8x40x64_n5x1,q8x40x64_n5x1,p8x40x128_n5x2,q8x40x128_n5x2,p16x16x256_n4x4,q16x16x256_n4x4,p8x32x64_n4x2,q8x32x64_n4x2,p8x20x128_n5x2,q8x20x128_n5x2,256x128_w64x64_k1_r3"
for st in 0 3 6 12; do
  echo "#### VGH_STAGGER=$st" >> $O/conv_ab.log
  VGH_LIB_PATH=$EXP VGH_STAGGER=$st timeout 300 python tools/conv_bench.py --shape $SHAPES --cfgs $CFGS --iters 30 >> $O/conv_ab.log 2>&1
done
echo "#### VGH_STAGGER=6 +res" >> $O/conv_ab.log
VGH_LIB_PATH=$EXP VGH_STAGGER=6 timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 64,160,160,96,96,3,1 --cfgs $CFGS --iters 30 --res >> $O/conv_ab.log 2>&1
echo "#### VGH_STAGGER=0 +res" >> $O/conv_ab.log
VGH_LIB_PATH=$EXP VGH_STAGGER=0 timeout 300 python tools/conv_bench.py --shape 64,80,80,128,128,3,1 64,160,160,96,96,3,1 --cfgs $CFGS --iters 30 --res >> $O/conv_ab.log 2>&1
python tools/make_q_table.py $O/conv_cfg_q.json > $O/net_ab.log
for tun in "" "--tuning $O/conv_cfg_q.json"; do
  for share in 1 2; do
    for st in 0 6; do
      VGH_LIB_PATH=$EXP VGH_GRID_SHARE=$share VGH_STAGGER=$st timeout 300 python tools/net_probe.py vgg_heads_l 64 --split 2 $tun >> $O/net_ab.log 2>&1
    done
  done
done
VGH_LIB_PATH=$EXP VGH_GRID_SHARE=1 VGH_STAGGER=0 timeout 300 python tools/net_probe.py vgg_heads_l 64 --split 1 >> $O/net_ab.log 2>&1
VGH_LIB_PATH=$EXP VGH_GRID_SHARE=1 VGH_STAGGER=6 timeout 300 python tools/net_probe.py vgg_heads_l 64 --split 1 --tuning $O/conv_cfg_q.json >> $O/net_ab.log 2>&1
VGH_LIB_PATH=$EXP VGH_GRID_SHARE=2 VGH_STAGGER=6 timeout 300 python tools/net_probe.py vgg_heads_m 32 --split 2 --tuning $O/conv_cfg_q.json >> $O/net_ab.log 2>&1
VGH_LIB_PATH=$EXP VGH_GRID_SHARE=1 VGH_STAGGER=0 timeout 300 python tools/net_probe.py vgg_heads_m 32 --split 2 >> $O/net_ab.log 2>&1
# the product library, default bench line (r01 tile table): baseline of this round's box + first run of the accuracy report
timeout 600 python bench.py --no-cpu-baseline --steps 100 > $O/bench_default.json 2> $O/bench_default.err
echo done > $O/done
