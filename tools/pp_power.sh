#!/bin/bash
# Watts / clock / time per PHASE of a conv tile: the experiments build's ablation switches under the rocm-smi sampler.
# For every VGH_CONV_ABLATE value a conv_bench hot loop of ~SECS seconds runs back to back while socket power and shader clock are sampled every 0.3 s.
#   gpurun -- 'tools/pp_power.sh r4k'      (CFG / SHAPE / ABLATES / SECS to vary; needs head_detector_amd/libvgh_exp.so = build --experiments)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export PYTHONPATH=$ROOT TMPDIR=/tmp
O=$ROOT/gpurun_out/$1; mkdir -p $O
SH=${SHAPE:-64,80,80,128,128,3,1}
OUT=$O/${TAG:-r04}_power_per_phase.txt
{
echo "# tools/pp_power.sh: conv_bench --shape $SH hot loops (${SECS:-8} s each, ~0.3 s rocm-smi samples after a ${WARM:-3} s warm-up), experiments build; one MI355X (cap 1400 W)"
echo "# VGH_CONV_ABLATE bits for the g / h tiles: 1 no LDS-DMA, 2 no MFMA, 8 no epilogue, 32 no fragment reads (MFMAs on stale registers), 128 epilogue without its stores"
echo "# for the p tiles: 1 no tile loads, 2 no compute (MFMA + fragment reads), 8 no epilogue, 32 MFMAs without their fragment reads"
} > $OUT
for cfg in ${CFGS:-g8x8x128_n8 p16x16x64_n4x1}; do
  for ab in ${ABLATES:-0 1 32 33 2 8 41}; do
    # calibrate the iteration count to ~SECS seconds
    ms=$(VGH_EXPERIMENTS=1 VGH_CONV_ABLATE=$ab VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_exp.so python tools/conv_bench.py --shape $SH --cfgs $cfg --iters 200 2>/dev/null | grep "cfg " | awk '{print $4}')
    [ -z "$ms" ] && continue
    iters=$(python -c "print(int(${SECS:-8} * 1000 / $ms))")
    : > $O/smi.txt
    VGH_EXPERIMENTS=1 VGH_CONV_ABLATE=$ab VGH_LIB_PATH=$ROOT/head_detector_amd/libvgh_exp.so python tools/conv_bench.py --shape $SH --cfgs $cfg --iters $iters > $O/run.log 2>&1 &
    pid=$!
    sleep ${WARM:-3}
    while kill -0 $pid 2>/dev/null; do
      rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' ' >> $O/smi.txt; echo >> $O/smi.txt
      sleep 0.3
    done
    wait $pid
    python - "$cfg" "$ab" $O/smi.txt $O/run.log >> $OUT <<'PY'
import re, sys, statistics
cfg, ab, smi, run = sys.argv[1:]
pw, ck = [], []
for ln in open(smi):
    # rocm-smi --showpower --showclocks --csv: "... card0,(fclk),lvl,(mclk),lvl,(sclk),lvl,(socclk),lvl,<socket power W>"
    m = re.search(r"card\d+,\((\d+)Mhz\),[^,]*,\((\d+)Mhz\),[^,]*,\((\d+)Mhz\),[^,]*,\((\d+)Mhz\),[^,]*,(\d+(?:\.\d+)?)", ln)
    if m:
        ck.append(int(m.group(3)))
        pw.append(float(m.group(5)))
line = [l for l in open(run) if l.startswith("cfg ")]
t = line[0].split() if line else ["?"] * 6
med = lambda v: statistics.median(v) if v else float("nan")
print(f"{cfg:18s} ablate={ab:>3s}  {t[3]:>8s} ms  {t[5]:>8s} TFLOP/s(nominal)   power W median {med(pw):7.1f} max {max(pw) if pw else float('nan'):7.1f}   sclk MHz median {med(ck):6.0f}   ({len(pw)} samples)")
PY
    tail -1 $OUT
  done
done
