#!/bin/bash
# Socket power and shader clock while the WHOLE two-lane forward runs back to back (is the forward as a whole at the 1400 W cap?)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export PYTHONPATH=$ROOT
O=$ROOT/gpurun_out/power_net.txt
: > $O
for cfg in "vgg_heads_l 64" "vgg_heads_m 32"; do
  python tools/net_probe.py $cfg --steps ${STEPS:-1500} > gpurun_out/power_net_run.log 2>&1 &
  pid=$!
  sleep ${WARM:-14}
  while kill -0 $pid 2>/dev/null; do
    echo "$(echo $cfg | tr ' ' _) $(rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0)" >> $O
    sleep 0.25
  done
  wait $pid
  grep -v amdgpu gpurun_out/power_net_run.log | tail -1 >> $O
done
python - <<'PY'
import re, collections, os
rows = collections.defaultdict(list)
for ln in open("gpurun_out/power_net.txt"):
    m = re.match(r"(\S+) card0,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\d+,\((\d+)Mhz\),\S+,\((\d+)Mhz\),\S+,([\d.]+)", ln)
    if m:
        rows[m.group(1)].append((int(m.group(4)), float(m.group(6))))
    elif "ms/forward" in ln:
        print(ln.strip())
for k, v in rows.items():
    s = sorted(x[0] for x in v)
    p = sorted(x[1] for x in v)
    print(f"{k}: {len(v)} samples  sclk MHz min/median/max {s[0]}/{s[len(s) // 2]}/{s[-1]}  socket power W min/median/max {p[0]:.0f}/{p[len(p) // 2]:.0f}/{p[-1]:.0f}  mean {sum(p) / len(p):.0f}")
PY
