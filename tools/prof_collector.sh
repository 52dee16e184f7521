#!/bin/bash
# per-kernel averages of the fused collector under rocprofv3: tools/prof_collector.sh [n_envs]  ->  gpurun_out/prof_collector_<n>.csv (top of the kernel stats)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-4096}
mkdir -p "$ROOT/gpurun_out"; cd /tmp && export TMPDIR=/tmp
export FUSED_ONLY=1
rm -rf /tmp/prof_coll
rocprofv3 --kernel-trace --stats -d /tmp/prof_coll -o coll --output-format csv -- python3 "$ROOT/tools/collector_probe.py" $N 128 3 > /tmp/prof_coll.log 2>&1
f=$(find /tmp/prof_coll -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/prof_collector_${N}_kernel_stats.csv"
python3 - "$f" <<'PY' | tee "$ROOT/gpurun_out/prof_collector_$N.txt"
import csv, sys
for i, r in enumerate(csv.reader(open(sys.argv[1]))):
    if i == 0 or i > 9: continue
    print("%-64s calls %6s  avg %9.2f us  %5s %%" % (r[0][:64], r[1], float(r[3]) / 1e3, r[4]))
PY
