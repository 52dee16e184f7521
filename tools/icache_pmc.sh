#!/bin/bash
# instruction-fetch counters of the step kernel (run on the GPU box): is the 80 KB kernel body bound by the shared 64 KB I-cache?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
grep -i -E "ICACHE|IFETCH|INST_CACHE|SQC_" "$OUT/counters.txt" | head -60 > "$OUT/icache_counters.txt"
ARGS="--steps 300 --warmup 50 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d "$OUT/pmc_ic" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_ic.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d "$OUT/pmc_if" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_if.log" 2>&1
python3 - "$OUT" <<'PY'
import sys,glob,csv,collections
out=sys.argv[1]
for sub in ("pmc_ic","pmc_if"):
    acc=collections.defaultdict(lambda: [0.0,0])
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "usim_step_kernel" in r["Kernel_Name"] and "Li0EEE" in r["Kernel_Name"].replace(" ",""):
                a=acc[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for k,(v,n) in sorted(acc.items()): print(sub, k, "per launch", v/max(n,1), "launches", n)
PY
tail -3 "$OUT/pmc_ic.log" | cut -c1-300
