#!/bin/bash
# kernel-trace stats only: tools/stats_only.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 1024 --warmup 256 --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1
tail -1 "$OUT/stats.log" | cut -c1-200
head -4 "$OUT/stats/stats_kernel_stats.csv" | cut -c1-60,150-300
