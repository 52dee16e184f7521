#!/bin/bash
# rocprofv3 runs of the FULL-torso workload (bench.py --workload full) for profiles/: kernel-trace stats + two PMC passes (run on the GPU box via gpurun)
# usage: tools/profile_full.sh <tag>
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_full_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload full --steps 40 --warmup 10 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/pmc_sq1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/pmc_sq2" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
tail -1 "$OUT/stats.log" | cut -c1-300
