#!/bin/bash
# Every record of profiles/<round>/ from ONE box and one gpurun call: tools/records.sh <round tag, e.g. r03>   (then tools/records_collect.sh <tag> here)
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
G=$ROOT/gpurun_out; mkdir -p "$G"
cd "$ROOT"
python3 -m pytest tests -m gpu -q --tb=line 2>&1 | tail -15 > "$G/${TAG}_gputests.txt"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "$ROOT/tools/micro/two_wave" "$ROOT/tools/micro/two_wave.hip" && "$ROOT/tools/micro/two_wave" > "$G/${TAG}_micro_two_wave.txt" 2>&1      # (always rebuilt from its source: a stale binary must not write a record)
# counters first: bench.py quotes profiles/<tag>/traffic.json and issue.json in its roofline object, so they are made from this box's passes before any bench line is recorded
tools/profile.sh ${TAG}_soft > /dev/null 2>&1
tools/profile.sh ${TAG}_rigid --workload rigid > /dev/null 2>&1
mkdir -p profiles/$TAG
python3 tools/make_traffic.py $G/prof_${TAG}_soft $G/prof_${TAG}_rigid 256 profiles/$TAG/issue.json > profiles/$TAG/traffic.json
cp profiles/$TAG/traffic.json "$G/${TAG}_traffic.json"; cp profiles/$TAG/issue.json "$G/${TAG}_issue.json"
tools/bench_matrix.sh > /dev/null
python3 bench.py > "$G/${TAG}_bench_soft.json" 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$G/${TAG}_bench_driver.json" 2> /dev/null
python3 bench.py --workload rigid --no-cpu-baseline > "$G/${TAG}_bench_rigid.json" 2> /dev/null
python3 bench.py --envs-per-gpu 8192 --randomize --steps 1024 --warmup 256 --no-cpu-baseline > "$G/${TAG}_bench_config5.json" 2> /dev/null
tools/stats_only.sh ${TAG}_config5 --envs-per-gpu 8192 --randomize > /dev/null 2>&1
tools/stats_only.sh ${TAG}_soft_spl1 --steps-per-launch 1 > /dev/null 2>&1
python3 tools/split_timeline.py 200 4096 32 > "$G/${TAG}_timeline.txt" 2>&1
USIM_PROFILE_NSUB=16 python3 tools/split_timeline.py 200 4096 32 > "$G/${TAG}_timeline_multi.txt" 2>&1
USIM_PROFILE_NSUB=16 python3 tools/split_timeline.py 200 8192 64 > "$G/${TAG}_timeline_g8.txt" 2>&1
python3 tests/studies/gpu_parity_fullsize.py > "$G/${TAG}_parity_fullsize.txt" 2>&1
{ python3 tests/studies/parity_report.py 256; python3 tests/studies/parity_report.py 4096 tracking,wrench; } 2>&1 | grep -v amdgpu.ids > "$G/${TAG}_parity_report.txt"
for m in tracking variable_z wrench; do python3 tools/gpu_policy_replay.py $m 2>&1 | grep -v amdgpu.ids; done > "$G/${TAG}_policy_replay.txt"
python3 tools/replay_medians.py 1024 3000 2>&1 | grep -v amdgpu.ids > "$G/${TAG}_replay_medians.txt"
python3 tools/collector_probe.py 4096 128 5 2>&1 | grep -v amdgpu.ids > "$G/${TAG}_collector_probe.txt"
tools/prof_collector.sh 4096 > "$G/${TAG}_prof_collector.txt" 2>&1
python3 tools/ppo_demo.py 60 4096 128 tracking fused 2>&1 | grep -v amdgpu.ids > "$G/${TAG}_ppo_fused.txt"
for m in tracking fixed variable_z wrench; do python3 tools/gpu_soak.py 4096 5000 $m 2>&1 | grep -v amdgpu.ids; done > "$G/${TAG}_soak.txt"
# the full torso (csrc/usim_full.h): bench lines, rocprofv3, replays of the reference checkpoints, full-size parity
python3 bench.py --workload full --steps 40 --warmup 10 > "$G/${TAG}_bench_full.json" 2> /dev/null
python3 bench.py --workload full --steps 20 --warmup 5 --no-cpu-baseline > "$G/${TAG}_bench_full_driver.json" 2> /dev/null
python3 bench.py --workload full --steps 40 --warmup 10 --pgs-iters 12 --no-cpu-baseline > "$G/${TAG}_bench_full_12.json" 2> /dev/null
python3 bench.py --workload full --steps 40 --warmup 10 --envs-per-gpu 8192 --no-cpu-baseline > "$G/${TAG}_bench_full_8192.json" 2> /dev/null
tools/profile_full.sh $TAG > /dev/null 2>&1
for m in tracking variable_z wrench; do python3 tools/gpu_policy_replay.py $m full 2>&1 | grep -v amdgpu.ids; done > "$G/${TAG}_policy_replay_full.txt"
python3 tests/studies/gpu_parity_fullsize.py full > "$G/${TAG}_parity_fullsize_full_torso.txt" 2>&1
cat "$G/${TAG}_gputests.txt" "$G/bench_matrix.txt"; tail -4 "$G/${TAG}_parity_fullsize.txt"
