#!/bin/bash
# copies what tools/records.sh left in gpurun_out/ into profiles/<tag>/ under the names the documents cite
TAG=${1:-r06}; cd "$(dirname "$0")/.."; G=gpurun_out; P=profiles/$TAG; mkdir -p $P
cp $G/bench_matrix.txt $P/bench_matrix.txt
cp $G/${TAG}_bench_soft.json $P/bench_soft.json; cp $G/${TAG}_bench_driver.json $P/bench_soft_driver_style.json
cp $G/${TAG}_bench_rigid.json $P/bench_rigid.json; cp $G/${TAG}_bench_config5.json $P/bench_config5_8192_randomised.json
cp $G/prof_${TAG}_soft/stats/stats_kernel_stats.csv $P/rocprofv3_soft_4096_kernel_stats.csv; cp $G/prof_${TAG}_soft/summary.txt $P/rocprofv3_soft_4096_summary.txt
cp $G/prof_${TAG}_rigid/stats/stats_kernel_stats.csv $P/rocprofv3_rigid_4096_kernel_stats.csv; cp $G/prof_${TAG}_rigid/summary.txt $P/rocprofv3_rigid_4096_summary.txt
cp $G/prof_${TAG}_config5/stats/stats_kernel_stats.csv $P/rocprofv3_config5_8192_kernel_stats.csv
cp $G/prof_${TAG}_soft_spl1/stats/stats_kernel_stats.csv $P/rocprofv3_soft_4096_single_step_launches_kernel_stats.csv
cp $G/${TAG}_timeline.txt $P/timeline_split_soft.txt; cp $G/${TAG}_timeline_multi.txt $P/timeline_split_soft_multi_step.txt; cp $G/${TAG}_timeline_g8.txt $P/timeline_split_soft_8lane_groups_8192.txt
cp $G/${TAG}_parity_fullsize.txt $P/parity_fullsize.txt; cp $G/${TAG}_parity_report.txt $P/parity_report.txt
cp $G/${TAG}_policy_replay.txt $P/policy_replay.txt; cp $G/${TAG}_replay_medians.txt $P/replay_medians.txt; cp $G/${TAG}_collector_probe.txt $P/collector_probe.txt
cp $G/${TAG}_prof_collector.txt $P/rocprofv3_fused_collector_4096_summary.txt; cp $G/${TAG}_ppo_fused.txt $P/ppo_demo_fused_collector.txt; cp $G/${TAG}_soak.txt $P/soak.txt
cp $G/${TAG}_traffic.json $P/traffic.json; cp $G/${TAG}_issue.json $P/issue.json      # made on the box by tools/records.sh, before the bench lines that quote them
ls -la $P
cp $G/${TAG}_gputests.txt $P/gputests.txt; cp $G/${TAG}_micro_two_wave.txt $P/micro_two_wave.txt 2>/dev/null
cp $G/${TAG}_bench_full.json $P/bench_full_torso.json; cp $G/${TAG}_bench_full_driver.json $P/bench_full_torso_driver_style.json; cp $G/${TAG}_bench_full_12.json $P/bench_full_torso_12_sweeps.json; cp $G/${TAG}_bench_full_8192.json $P/bench_full_torso_8192.json
cp $G/prof_full_${TAG}/stats/stats_kernel_stats.csv $P/rocprofv3_full_torso_4096_kernel_stats.csv; grep -v "at::native\|rocclr\|bank_items" $G/prof_full_${TAG}/summary.txt > $P/rocprofv3_full_torso_4096_summary.txt
cp $G/${TAG}_policy_replay_full.txt $P/policy_replay_full_torso.txt; cp $G/${TAG}_parity_fullsize_full_torso.txt $P/parity_fullsize_full_torso.txt
