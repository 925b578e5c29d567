#!/bin/bash
# Round-5 rocprofv3 evidence (run on the MI355X box):   tools/profile_r05.sh  ->  gpurun_out/r05_*.txt (copy into profiles/)
#   kernel-trace + stats of the default bench (f32) and of its labelled bf16x3 variant, the sequential tracker, and one --pmc pass per
#   counter group (never with a trace domain other than --kernel-trace) over the kernels whose PMC files predate their last rewrite.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r05_bench -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extras --one-stream > $root/gpurun_out/prof_r05_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r05_bench_x3 -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extras --one-stream --conv-bf16x3 > $root/gpurun_out/prof_r05_bench_x3.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r05_conv1d_x3 -o c1d -- python3 $root/tools/bench_conv1d_x3.py > $root/gpurun_out/prof_r05_conv1d_x3.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r05_tracker -o trk -- python3 $root/tools/bench_tracker.py > $root/gpurun_out/prof_r05_tracker.log 2>&1
cd $root
python3 tools/rocprof_summary.py gpurun_out/prof_r05_bench --last-full-step > gpurun_out/r05_bench_kernel_stats_last_step.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r05_bench > gpurun_out/r05_bench_kernel_stats_whole_run.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r05_bench_x3 --last-full-step > gpurun_out/r05_bench_x3_kernel_stats_last_step.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r05_conv1d_x3 > gpurun_out/r05_conv1d_x3_kernel_stats.txt 2>&1
cp gpurun_out/prof_r05_conv1d_x3.log gpurun_out/r05_conv1d_x3_check_and_timing.txt
python3 tools/rocprof_summary.py gpurun_out/prof_r05_tracker --last-full-step > gpurun_out/r05_tracker_kernel_stats_last_frame.txt 2>&1
tail -1 gpurun_out/prof_r05_bench.log | cut -c1-200; tail -1 gpurun_out/prof_r05_bench_x3.log | cut -c1-200
CONV_ONLY=convc2 tools/pmc_kernel.sh gpurun_out/pmc_r05_wino k_conv_wino bench_conv_x3.py > /dev/null 2>&1; cp gpurun_out/pmc_r05_wino/summary.txt gpurun_out/r05_pmc_conv_wino_and_x3.txt
CONV_ONLY=gru tools/pmc_kernel.sh gpurun_out/pmc_r05_wino1d k_conv_wino1d bench_conv_wino.py > /dev/null 2>&1; cp gpurun_out/pmc_r05_wino1d/summary.txt gpurun_out/r05_pmc_conv_wino1d.txt
CONV_N=32 tools/pmc_kernel.sh gpurun_out/pmc_r05_wino1d_x3 k_conv_wino1d bench_conv1d_x3.py > /dev/null 2>&1; cp gpurun_out/pmc_r05_wino1d_x3/summary.txt gpurun_out/r05_pmc_conv_wino1d_and_x3.txt
tools/pmc_kernel.sh gpurun_out/pmc_r05_conv1x1 k_conv1x1 bench_conv1x1_x3.py > /dev/null 2>&1; cp gpurun_out/pmc_r05_conv1x1/summary.txt gpurun_out/r05_pmc_conv1x1.txt
tools/pmc_lookup.sh gpurun_out/pmc_r05_build --only build > /dev/null 2>&1; cp gpurun_out/pmc_r05_build/summary.txt gpurun_out/r05_pmc_corr_build.txt 2>/dev/null
tools/pmc_bench.sh gpurun_out/pmc_r05_bench > /dev/null 2>&1; cp gpurun_out/pmc_r05_bench/summary.txt gpurun_out/r05_pmc_bench_lookup.txt; cp gpurun_out/pmc_r05_bench/pmc_traffic_bench.json gpurun_out/r05_pmc_traffic_bench.json 2>/dev/null
ls -la gpurun_out/r05_*
# only the summaries travel back (gpurun merges at most 64 MiB): drop the raw databases and counter CSVs
rm -rf gpurun_out/prof_r05_bench gpurun_out/prof_r05_bench_x3 gpurun_out/prof_r05_tracker gpurun_out/prof_r05_conv1d_x3 gpurun_out/pmc_r05_*
