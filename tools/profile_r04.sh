#!/bin/bash
# Round-4 rocprofv3 evidence (run on the MI355X box): kernel-trace + stats of the default bench, the chunked sequence bench and the
# sequential tracker.   tools/profile_r04.sh  ->  gpurun_out/prof_r04_*/...  + text summaries under gpurun_out/
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r04_bench -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extras --one-stream > $root/gpurun_out/prof_r04_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r04_tracker -o trk -- python3 $root/tools/bench_tracker.py > $root/gpurun_out/prof_r04_tracker.log 2>&1
cd $root
python3 tools/rocprof_summary.py gpurun_out/prof_r04_bench --last-full-step > gpurun_out/r04_bench_kernel_stats_last_step.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r04_bench > gpurun_out/r04_bench_kernel_stats_whole_run.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r04_tracker > gpurun_out/r04_tracker_kernel_stats_whole_run.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r04_tracker --last-full-step > gpurun_out/r04_tracker_kernel_stats_last_frame.txt 2>&1
tail -2 gpurun_out/prof_r04_bench.log | cut -c1-300; tail -3 gpurun_out/prof_r04_tracker.log
head -45 gpurun_out/r04_bench_kernel_stats_last_step.txt
