#!/bin/bash
# Round-6 rocprofv3 evidence (run on the MI355X box):   tools/profile_r06.sh  ->  gpurun_out/r06_*.txt (copy into profiles/)
#   kernel-trace + stats of the default bench step and of the sequential tracker (launch lists, fused lookup + convc1, one-launch solve);
#   one --pmc pass per counter (never with a trace domain other than --kernel-trace): HBM traffic of bench.py's own lookups and solves, and
#   of the fused lookup + convc1 kernel beside the two kernels it replaces.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r06_bench -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extras --one-stream > $root/gpurun_out/prof_r06_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $root/gpurun_out/prof_r06_tracker -o trk -- python3 $root/tools/bench_tracker.py > $root/gpurun_out/prof_r06_tracker.log 2>&1
cd $root
python3 tools/rocprof_summary.py gpurun_out/prof_r06_bench --last-full-step > gpurun_out/r06_bench_kernel_stats_last_step.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r06_bench > gpurun_out/r06_bench_kernel_stats_whole_run.txt 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_r06_tracker --last-full-step > gpurun_out/r06_tracker_kernel_stats_last_frame.txt 2>&1
tail -1 gpurun_out/prof_r06_bench.log | cut -c1-200; tail -2 gpurun_out/prof_r06_tracker.log
tools/pmc_bench.sh gpurun_out/pmc_r06_bench > /dev/null 2>&1; cp gpurun_out/pmc_r06_bench/summary.txt gpurun_out/r06_pmc_bench_lookup.txt; cp gpurun_out/pmc_r06_bench/pmc_traffic_bench.json gpurun_out/r06_pmc_traffic_bench.json 2>/dev/null
# the fused kernel: FETCH_SIZE / WRITE_SIZE (and the matrix-pipe counters) beside k_corr_lookup + k_conv1x1 / k_conv_igemm on the same inputs
mkdir -p gpurun_out/pmc_r06_fused
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/gpurun_out/pmc_r06_fused/$n -o p -- python3 $root/tools/bench_lookup_conv.py > $root/gpurun_out/pmc_r06_fused/$n.log 2>&1
done
cd $root
python3 - <<'PY' > gpurun_out/r06_pmc_lookup_conv1x1.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_r06_fused/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if any(s in k for s in ('k_lookup_conv1x1', 'k_corr_lookup', 'k_conv1x1', 'k_conv_igemm')):
            agg[(k[:44], r.get('Grid_Size', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
print('# tools/bench_lookup_conv.py (batches 32 16 8 4 2 1 at 640x512) under rocprofv3 --pmc <one group per pass> --kernel-trace; FETCH_SIZE / WRITE_SIZE in KiB')
print('# (read bytes = 2 x FETCH_SIZE x 1024 on gfx950, write bytes = WRITE_SIZE x 1024); grid = work-items: 655360 = 32 pairs of k_lookup_conv1x1, ...')
for (k, g, c), v in sorted(agg.items()):
    print(f'{k:46s} grid={g:>9s} {c:28s} launches={len(v):4d} mean={sum(v)/len(v):.6g}')
PY
cat gpurun_out/r06_pmc_lookup_conv1x1.txt | head -40
ls -la gpurun_out/r06_*
rm -rf gpurun_out/prof_r06_bench gpurun_out/prof_r06_tracker gpurun_out/pmc_r06_*
