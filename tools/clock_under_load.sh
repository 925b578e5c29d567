#!/bin/bash
# What shader clock does the board hold under the bench's load?  Samples rocm-smi while bench.py's timed region runs.
#   tools/clock_under_load.sh   (on the MI355X box)  ->  gpurun_out/clock_under_load.txt
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/clock_under_load.txt
mkdir -p $root/gpurun_out
( echo "idle:"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|power" ) > $out
python3 $root/bench.py --steps 60 --warmup 5 --cpu-frames 0 --no-extras --no-live-traffic > $root/gpurun_out/clock_bench.json 2> /dev/null &
pid=$!
sleep 14                                    # import + model set-up + warm-up
for i in 1 2 3 4 5 6; do
  ( echo "under load, sample $i:"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|power" ) >> $out
  sleep 0.5
done
wait $pid
python3 -c "import json; d=json.loads(open('$root/gpurun_out/clock_bench.json').read()); print('bench:', d['value'], 'solves/s', d['ms_per_step'], 'ms/step')" >> $out
cat $out
