#!/bin/bash
# HBM traffic of k_corr_lookup measured on bench.py's OWN launches: rocprofv3 PMC passes (one counter group per pass, --kernel-trace
# only, the program itself after `--`) over `python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-extras`.
#   tools/pmc_bench.sh OUTDIR      (run on the MI355X box; -> OUTDIR/summary.txt, OUTDIR/pmc_traffic_bench.json)
# gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section; calibrated in round 1 on k_pose_reduce's known 220.2 MB):
# read bytes = 2 x FETCH_SIZE KiB (128-B requests are tallied at 64 B), write bytes = WRITE_SIZE KiB x 1024.
out=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/$out/$n -o p -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-extras > $root/$out/$n.log 2>&1
done
cd $root
python3 - $out <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for name in ('k_corr_lookup', 'k_pose_reduce', 'k_corr_build'):
            if name in k:
                agg[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for (k, c), v in sorted(agg.items()):
        # the lookup's launches over the GRU iterations: report all and the steady half (later iterations = what bench.py times too)
        print(f'{k:16s} {c:26s} launches={len(v):4d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}', file=fh)
lk_f, lk_w = agg.get(('k_corr_lookup', 'FETCH_SIZE')), agg.get(('k_corr_lookup', 'WRITE_SIZE'))
if lk_f and lk_w:
    rd, wr = 2 * 1024 * sum(lk_f) / len(lk_f), 1024 * sum(lk_w) / len(lk_w)
    rq = agg.get(('k_corr_lookup', 'TCC_EA0_RDREQ_sum'))
    d = {'k_corr_lookup_bench': {
        'workload': 'bench.py default workload (16 frame pairs = RAFT batch 32, 640x512, 12 GRU iterations): every k_corr_lookup launch of the run',
        'launches': len(lk_f), 'fetch_size_kib': sum(lk_f) / len(lk_f), 'write_size_kib': sum(lk_w) / len(lk_w),
        'read_bytes': rd, 'write_bytes': wr, 'traffic_bytes_per_launch': rd + wr,
        'read_requests_128B': (sum(rq) / len(rq)) if rq else None,
        'correction': 'read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-B requests at 64 B), write bytes = WRITE_SIZE KiB x 1024',
        'source': 'tools/pmc_bench.sh: rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-extras'}}
    json.dump(d, open(out + '/pmc_traffic_bench.json', 'w'), indent=1)
    print(json.dumps(d))
print(open(out + '/summary.txt').read())
PY
