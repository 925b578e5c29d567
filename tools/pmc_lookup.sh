#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) over the correlation kernels at bench geometry.
#   tools/pmc_lookup.sh OUTDIR [bench_kernels.py args...]      (run on the MI355X box; results -> OUTDIR/summary.txt)
out=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/$out/$n -o p -- python3 $root/tools/bench_kernels.py --reps 3 "$@" > $root/$out/$n.log 2>&1
done
cd $root
python3 - $out <<'PY' > $out/summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not k.startswith('k_corr') and 'k_corr' not in k[:40]: continue
        agg[(k.split('(')[0][:28], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    print(f'{k:30s} {c:26s} launches={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}')
PY
cat $out/summary.txt
