#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) over any tool in tools/, filtered by kernel-name substring.
#   tools/pmc_kernel.sh OUTDIR KERNEL_SUBSTRING TOOL.py [args...]      (run on the MI355X box; results -> OUTDIR/summary.txt)
out=$1; filt=$2; tool=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_MFMA"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $root/$out/$n -o p -- python3 $root/tools/$tool "$@" > $root/$out/$n.log 2>&1
done
cd $root
python3 - $out "$filt" <<'PY' > $out/summary.txt
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sys.argv[2] not in k: continue
        agg[(k.split('(')[0][:40], r.get('Grid_Size', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, g, c), v in sorted(agg.items()):
    print(f'{k:42s} grid={g:>9s} {c:28s} launches={len(v):3d} mean={sum(v)/len(v):.6g}')
PY
cat $out/summary.txt
