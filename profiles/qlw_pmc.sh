#!/bin/bash
# Run on the GPU box from the repo root: PMC counters of the three kernels of the tridiagonal path.  bash profiles/qlw_pmc.sh 64
R=$(pwd)
SIZES=${1:-"64"}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/qlw_pmc_$name
  QLW_SIZES=$SIZES QLW_TIMING_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/qlw_pmc_$name -- python3 $R/profiles/qlw_probe.py child > /dev/null 2> $R/gpurun_out/qlw_pmc_$name.err
done
cd $R
python3 - <<'PY'
import csv, glob, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/qlw_pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        nm = re.sub(r'\(.*', '', r['Kernel_Name'])
        if 'tridiag' in nm or 'backtransform' in nm or 'replay' in nm:
            agg[nm][r['Counter_Name']].append(float(r['Counter_Value']))
for nm in sorted(agg):
    print(nm)
    for c in sorted(agg[nm]):
        v = agg[nm][c]
        print('   %-28s last %.4g   (n=%d)' % (c, v[-1], len(v)))
PY
