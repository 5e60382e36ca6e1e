set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_E; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench_configs.py E --reps 2 > /dev/null 2> $OUT/pmc_$name.err
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_E/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_solve_wave' in r['Kernel_Name']:
            acc['wave'][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc['wave'].items(): print(k, '%.4g'%(sum(v)/len(v)), len(v))
PY
