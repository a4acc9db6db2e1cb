set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ert_fat
rm -rf $OUT; mkdir -p $OUT
EARGS="bench.py --ert --steps 2 --warmup 1 --no-cpu-baseline --no-pe --no-hard-genome"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $EARGS > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $EARGS > $OUT/w.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $OUT/s -- python3 $EARGS > $OUT/s.log 2>&1 || true
python3 - <<'PY'
import csv,glob,collections
for d in ('f','w','s'):
    for f in glob.glob(f'gpurun_out/pmc_ert_fat/{d}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ert_profile_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,sum(v)/len(v),len(v))
PY
