set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06/dense_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cmd="python $REPO/tools/dense_rows_pmc.py ${ELO_PMC_SHAPE:-172800 128 128 dx}"
KERNEL=${ELO_PMC_KERNEL:-dense_rows}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o a -- $cmd >> $OUT/log.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/b -o b -- $cmd >> $OUT/log.txt 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/c -o c -- $cmd >> $OUT/log.txt 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/e -o e -- $cmd >> $OUT/log.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/d -o d -- $cmd >> $OUT/log.txt 2>&1
cd $REPO
KERNEL=$KERNEL python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob("gpurun_out/r06/dense_pmc/*/*counter_collection.csv") + glob.glob("gpurun_out/r06/dense_pmc/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if os.environ.get("KERNEL", "dense_rows") in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, len(v), sum(v) / len(v))
PY
