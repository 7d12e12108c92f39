#!/bin/bash
# Cumulative cost of cv1_kernel's phases at batch B: builds that stop after the grouping (1), after the gather + geometry
# encode (2), after the six dense layers (3), and the whole kernel; kernel time from --kernel-trace and instruction / LDS
# bank-conflict counters from a separate --pmc pass.
#   usage (GPU box, repo root):  bash tools/cv1_phases.sh [batch]
set -u
B=${1:-8}; REPO=$(pwd)
XSRC=$(dirname $(bash $REPO/tools/micro/experiment_source.sh))     # csrc/ with the experiment switches patched in (ELO_CV1_STOP)
cd /tmp; export TMPDIR=/tmp
for stop in 1 2 3 0; do
    def=""; [ $stop != 0 ] && def="-DELO_CV1_STOP=$stop"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $def -I$REPO/include \
        $XSRC/*.hip $XSRC/*.cpp -o /tmp/libelo_stop$stop.so &
done
wait
for stop in 1 2 3 0; do
    export ELO_LIB_PATH=/tmp/libelo_stop$stop.so
    cmd="python $REPO/tools/roofline_micro.py --kernel cv1 --batch $B --reps 12"
    rm -rf /tmp/ph1 /tmp/ph2
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d /tmp/ph1 -o c -- $cmd > /dev/null 2>&1
    rocprofv3 --kernel-trace --output-format csv -d /tmp/ph2 -o t -- $cmd > /dev/null 2>&1
    python - $stop <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/ph1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cv1_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
c = {k: sum(v[3:]) / len(v[3:]) for k, v in agg.items()}
w = c.get('SQ_WAVES', 1)
d = []
for f in glob.glob('/tmp/ph2/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cv1_kernel' in r['Kernel_Name']: d.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
name = {'1': 'through grouping', '2': 'through gather+encode', '3': 'through the dense layers', '0': 'whole kernel'}[sys.argv[1]]
print("%-26s %7.1f us | per wave: VALU %5.0f (MFMA %4.0f) SALU %5.0f LDS %4.0f VMEM %4.0f | LDS bank-conflict cycles %9.0f" % (
    name, sum(d[3:]) / len(d[3:]) / 1e3, c['SQ_INSTS_VALU'] / w, c['SQ_INSTS_MFMA'] / w, c['SQ_INSTS_SALU'] / w,
    c['SQ_INSTS_LDS'] / w, c['SQ_INSTS_VMEM'] / w, c['SQ_LDS_BANK_CONFLICT']))
PY
done
