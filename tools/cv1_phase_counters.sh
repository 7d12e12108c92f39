cd /tmp; export TMPDIR=/tmp
cmd="python /root/repo/tools/roofline_micro.py --kernel cv1 --batch 8 --reps 12"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d /tmp/ph1 -o c -- $cmd > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/ph2 -o t -- $cmd > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('/tmp/ph1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cv1_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v[3:])/len(v[3:])) for k,v in agg.items()})
d=[]
for f in glob.glob('/tmp/ph2/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cv1_kernel' in r['Kernel_Name']: d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print('avg us', sum(d[3:])/len(d[3:])/1e3)
PY
