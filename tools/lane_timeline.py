"""Per-hardware-queue timeline of a multi-lane bench run from a rocprofv3 kernel-trace CSV
(`rocprofv3 --kernel-trace -d DIR -- python bench.py --no-legs --steps 400 --lanes 8`): for every queue the span of its last
N forwards, the kernel time inside it and the idle time between kernels; then the per-kernel mean duration over all queues.
python tools/lane_timeline.py trace.csv [forwards per queue to look at, default 40]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = int(sys.argv[2]) if len(sys.argv) > 2 else 40
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
short = lambda n: n.replace("elo::(anonymous namespace)::", "").replace("at::native::", "").replace("void ", "")[:60]
per_kernel = collections.defaultdict(lambda: [0, 0.0])
total_fw, total_span = 0, 0.0
for q, ks in sorted(by_q.items()):
    ks.sort()
    heads = [i for i, k in enumerate(ks) if "setconv_small_kernel<6" in k[2]]        # a forward starts at layer0's set-conv
    if len(heads) < want + 2:
        continue
    a, b = heads[-want - 1], heads[-1]
    span = ks[b][0] - ks[a][0]
    busy = sum(e - s for s, e, _ in ks[a:b])
    gaps = sum(max(0, ks[i + 1][0] - ks[i][1]) for i in range(a, b))
    overlap = sum(max(0, ks[i][1] - ks[i + 1][0]) for i in range(a, b))
    print("queue %s: %d forwards, %.1f us per forward, %d launches per forward, kernel time %.1f us, idle between kernels %.1f us, overlap %.1f us"
          % (q, want, span / want / 1e3, (b - a) // want, busy / want / 1e3, gaps / want / 1e3, overlap / want / 1e3))
    total_fw += want
    total_span = max(total_span, span)
    for s, e, n in ks[a:b]:
        p = per_kernel[short(n)]
        p[0] += 1
        p[1] += e - s
print("all queues: %.0f forwards/s over the window" % (total_fw / (total_span / 1e9)))
for n, (c, tns) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
    print("%7.1f us x %5.2f per forward  %s" % (tns / c / 1e3, c / total_fw, n))
