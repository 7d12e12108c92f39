"""Print the ordered kernel sequence of ONE captured forward from a rocprofv3 kernel-trace CSV of
`bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline` (the last replay in the trace)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a forward starts at the first set-conv small kernel (layer0)
starts = [i for i, n in enumerate(names) if "setconv_small_kernel<6" in n]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n.replace("elo::(anonymous namespace)::", "").replace("at::native::", "")
    print("%8.1f  gap %5.1f  dur %6.1f  grid %-6s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), n[:110]))
    prev_end = e
print("kernels:", b - a, " span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
