"""Print the ordered kernel sequence of ONE captured forward from a rocprofv3 kernel-trace CSV of
`bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --pool 1` (the last replay in the trace):
start offset, idle gap since the previous kernel ended, duration, grid, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "setconv_small_kernel<6" in n]      # a forward starts at layer0's set-conv
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end, busy, gaps = t0, 0, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("elo::(anonymous namespace)::", "").replace("at::native::", "").replace("void ", "")
    print("%8.1f  gap %5.1f  dur %6.1f  wg %-6s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
          int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else "", n[:90]))
    busy += e - s
    gaps += max(0, s - prev_end)
    prev_end = e
print("kernels: %d  span %.1f us  busy %.1f us  idle between kernels %.1f us" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, gaps / 1e3))
