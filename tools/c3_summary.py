"""Fold tools/c3_capture.sh's four rocprofv3 passes into one per-kernel table: calls, average duration at 1 lane and at
8 lanes, HBM traffic per launch (FETCH_SIZE x2 + WRITE_SIZE: MI355X_MICROARCH.md's gfx950 note; both in KB), per kernel
NAME and GRID SIZE (the same kernel runs at several pyramid levels).
    python tools/c3_summary.py gpurun_out/r02_c3 profiles/r02_c3_summary.json"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void elo::(anonymous namespace)::", "").replace("elo::(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def trace(d):
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
            rows[(short(r["Kernel_Name"]), grid)].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return rows


def counters(d, name):
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                rows[(short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0))].append(float(r["Counter_Value"]))
    return rows


def main(src, dst):
    t1, t8 = trace(os.path.join(src, "trace_1lane")), trace(os.path.join(src, "trace_8lanes"))
    fe, wr = counters(os.path.join(src, "fetch_1lane"), "FETCH_SIZE"), counters(os.path.join(src, "write_1lane"), "WRITE_SIZE")
    avg = lambda v: sum(v) / len(v) if v else None
    out = []
    for key in sorted(t1, key=lambda k: -sum(t1[k])):
        f, w = avg(fe.get(key, [])), avg(wr.get(key, []))
        out.append({"kernel": key[0], "grid_threads": key[1], "calls_1lane": len(t1[key]),
                    "avg_us_1lane": round(avg(t1[key]) / 1e3, 2), "total_us_1lane": round(sum(t1[key]) / 1e3, 1),
                    "avg_us_8lanes": round(avg(t8.get(key, [])) / 1e3, 2) if t8.get(key) else None,
                    "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                    "hbm_MB_per_launch_fetch_x2": round((2 * f + w) * 1024 / 1e6, 3) if f is not None and w is not None else None})
    elo = [r for r in out if "Cijk" not in r["kernel"] and "at::" not in r["kernel"]]
    tot = sum(r["total_us_1lane"] for r in out)
    res = {"command": "bench.py --batch 8 --features f16 --no-legs --steps 40 --warmup 16 [--lanes 1|8]",
           "kernel_time_us_total_1lane": round(tot, 1), "kernels": out}
    json.dump(res, open(dst, "w"), indent=1)
    for r in elo[:24]:
        print("%-60s grid %8d  x%-5d %8.2f us  fetch %s KB write %s KB" % (r["kernel"], r["grid_threads"], r["calls_1lane"],
              r["avg_us_1lane"], r["FETCH_SIZE_KB"] and round(r["FETCH_SIZE_KB"], 1), r["WRITE_SIZE_KB"] and round(r["WRITE_SIZE_KB"], 1)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
