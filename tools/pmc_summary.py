"""Fold the rocprofv3 captures of tools/roofline_micro.py (kernel trace + FETCH_SIZE + WRITE_SIZE passes, one
directory set per <kernel>_b<batch> tag) into one JSON: duration, algorithmic GB/s or TFLOP/s, HBM traffic.
    python tools/pmc_summary.py gpurun_out/pmc2 profiles/r01_pmc/summary.json
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(MI355X_MICROARCH.md, HBM section), so both the raw and the fetch-doubled totals are kept."""
import csv, glob, json, os, sys

KNAME = {"cv1": "cv1_kernel", "pool": "softmax_pool", "pool2": "softmax_pool", "encode1": "cv_encode1", "encode2": "cv_encode2",
         "select32": "group_select_k", "select32_l2": "group_select_k", "random16": "group_random_k"}


def main(src, dst):
    res = []
    for info in sorted(glob.glob(os.path.join(src, "*.info"))):
        tag = os.path.basename(info)[:-5]
        meta = json.load(open(info))
        kname = KNAME[meta["kernel"]]
        meta.setdefault("half", False)
        rows = list(csv.DictReader(open(os.path.join(src, tag + ".trace", "t_kernel_trace.csv"))))
        if kname == "cv1_kernel" and not any(kname in r["Kernel_Name"] for r in rows):
            kname = "cv1_rr_kernel"            # from 24.6 k rows on stage 1 is a select-k launch + the register-resident kernel
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if kname in r["Kernel_Name"]][5:]
        avg = sum(durs) / len(durs)

        def counter(sub, f, name):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, tag + sub, f)))
                    if kname in r["Kernel_Name"] and r["Counter_Name"] == name][5:]
            return sum(vals) / len(vals) if vals else None
        fetch = counter(".fetch", "f_counter_collection.csv", "FETCH_SIZE")
        write = counter(".write", "w_counter_collection.csv", "WRITE_SIZE")
        ab = meta["algorithmic_bytes"]
        row = dict(tag=tag, kernel=kname, batch=meta["batch"], avg_us=round(avg / 1e3, 2), algorithmic_MB=round(ab / 1e6, 2),
                   algorithmic_GBps=round(ab / avg, 1), frac_of_8TBps=round(ab / avg / 8000, 4), FETCH_SIZE_KB=fetch, WRITE_SIZE_KB=write,
                   hbm_MB_raw=round((fetch + write) * 1024 / 1e6, 2), hbm_MB_fetch_x2=round((2 * fetch + write) * 1024 / 1e6, 2))
        if "flops" in meta:
            row["TFLOPs"] = round(meta["flops"] / avg / 1e3, 2)
            # algorithmic (fp32-class) flops against what NATIVE fp32 MFMA could deliver at best (157.3 TFLOP/s): a ratio, not a
            # utilisation -- the kernel issues fp16 MFMAs (3 products per algorithmic one) and may exceed 1; the utilisation
            # figure is bench.py's mfma.frac (executed flops against the 2.5 PFLOP/s fp16 peak)
            row["algorithmic_TFLOPs_over_native_fp32_mfma_peak"] = round(meta["flops"] / avg / 1e3 / 157.3, 4)
        res.append(row)
        print(row)
    # keyed the way bench.py's `traffic` lookup wants it: "<kernel>/b<batch>/<f32|f16>" -> HBM bytes per launch (FETCH_SIZE x2)
    table = {"%s/b%d/%s" % (r["kernel"], r["batch"], "f16" if r["tag"].split("_b")[0].endswith("_f16") else "f32"):
             {"traffic_bytes": int(r["hbm_MB_fetch_x2"] * 1e6), "avg_us": r["avg_us"], "tag": r["tag"]} for r in res}
    table["rows"] = res
    json.dump(table, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
