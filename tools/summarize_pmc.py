"""Per-kernel summary of a tools/collect_sq_counters.sh run: average launch time (rocprofv3 --kernel-trace --stats) and
counter values per dispatch of libwol's kernels (names starting with k_).
    python tools/summarize_pmc.py <dir> <out.json> ['<json object stored under "_config">']"""
import collections, csv, glob, json, os, sys

csv.field_size_limit(1 << 30)


def short(name):
    n = name.strip('"')
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0]


def main():
    d, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    meta = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn, newline="")):
            k = short(row["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            disp[k][c].add((fn, row["Dispatch_Id"]))
            meta[k] = {"workgroup": int(row["Workgroup_Size"]), "grid": int(row["Grid_Size"]), "lds_bytes": int(row["LDS_Block_Size"]),
                       "vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"])}
    stats = {}
    for fn in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(fn, newline="")):
            k = short(row["Name"])
            if k.startswith("k_"):
                stats[k] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3, "total_ms": float(row["TotalDurationNs"]) / 1e6,
                            "percent": float(row["Percentage"])}
    res = {}
    for k in sorted(set(acc) | set(stats), key=lambda k: -stats.get(k, {}).get("total_ms", 0.0)):
        e = dict(meta.get(k, {}))
        e.update(stats.get(k, {}))
        e["counters_per_dispatch"] = {c: acc[k][c] / max(len(disp[k][c]), 1) for c in sorted(acc[k])}
        res[k] = e
    if len(sys.argv) > 3:
        res["_config"] = json.loads(sys.argv[3])
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res.items():
        if k.startswith("_"):
            continue
        print("%-34s %6s calls  avg %9.1f us  %5.1f %%   wg %4s vgpr %4s lds %6s" % (k[:34], e.get("calls", "-"), e.get("avg_us", 0.0),
              e.get("percent", 0.0), e.get("workgroup", "-"), e.get("vgpr", "-"), e.get("lds_bytes", "-")))
        for c, v in e["counters_per_dispatch"].items():
            print("      %-24s %.6g" % (c, v))


if __name__ == "__main__":
    main()
