"""Summarise rocprofv3 CSV output of scripts/profile_round.sh into one small text file."""
import collections, csv, glob, os, sys
root = sys.argv[1]
lines = []
for f in sorted(glob.glob(os.path.join(root, "stats", "*", "*_kernel_stats.csv"))):
    lines.append(f"== kernel stats ({os.path.basename(f)})")
    lines += [l.rstrip() for l in open(f)]
for d in ("pmc_valu", "pmc_fetch", "pmc_write"):
    for f in sorted(glob.glob(os.path.join(root, d, "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rt_trace" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines.append(f"== {d}: mean per launch of rt_trace_* over {max(map(len, agg.values()), default=0)} launches (C1, 1920x1080x64spp)")
        for k in sorted(agg):
            lines.append(f"{k:28s} {sum(agg[k]) / len(agg[k]):.6g}")
        if "SQ_THREAD_CYCLES_VALU" in agg:
            u = sum(agg["SQ_THREAD_CYCLES_VALU"]) / sum(agg["SQ_ACTIVE_INST_VALU"]) / 64
            lines.append(f"{'VALU lane utilisation':28s} {u:.4f}   (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64)")
        if "FETCH_SIZE" in agg:
            kb = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
            lines.append(f"{'HBM read bytes (FETCH_SIZE KiB x1024 x2, gfx950 correction)':28s} {kb * 1024 * 2:.6g}")
        if "WRITE_SIZE" in agg:
            kb = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
            lines.append(f"{'HBM write bytes (WRITE_SIZE KiB x1024)':28s} {kb * 1024:.6g}")
print("\n".join(lines))
