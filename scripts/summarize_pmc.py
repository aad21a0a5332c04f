"""Summarise rocprofv3 CSV output of scripts/profile_round.sh into one small text file (stdout) and, with a
third argument, merge the traffic figures into profiles/pmc_latest.json (what bench.py reports as
roofline.traffic).  usage: summarize_pmc.py gpurun_out/<tag> <config> [profiles/pmc_latest.json]"""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "C1"
lines = []
for sub, what in (("stats", "bench.py with frames in flight (launches overlap; rocprofv3 serialises what it can)"),
                  ("stats_serial", "bench.py --depth 1: one frame in flight, launches never overlap")):
    for f in sorted(glob.glob(os.path.join(root, sub, "**", "*_kernel_stats.csv"), recursive=True)):
        lines.append(f"== kernel stats, {what} ({os.path.basename(f)})")
        lines += [l.rstrip() for l in open(f)]
    j = os.path.join(root, "bench_under_rocprof.json" if sub == "stats" else "bench_under_rocprof_serial.json")
    if os.path.exists(j):
        try:
            d = json.loads(open(j).read().strip().splitlines()[-1])
            r = d.get("roofline", {})
            lines.append(f"   the bench line of that same process: ms_per_step {d.get('ms_per_step')}, roofline.avg_kernel_ms {r.get('avg_kernel_ms')} "
                         f"(per-launch events {r.get('avg_kernel_ms_per_launch_events')}, span {r.get('avg_kernel_ms_span')}), frac {r.get('frac')}")
        except Exception as e:
            lines.append(f"   ({j}: {e})")
totals = {}
KERNELS = ("rt_trace", "rt_primary_pass")
for d in ("pmc_valu", "pmc_busy", "pmc_mix1", "pmc_mix2", "pmc_fetch", "pmc_write"):
    for f in sorted(glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv"), recursive=True)):
        per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in KERNELS:
            agg = per_kernel.get(k)
            if not agg:
                continue
            lines.append(f"== {d}: mean per launch of {k}* over {max(map(len, agg.values()), default=0)} launches ({config})")
            for c in sorted(agg):
                mean = sum(agg[c]) / len(agg[c])
                lines.append(f"{c:28s} {mean:.6g}")
                totals.setdefault(k, {})[c] = mean
            if "SQ_THREAD_CYCLES_VALU" in agg:
                u = sum(agg["SQ_THREAD_CYCLES_VALU"]) / sum(agg["SQ_ACTIVE_INST_VALU"]) / 64
                lines.append(f"{'VALU lane utilisation':28s} {u:.4f}   (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64)")
            if "SQ_ACTIVE_INST_VALU" in agg and "SQ_WAVE_CYCLES" in agg:
                lines.append(f"{'VALU-active share of wave cycles':28s} {sum(agg['SQ_ACTIVE_INST_VALU']) / sum(agg['SQ_WAVE_CYCLES']):.4f}   (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both in quad-cycles)")
            if "FETCH_SIZE" in agg:
                kb = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
                lines.append(f"{'L2->fabric read bytes':28s} {kb * 1024:.6g}   (FETCH_SIZE KiB x 1024; x2 for wide streaming reads on gfx950: {kb * 2048:.6g})")
            if "WRITE_SIZE" in agg:
                kb = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
                lines.append(f"{'L2->fabric write bytes':28s} {kb * 1024:.6g}   (WRITE_SIZE KiB x 1024)")
print("\n".join(lines))
if len(sys.argv) > 3:
    path = sys.argv[3]
    table = json.load(open(path)) if os.path.exists(path) else {}
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in totals.values()) * 1024
    write = sum(v.get("WRITE_SIZE", 0.0) for v in totals.values()) * 1024
    valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in totals.values())
    lanes = totals.get("rt_trace", {}).get("VALUUtilization")
    # per kernel: every counter's mean per launch (bench.py prices the two kernels apart: they are bound differently), with the
    # traffic in bytes; `source` names the TRACKED summary these numbers are printed in, beside this json
    kernels = {}
    for k, c in totals.items():
        e = dict(c)
        if "FETCH_SIZE" in c: e["fetch_bytes"] = c["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in c: e["write_bytes"] = c["WRITE_SIZE"] * 1024
        kernels[k] = e
    tag = os.environ.get("PMC_SOURCE", root)
    table[config] = {"kernel": "rt_primary_pass + rt_trace_*", "fetch_bytes": fetch, "write_bytes": write,
                     "valu_instructions": valu, "valu_lane_utilisation_pct": lanes, "kernels": kernels,
                     "source": f"rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in KiB x 1024, separate passes; means per launch and kernel), {tag}; "
                               "reads are scattered 4-byte skybox gathers served by the Infinity Cache, so the x2 streaming correction is not applied"}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
