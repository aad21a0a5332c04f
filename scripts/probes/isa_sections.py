"""Development aid: static instruction counts of the trace kernel per source section, from an ISA listing made
with `scripts/spec_asm.py scene out.s -gline-tables-only`.  Instructions inlined from rt_math.hip.h are booked
to the section of the last rt_kernels.hip line seen before them.  usage: isa_sections.py out.s"""
import re, sys, collections
src = open("ray_tracing_amd/csrc/rt_kernels.hip").read().split("\n")
def line_of(pat):
    for i, l in enumerate(src):
        if pat in l: return i + 1
    raise KeyError(pat)
marks = [("in-order sum", line_of("auto add_finished_samples = [&]()")), ("supply", line_of("---- 1. sample supply")), ("shade", line_of("---- 2. shade the pending")),
         ("push", line_of("---- 3. the shadow taps go")), ("mains", line_of("---- 4. the bounce rays")), ("back", line_of("---- 5. back: retire")),
         ("sum call", line_of("---- 6. add the finished samples"))]
body = line_of("RT_DEV void wavefront_body"); body_end = marks[-1][1] + 80
fn = {"prepare_ray": line_of("RT_DEV RayPrep prepare_ray"), "box_entry_fast": line_of("RT_DEV bool box_entry_fast"), "sqrt64": line_of("RT_DEV double sqrt_of_float64"),
      "ball_entry_fast": line_of("RT_DEV bool ball_entry_fast"), "nearest_hit_fast": line_of("RT_DEV Hit nearest_hit_fast"),
      "nearest_hit_spec": line_of("RT_DEV Hit nearest_hit_spec"), "sky_lookup": line_of("RT_DEV V3 sky_lookup"), "global_row": line_of("RT_DEV int global_row")}
fn_sorted = sorted(fn.items(), key=lambda kv: kv[1])
def where(line):
    if line >= body:
        sec = "prologue"
        for name, l in marks:
            if line >= l: sec = name
        return sec
    name = None
    for n, l in fn_sorted:
        if line >= l: name = n
    return name
cnt = collections.defaultdict(collections.Counter)
sec = "prologue"; sub = None
for l in open(sys.argv[1]):
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        f, ln = int(m.group(1)), int(m.group(2))
        if f in (0, 1):
            w = where(ln)
            if ln >= body: sec, sub = w, None
            else: sub = w
        continue
    m = re.match(r"\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)", l)
    if not m: continue
    op = m.group(1)
    kind = "valu" if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane")) else ("lane" if op.startswith("v_") else ("nop" if op.startswith("s_nop") else ("salu" if op.startswith("s_") else "mem")))
    cnt[(sec, sub)][kind] += 1
tot = collections.Counter()
for k in sorted(cnt, key=lambda k: (k[0], str(k[1]))):
    c = cnt[k]; tot.update(c)
    print("%-10s %-18s valu %5d  lane-spill %4d  salu %5d  nop %4d  mem %4d" % (k[0], k[1] or "", c["valu"], c["lane"], c["salu"], c["nop"], c["mem"]))
print("total", dict(tot))
