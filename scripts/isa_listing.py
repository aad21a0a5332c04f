"""Development aid: the ISA of the trace kernel with the source line every instruction was booked to
(listing made with `scripts/spec_asm.py scene out.s -gline-tables-only`).  usage: isa_listing.py out.s [first_line last_line]
Prints `file:line  instruction`; with a line range only instructions booked to rt_kernels.hip lines in it (and to
whatever rt_math.hip.h code was inlined while the last rt_kernels.hip line seen was in the range)."""
import re, sys
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 10**9)
files = {}
cur = (0, 0); last_kernel_line = 0
for l in open(sys.argv[1]):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        if files.get(cur[0], "").startswith("rt_kernels"): last_kernel_line = cur[1]
        continue
    m = re.match(r"\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|scratch_[a-z0-9_]+|buffer_[a-z0-9_]+)(.*)", l)
    if re.match(r"^[.A-Za-z_][\w.$]*:", l) and lo <= last_kernel_line <= hi:
        print(l.rstrip()); continue
    if not m: continue
    if lo <= last_kernel_line <= hi:
        print("%-16s %4d  %s%s" % (files.get(cur[0], "?")[:16], cur[1], m.group(1), m.group(2).split(";")[0].rstrip()))
