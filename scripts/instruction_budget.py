"""The instruction budget of the compiled trace kernel: VALU lane-instructions... per sample, by section of a round and by issue
class, STATIC count x MEASURED executions.

  static   the kernel as the library's build compiles it (csrc/compile_scene.py, the embedded kernels' compiler) with
           -gline-tables-only added, disassembled; every instruction is booked -- through its inline stack (llvm-symbolizer) -- to
           a block of the source: a section of wavefront_body's round, the tap trace, a sphere's fp64 roots, a fallback path ...
  dynamic  how often a wave ran each block: the per-site execution counters of the instrumented build (-DRT_STATS through
           rt_tuning.jit_flags; csrc/rt_stats.hip.h) on the same frame -- `stats` mode below, on the GPU.

usage:  instruction_budget.py stats <scene 0|1> <out.json>        (GPU) the site counters of C1 / C2, as JSON
        instruction_budget.py budget <scene 0|1> <stats.json> [SQ_INSTS_VALU per frame from the PMC pass]   (no GPU needed)

Issue classes (profiles/r02/valu_rates.txt, four waves per SIMD): `fast` = 2.4 clk per wave64 instruction (f32 add / sub / mul / fma /
mac, v_mov, 32-bit integer add / sub / xor); `slow` = ~4.2 clk (compares, selects, min / max / med3, shifts and logic, 64-bit and
mul integer, fp64, conversions, cross-lane); `trans` = 8.2 clk (rcp, rsq, sqrt).
"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracing_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
CFG = {0: (1920, 1080, 64, 4), 1: (1920, 1080, 256, 8)}


def stats(scene, out):
    import ctypes as C
    sys.path.insert(0, ROOT)
    import ray_tracing_amd as rt
    W, H, spp, nb = CFG[scene]
    g = rt.Renderer(0)
    g.set_skybox(rt.load_skybox()); g.set_scene(f"{rt.DATA_DIR}/scene_{scene}.txt")
    g.set_tuning(jit_flags="-DRT_STATS"); g.compile_scene()
    buf = (C.c_ulonglong * 64)()
    f = rt.lib().rt_spec_stats_read
    f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    f(g._ctx, buf, 1)
    g.render(W, H, spp, nb)
    # sites 32..63 live in words 64..127 of the module's rt_stats: read the whole array
    whole = (C.c_ulonglong * 128)()
    n = C.c_size_t(0)
    r = rt.lib().rt_spec_symbol_read
    r.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    assert r(g._ctx, b"rt_stats", whole, 128 * 8, C.byref(n)) == 0
    rc, rep = g.last_launch_report()
    json.dump({"scene": scene, "frame": [W, H, spp, nb], "samples": W * H * spp, "execs": {str(k): int(whole[2 * k]) for k in range(64)},
               "lanes": {str(k): int(whole[2 * k + 1]) for k in range(64)}, "waves": int(rep["waves_left"]), "object_pixels": int(rep["pixels_listed"])}, open(out, "w"), indent=1)
    print("wrote", out)


def build_listing(scene):
    """(address, opcode) of rt_trace_spec and the inline stack of every address."""
    torch_rtc = subprocess.check_output([sys.executable, "-c", "import importlib.util, os; s = importlib.util.find_spec('torch'); "
                                         "print(os.path.join(s.submodule_search_locations[0], 'lib', 'libhiprtc.so'))"], text=True).strip()
    d = tempfile.mkdtemp()
    co = os.path.join(d, "k.co")
    hdr = os.path.join(CSRC, "embedded", f"scene_{scene}.h")
    subprocess.check_call([sys.executable, os.path.join(CSRC, "compile_scene.py"), "hiprtc", torch_rtc, hdr, co, "-gline-tables-only"], stdout=subprocess.DEVNULL)
    dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)
    ins = []
    for l in dis.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s.*//\s*([0-9A-F]{12}):", l)
        if m: ins.append((int(m.group(2), 16), m.group(1)))
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + co, "--inlines", "--output-style=GNU", "--basenames"],
                         input="\n".join(hex(a) for a, _ in ins) + "\n", text=True, capture_output=True, check=True).stdout
    stacks, cur = [], []
    lines = sym.splitlines()
    i = 0
    # GNU style: function line, then file:line line, repeated per frame; frames of one address end where the next address's first function starts --
    # llvm-symbolizer separates addresses' outputs only in LLVM style, so ask again in that style (blank line between addresses)
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + co, "--inlines", "--basenames"],
                         input="\n".join(hex(a) for a, _ in ins) + "\n", text=True, capture_output=True, check=True).stdout
    for chunk in sym.strip().split("\n\n"):
        ls = chunk.splitlines()
        stacks.append([(ls[k], ls[k + 1]) for k in range(0, len(ls) - 1, 2)])       # innermost first: (function, file:line:col)
    assert len(stacks) == len(ins), (len(stacks), len(ins))
    return ins, stacks


def source_marks():
    src = open(os.path.join(CSRC, "rt_kernels.hip")).read().split("\n")
    def line_of(pat, after=0):
        for i, l in enumerate(src):
            if i + 1 > after and pat in l: return i + 1
        raise KeyError(pat)
    m = {}
    m["body"] = line_of("RT_DEV void wavefront_body(")
    m["sum_lambda"] = line_of("auto add_finished_samples = [&]()")
    m["loop"] = line_of("for (;; phase = phase == 2u")
    m["supply"] = line_of("---- 1. sample supply")
    m["fetch"] = line_of("if (nmask != 0ull) {", m["supply"])
    m["handout"] = line_of("if (!direct) {", m["fetch"] + 40)
    m["round_check"] = line_of("if (__ballot(f_live || ((rec1 | rec2) & REC_VALID) != 0) == 0ull)")
    m["shade"] = line_of("---- 2. shade the pending")
    m["specular"] = line_of("if (specular) {", m["shade"])
    m["specular_end"] = line_of("out_dir = scatter;", m["specular"])
    m["push"] = line_of("---- 3. the shadow taps go")
    m["trace_taps"] = line_of("auto trace_taps = [&](int count)")
    m["trace_taps_end"] = line_of("q_head += (unsigned int) count;", m["trace_taps"])
    m["tap_call_fast"] = line_of("{ STAT(40); trace_taps(64); }")
    m["tap_call_slow"] = line_of("{ STAT(41); trace_taps(64); }")
    m["tap_call_due"] = line_of("{ STAT(42); trace_taps(")
    m["bounce"] = line_of("---- 4. the bounce rays")
    m["due"] = line_of("const unsigned int due = phase == 2u", m["bounce"])
    m["back"] = line_of("---- 5. back: retire")
    m["back_valid"] = line_of("if (rec2 & REC_VALID) {", m["back"])
    m["back_taps"] = line_of("if (ptaps) {", m["back"])
    m["back_last"] = line_of("if (rec2 & REC_LAST) {", m["back"])
    m["back_shift"] = line_of("rec2 = rec1; sky2 = sky1; slot2 = slot1;")
    m["sum_call"] = line_of("---- 6. add the finished samples")
    m["leave"] = line_of("leave_launch<BLOCK>(&W, wave);")
    m["ball"] = line_of("RT_DEV bool ball_entry_fast(")
    m["ball_roots"] = line_of("if (!(discr > 0)) return false;", m["ball"])
    m["ball_slow"] = line_of("/* reference order, scene.c:117-127 */", m["ball"])
    m["ball_end"] = line_of("RT_DEV Hit nearest_hit_fast(")
    # statements of the rare fallback paths inside helpers (a wave with an operand outside an exact shortcut's window takes the plain IEEE form)
    m["push_one_by_one"] = (line_of("for (int kind = 2; kind < 5; kind++) {"), line_of("default: push((tapmask & 4) != 0, hp, tap_j2, 4); break;") + 1)
    m["push_lambda"] = (line_of("auto push = [&](bool on, V3 qo, V3 qd, int k) {"), line_of("auto tap_ray = [&]("))
    m["grids_from_memory"] = line_of(": rt_lit_bit_of(lit_grids_mem + hit.obj")
    m["tap_sides_slow"] = line_of("tapmask = (dot3(unit3_of_vector<FAST>(tap_j0), hn) > 0 ? 1 : 0)")
    m["sky_slow"] = (line_of("u = clamp11(nu / m);"), line_of("v = clamp11(nv / m);"))
    math = open(os.path.join(CSRC, "rt_math.hip.h")).read().split("\n")
    def mline(pat, after=0):
        for i, l in enumerate(math):
            if i + 1 > after and pat in l: return i + 1
        raise KeyError(pat)
    u = mline("RT_DEV V3 unit3_fast(V3 v)")
    m["unit3_fast_slow"] = (mline("RT_STAT_UNIT_SLOW;", u), mline("return mk3(v.x / len, v.y / len, v.z / len);", u))
    m["third_slow"] = mline("return x / 3.0f;")
    return m


FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32|^v_mov_b32|^v_(add|sub|subrev)_u32|^v_xor_b32|^v_add_co_u32$|^v_pk_")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")


def issue_class(op):
    if TRANS.match(op): return "trans"
    if re.match(r"^v_(add|sub|subrev|mul|fma|fmac|mac)_f32", op) or op.startswith("v_mov_b32") or re.match(r"^v_(add|sub|subrev)_u32", op) or op.startswith("v_xor_b32"):
        return "fast"
    return "slow"


def block_of(stack, m):
    """stack: innermost first.  Returns the block the instruction is booked to."""
    frames = [(fn, int(loc.split(":")[1]) if loc.count(":") >= 1 and loc.split(":")[1].isdigit() else 0, os.path.basename(loc.split(":")[0])) for fn, loc in stack]
    names = [f[0] for f in frames]
    # rare fallback paths, whatever section they sit in
    for fn, line, file in frames:
        if fn == "unit3" and file.startswith("rt_math"): return "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)"
        if fn.startswith("nearest_hit_fast") or fn.startswith("box_entry_fast"): return "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)"
        if fn.startswith("ball_entry_fast") and file.startswith("rt_kernels") and line >= m["ball_slow"]: return "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)"
        if file.startswith("rt_math") and (m["unit3_fast_slow"][0] <= line <= m["unit3_fast_slow"][1] or line == m["third_slow"]): return "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)"
        if file.startswith("rt_kernels") and (line in m["sky_slow"] or (fn.startswith("wavefront_body") and m["tap_sides_slow"] <= line <= m["tap_sides_slow"] + 2)): return "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)"
        if file.startswith("rt_kernels") and (fn.startswith("wavefront_body") or fn == "operator()") and \
           (m["push_one_by_one"][0] <= line <= m["push_one_by_one"][1] or m["push_lambda"][0] <= line < m["push_lambda"][1] or line == m["grids_from_memory"]):
            return "not taken on this frame (taps pushed kind by kind: ring nearly full; lit-taps grids read from memory: they are in LDS)"
    # lines of wavefront_body (outermost rt_kernels frame inside the body) and of its lambdas
    in_roots = any(fn.startswith("ball_entry_fast") and file.startswith("rt_kernels") and m["ball_roots"] < line < m["ball_slow"] for fn, line, file in frames) or \
               (any(fn.startswith("sqrt_of_float64") or fn.startswith("div_by_refined64") for fn in names))
    body_lines = [line for fn, line, file in frames if file.startswith("rt_kernels") and (fn.startswith("wavefront_body") or fn == "operator()") and line >= m["body"]]
    if not body_lines: return "prologue / other"
    in_tap_trace = any(m["trace_taps"] <= l <= m["trace_taps_end"] for l in body_lines)
    in_sum = any(m["sum_lambda"] <= l < m["loop"] for l in body_lines)
    outer = body_lines[-1]          # the line of wavefront_body itself (outermost frame)
    if in_sum: return "6 in-order sum"
    if in_tap_trace:
        site = "full batch, all kinds pushed at once" if outer == m["tap_call_fast"] else ("full batch, kinds pushed one by one (ring nearly full)" if outer == m["tap_call_slow"] else "what is left of a bounce due for retirement")
        return f"3 tap trace ({site}): fp64 sphere roots" if in_roots else f"3 tap trace ({site})"
    if outer >= m["leave"]: return "leave_launch"
    if outer >= m["sum_call"]: return "6 in-order sum"
    if outer >= m["back"]:
        if outer >= m["back_shift"]: return "5 back: hand-over of the round's results, sky texel"
        if outer >= m["back_last"]: return "5 back: a path ends (sky colour, clamp, window slot)"
        if outer >= m["back_taps"]: return "5 back: light term"
        if outer >= m["back_valid"]: return "5 back: emission, albedo"
        return "5 back: every round"
    if outer >= m["due"]: return "4 due taps loop"
    if outer >= m["bounce"]: return "4 bounce trace: fp64 sphere roots" if in_roots else "4 bounce trace, lit-cell lookup"
    if outer >= m["push"]: return "3 tap queue"
    if outer >= m["shade"]:
        if m["specular"] <= outer < m["specular_end"]: return "2 shade: specular direction"
        return "2 shade"
    if outer >= m["round_check"]: return "1 round check"
    if outer >= m["handout"]: return "1 supply: hand-out"
    if outer >= m["fetch"]: return "1 supply: pixel fetch"
    if outer >= m["supply"]: return "1 supply: attempt"
    return "prologue / other"


def budget(scene, stats_json, pmc_valu):
    st = json.load(open(stats_json))
    ex = {int(k): v for k, v in st["execs"].items()}
    samples = st["samples"]
    m = source_marks()
    ins, stacks = build_listing(scene)
    static = collections.defaultdict(collections.Counter)
    for (addr, op), stack in zip(ins, stacks):
        if not op.startswith("v_") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            kind = "lane" if op.startswith("v_") else None
            if not kind: continue
            static[block_of(stack, m)]["slow"] += 1
            continue
        static[block_of(stack, m)][issue_class(op)] += 1
    roots_share = ex[4] / max(ex[3], 1)              # sphere tests that go on to the roots / sphere tests, per wave-level test
    spheres = 3 if scene == 0 else 1
    execs = {
        "prologue / other": st["waves"], "leave_launch": st["waves"],
        "1 supply: attempt": ex[20], "1 supply: pixel fetch": ex[21], "1 supply: hand-out": ex[20], "1 round check": ex[7] + ex[23] * 0,
        "2 shade": ex[8], "2 shade: specular direction": ex[15],
        "3 tap queue": ex[7],
        "4 bounce trace, lit-cell lookup": ex[12], "4 bounce trace: fp64 sphere roots": ex[12] * roots_share, "4 due taps loop": ex[7],
        "5 back: every round": ex[16], "5 back: emission, albedo": ex[17], "5 back: light term": ex[38], "5 back: a path ends (sky colour, clamp, window slot)": ex[39],
        "5 back: hand-over of the round's results, sky texel": ex[16],
        "6 in-order sum": ex[23],
        "fallbacks (reference-order normalize, / instead of shared reciprocals, generic trace)": 0,
        "not taken on this frame (taps pushed kind by kind: ring nearly full; lit-taps grids read from memory: they are in LDS)": 0,
    }
    for site, e in (("full batch, all kinds pushed at once", ex[40]), ("full batch, kinds pushed one by one (ring nearly full)", ex[41]), ("what is left of a bounce due for retirement", ex[42])):
        execs[f"3 tap trace ({site})"] = e
        execs[f"3 tap trace ({site}): fp64 sphere roots"] = e * roots_share
    print(f"# Instruction budget of rt_trace_spec, scene_{scene} {st['frame'][0]}x{st['frame'][1]}, {st['frame'][2]} spp, {st['frame'][3]} bounces")
    print(f"# static VALU instructions of a block (the embedded kernels' compiler + line tables) x executions of the block by a wave (-DRT_STATS site counters)")
    print(f"# rounds per wave: {ex[7] / st['waves']:.0f}; waves: {st['waves']}; samples: {samples}; a sphere test goes on to its fp64 roots in {roots_share:.2f} of the wave-level tests\n")
    print(f"{'block':86s} {'static fast/slow/trans':>24s} {'execs/wave':>11s} {'VALU wave-instr per 64 samples: fast':>38s} {'slow':>9s} {'trans':>7s} {'share':>7s}")
    tot = collections.Counter(); rows = []
    for b in sorted(static):
        c = static[b]; e = execs.get(b)
        if e is None: e = 0; print("  (no execution count for", b, ")")
        dyn = {k: c[k] * e for k in ("fast", "slow", "trans")}
        tot.update(dyn); rows.append((b, c, e, dyn))
    grand = sum(tot.values())
    per = 64.0 / samples
    for b, c, e, dyn in rows:
        print(f"{b:86s} {c['fast']:8d}/{c['slow']:6d}/{c['trans']:5d} {e / st['waves']:11.1f} {dyn['fast'] * per:38.1f} {dyn['slow'] * per:9.1f} {dyn['trans'] * per:7.1f} {sum(dyn.values()) / grand * 100:6.1f}%")
    print(f"\n{'total':86s} {'':24s} {'':11s} {tot['fast'] * per:38.1f} {tot['slow'] * per:9.1f} {tot['trans'] * per:7.1f}")
    print(f"\nmodelled VALU instructions per frame: {grand:.4g}  (fast {tot['fast']:.4g}, slow {tot['slow']:.4g}, transcendental {tot['trans']:.4g})")
    clk = tot["fast"] * 2.4 + tot["slow"] * 4.2 + tot["trans"] * 8.2
    print(f"issue time at the measured class costs (2.4 / 4.2 / 8.2 clk): {clk / (1024 * 2.4e9) * 1e3:.3f} ms on 1024 SIMDs at 2.4 GHz; slow class = {tot['slow'] * 4.2 / clk * 100:.0f} % of it")
    if pmc_valu:
        print(f"SQ_INSTS_VALU of the PMC pass: {pmc_valu:.4g} per frame -> model / counter = {grand / pmc_valu:.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(int(sys.argv[2]), sys.argv[3])
    else:
        budget(int(sys.argv[2]), sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else None)
