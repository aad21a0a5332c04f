#!/usr/bin/env python3
"""The resource usage of the kernels as BUILT, held against the pin in ray_tracing_amd/csrc/kernel_pin.json.

The compiled trace kernel sits at 125-127 of the 128 vector registers its four waves per SIMD leave it, with two dozen scalar
registers spilled into lanes of a vector register; a source change that is neutral under one compiler costs 4-7 % under another
(profiles/r05/ab_member_function_split.txt), and one register too many turns into scratch memory traffic in the hot loop.  None
of that shows in a parity test.  So the numbers the measured performance rests on are pinned:

  * every kernel: no scratch (`.private_segment_fixed_size` 0), no spilled vector registers;
  * per kernel: a cap on vector registers (what its launch bounds allow) and on spilled scalar registers (what it had when the
    committed profiles were taken, plus a little);
  * the two compilers -- the toolchain's hipcc for the library, and whatever made the embedded scene kernels (csrc/Makefile
    SPEC_COMPILER: PyTorch's bundled hiprtc when present) -- as recorded when the profiles/ evidence was taken.

`python scripts/kernel_resources.py`           print the table, exit 1 on a violation (tests/test_kernel_build_pin.py, build())
`python scripts/kernel_resources.py --pin`     rewrite the pin's `recorded` block and compilers from the current build (after
                                               re-measuring: the pin says which profiles it belongs to)
Reads code objects only (llvm-readelf --notes): no GPU, no compilation."""
import fnmatch
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracing_amd", "csrc")
PIN = os.path.join(CSRC, "kernel_pin.json")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = {"vgpr": ".vgpr_count", "agpr": ".agpr_count", "sgpr": ".sgpr_count", "sgpr_spill": ".sgpr_spill_count", "vgpr_spill": ".vgpr_spill_count",
          "scratch": ".private_segment_fixed_size", "lds_static": ".group_segment_fixed_size"}


def demangle(name):
    """_Z18rt_trace_wavefrontILb1ELb0ELb0EEv9rt_launchPj -> rt_trace_wavefront<1,0,0> (the kernels' template arguments are all bool)"""
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I((?:Lb[01]E)+)E", name)
    if not m:
        return name
    return m.group(1) + "<" + ",".join(re.findall(r"Lb([01])E", m.group(2))) + ">"


def kernels_of_code_object(path):
    """{kernel name: {vgpr, sgpr, sgpr_spill, vgpr_spill, scratch, ...}} from the code object's AMDGPU metadata note"""
    text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in re.split(r"\n  - \.agpr_count:", "\n" + text.split("amdhsa.kernels:", 1)[1].split("amdhsa.target:", 1)[0])[1:]:
        block = "  - .agpr_count:" + block
        name = re.search(r"^\s+\.name:\s+(\S+)", block, re.M).group(1)
        vals = {}
        for key, field in FIELDS.items():
            m = re.search(r"^\s+(?:- )?" + re.escape(field) + r":\s+(\d+)", block, re.M)
            vals[key] = int(m.group(1)) if m else 0
        out[demangle(name)] = vals
    return out


def device_code_of_host_object(obj, out_path):
    """the gfx950 code object hipcc embedded in a host object (.hip_fatbin section, clang offload bundle)"""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fatbin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out_path], check=True, capture_output=True)


def hipcc_version():
    v = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()
    return next((l.split(":", 1)[1].strip() for l in v if l.startswith("HIP version")), "?")


def built():
    """what the build left: {"library": {kernel: usage}, "embedded": {scene: {kernel: usage}}, "compilers": {...}}"""
    res = {"library": {}, "embedded": {}, "compilers": {"library": "hipcc " + hipcc_version(), "embedded": {}}}
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "rt_kernels.co")
        device_code_of_host_object(os.path.join(CSRC, "rt_kernels.o"), co)
        res["library"] = kernels_of_code_object(co)
    emb = os.path.join(CSRC, "embedded")
    for f in sorted(os.listdir(emb)) if os.path.isdir(emb) else []:
        if f.endswith(".co"):
            scene = f[:-3]
            res["embedded"][scene] = kernels_of_code_object(os.path.join(emb, f))
            comp = os.path.join(emb, scene + ".compiler")
            res["compilers"]["embedded"][scene] = open(comp).read().strip() if os.path.exists(comp) else "?"
    return res


def cap_for(caps, name):
    """the cap of the most specific (longest) pattern that matches the kernel's name"""
    for pattern in sorted(caps, key=len, reverse=True):
        if fnmatch.fnmatchcase(name, pattern):
            return caps[pattern]
    return None


def violations(res, pin):
    """every way the build differs from what the pin allows, as text lines; [] = the build is the one the profiles describe"""
    bad = []
    caps = pin["caps"]
    groups = [("librt_hip.so", res["library"])] + [("embedded " + s, k) for s, k in sorted(res["embedded"].items())]
    for where, kernels in groups:
        for name, u in sorted(kernels.items()):
            cap = cap_for(caps, name)
            if cap is None:
                bad.append(f"{where}: kernel {name} has no cap in kernel_pin.json (add one)")
                continue
            for key, limit in cap.items():
                if u[key] > limit:
                    bad.append(f"{where}: {name}: {key} = {u[key]} > {limit} (kernel_pin.json; recorded when pinned: {pin['recorded'].get(where, {}).get(name, {}).get(key, '?')})")
    for scene in pin["embedded_scenes"]:
        if scene not in res["embedded"] or "rt_trace_spec" not in res["embedded"].get(scene, {}):
            bad.append(f"embedded {scene}: no rt_trace_spec was built (csrc/Makefile SCENES / data/{scene}.txt)")
    for pattern in pin["required_kernels"]:
        if not any(fnmatch.fnmatchcase(n, pattern) for n in res["library"]):
            bad.append(f"librt_hip.so: no kernel matches {pattern}")
    return bad


def compiler_changes(res, pin):
    """the compilers of this build that are not the ones the committed profiles were measured with"""
    changed = []
    if res["compilers"]["library"] != pin["compilers"]["library"]:
        changed.append(f"library kernels: built by {res['compilers']['library']!r}, pinned {pin['compilers']['library']!r}")
    for scene, comp in sorted(res["compilers"]["embedded"].items()):
        if comp != pin["compilers"]["embedded"]:
            changed.append(f"embedded {scene}: built by {comp!r}, pinned {pin['compilers']['embedded']!r}")
    return changed


def table(res):
    rows = []
    groups = [("librt_hip.so", res["library"])] + [("embedded " + s, k) for s, k in sorted(res["embedded"].items())]
    for where, kernels in groups:
        for name, u in sorted(kernels.items()):
            rows.append(f"{where:20s} {name:34s} vgpr {u['vgpr']:3d}  sgpr {u['sgpr']:3d}  sgpr_spill {u['sgpr_spill']:3d}  vgpr_spill {u['vgpr_spill']:2d}  scratch {u['scratch']:3d} B  static lds {u['lds_static']:6d} B")
    return "\n".join(rows)


def main(argv):
    res = built()
    pin = json.load(open(PIN))
    if "--pin" in argv:
        pin["recorded"] = {"librt_hip.so": res["library"], **{"embedded " + s: k for s, k in res["embedded"].items()}}
        pin["compilers"]["library"] = res["compilers"]["library"]
        emb = sorted(set(res["compilers"]["embedded"].values()))
        if len(emb) == 1:
            pin["compilers"]["embedded"] = emb[0]
        json.dump(pin, open(PIN, "w"), indent=1, sort_keys=True)
        open(PIN, "a").write("\n")
        print("pinned", PIN)
    print(table(res))
    print("compilers:", json.dumps(res["compilers"]))
    bad = violations(res, pin)
    changed = compiler_changes(res, pin)
    for line in bad:
        print("VIOLATION:", line)
    for line in changed:
        print("COMPILER CHANGED:", line)
    return 1 if bad or changed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
