"""Build step: embed_scenes.py <header.h>:<code.co>:<compiler.txt>[:<extra option>] ...  ->  C++ table of the scenes whose kernels were
compiled when the library was built (rt_jit.cpp looks a scene's generated header -- and the extra option it was asked for, "" or
-DRT_SPEC_AUDIT -- up in it before it turns to hiprtc)."""
import sys
print("struct rt_embedded_scene { const char *header; const unsigned char *code; size_t size; const char *compiler; const char *flags; };")
rows = []
for k, spec in enumerate(sys.argv[1:]):
    hdr, co, comp, flags = (spec.split(":") + [""])[:4]
    text = open(hdr).read()
    assert ")RTSRC\"" not in text
    code = open(co, "rb").read()
    print(f"static const char rt_emb_header_{k}[] = R\"RTSRC({text})RTSRC\";")
    print(f"alignas(4096) static const unsigned char rt_emb_code_{k}[] = {{")
    for i in range(0, len(code), 32):
        print(",".join(str(b) for b in code[i:i + 32]) + ",")
    print("};")
    rows.append(f"\t{{ rt_emb_header_{k}, rt_emb_code_{k}, {len(code)}, \"{open(comp).read().strip()}\", \"{flags}\" }},")
print("static const rt_embedded_scene rt_embedded_scenes[] = {")
print("\n".join(rows))
print("\t{ nullptr, nullptr, 0, nullptr, nullptr }\n};")
