"""Emit C++ raw string literals holding source files: embed_sources.py file:symbol ...  (build step)"""
import sys
for spec in sys.argv[1:]:
    path, sym = spec.split(":")
    text = open(path).read()
    assert ")RTSRC\"" not in text
    # split into chunks: some compilers limit the length of a single literal
    chunks = [text[i:i + 12000] for i in range(0, len(text), 12000)]
    print(f"static const char {sym}[] =")
    for c in chunks:
        print('R"RTSRC(' + c + ')RTSRC"')
    print(";")
