#!/bin/bash
# Development aid: build a patched copy of the library for scripts/ab.py.
# usage: scripts/variant.sh <name> [patch.py]   -> variants/<name>.so
# The sources are copied to and built in ${TMPDIR:-/tmp}/rt_variants/<name> (nothing but the finished library comes back into
# the tree: variants/ is git-ignored and holds .so files only, so that they travel to the GPU box with the snapshot).
# patch.py runs with cwd = the copied csrc directory and edits the sources in place.
# VARIANT_MAKE_ARGS: extra make arguments, e.g. "SPEC_COMPILER=hiprtc SPEC_HIPRTC=/path/libhiprtc.so" (the embedded scene kernels' compiler)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1
T=${TMPDIR:-/tmp}/rt_variants
D=$T/$N
rm -rf "$D"; mkdir -p "$D/ray_tracing_amd/csrc" "$D/include" "$R/variants"
cp $R/include/*.h "$D/include/"
cp $R/ray_tracing_amd/csrc/{Makefile,*.py,*.cpp,*.c,*.h,*.hip} "$D/ray_tracing_amd/csrc/"
if [ -n "${2:-}" ]; then (cd "$D/ray_tracing_amd/csrc" && python3 "$(cd "$(dirname "$2")" && pwd)/$(basename "$2")"); fi
make -C "$D/ray_tracing_amd/csrc" -j4 DATA="$R/data" ${VARIANT_MAKE_ARGS:-} ../librt_hip.so 2>&1 | grep -E "error|Error" || true
cp "$D/ray_tracing_amd/librt_hip.so" "$R/variants/$N.so"
ls -la "$R/variants/$N.so"
