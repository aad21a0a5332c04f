#!/bin/bash
# Development aid: build a patched copy of the library for scripts/ab.py.
# usage: scripts/variant.sh <name> [patch.py]   -> build_variants/<name>/librt_hip.so
# patch.py runs with cwd = the copied csrc directory and edits the sources in place.
# VARIANT_MAKE_ARGS: extra make arguments, e.g. "SPEC_COMPILER=hiprtc SPEC_HIPRTC=/path/libhiprtc.so" (the embedded scene kernels' compiler)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1
D=$R/build_variants/$N
rm -rf "$D"; mkdir -p "$D/csrc" "$R/build_variants/include"
cp $R/include/*.h $R/build_variants/include/
cp $R/ray_tracing_amd/csrc/{Makefile,*.py,*.cpp,*.c,*.h,*.hip} "$D/csrc/"
if [ -n "${2:-}" ]; then (cd "$D/csrc" && python3 "$2"); fi
make -C "$D/csrc" -j4 DATA="$R/data" ${VARIANT_MAKE_ARGS:-} ../librt_hip.so 2>&1 | grep -E "error|Error" || true
ls -la "$D/librt_hip.so"
