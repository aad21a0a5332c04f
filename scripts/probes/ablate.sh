#!/bin/bash
# Development aid: timing-only ablations of the compiled trace kernel.  First: scripts/variant.sh ablations $PWD/scripts/patches/ablations.py
# (scripts/patches/ablations.py puts the #ifdef ABL_* blocks into a copy of the sources): the same library
# compiled by hiprtc without and with one -DABL_* at a time, interleaved on one device.  Frames are WRONG with any of them.
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/variants/ablations.so
for f in ABL_CVT ABL_RNG ABL_TAPTRACE ABL_F32ROOTS ABL_SKY ABL_BOUNCETRACE ABL_SUM ABL_SPEC; do
	echo "== $f"
	AB_CFGS=${AB_CFGS:-C1,C2} AB_JIT=1 AB_JITFLAGS_B=-D$f python3 $R/scripts/ab.py $LIB $LIB 5 2>&1 | grep -v amdgpu | cut -c1-140
done
