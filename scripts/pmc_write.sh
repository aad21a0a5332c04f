#!/bin/bash
# L2<->fabric traffic of one build.  usage: scripts/pmc_write.sh <tag> <lib.so> [C1|C2|C3]
set -u
TAG=$1; LIB=$2; CFG=${3:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RT_LIB_FILE=$R/$LIB
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_fetch.log 2>&1
python3 $R/scripts/summarize_pmc.py $OUT $CFG
