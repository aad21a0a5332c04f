#!/bin/bash
# VALU instruction counts of one build.  usage: scripts/pmc_valu.sh <tag> <lib.so> [C1|C2|C3]
set -u
TAG=$1; LIB=$2; CFG=${3:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RT_LIB_FILE=$R/$LIB
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_valu -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_valu.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc_mix1 -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_mix1.log 2>&1
python3 $R/scripts/summarize_pmc.py $OUT $CFG | grep -v primary | grep -A9 "rt_trace"
