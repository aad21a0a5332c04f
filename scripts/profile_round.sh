#!/bin/bash
# Collect one config's profile evidence on the GPU box: bench line, rocprofv3 kernel-trace stats of the same
# command, and separate PMC passes (VALU busy / lane utilisation, L2<->fabric traffic).
# usage: scripts/profile_round.sh <tag> [C1|C2|C3]      outputs under gpurun_out/<tag>/
# then:  python3 scripts/summarize_pmc.py gpurun_out/<tag> <config> > profiles/rNN/<tag>_summary.txt
set -u
TAG=${1:-prof}
CFG=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config $CFG --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
# the same with ONE frame in flight (--depth 1: submit, wait, submit ...): no two launches share the GPU, so the bench line's
# per-launch HIP-event time and rocprofv3's mean kernel durations of this very process measure the same thing
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $R/bench.py --config $CFG --steps 20 --warmup 3 --depth 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof_serial.json 2> $OUT/stats_serial.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_valu -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_valu.log 2>&1
rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $OUT/pmc_busy -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_busy.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc_mix1 -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_mix1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_mix2 -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_mix2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_write.log 2>&1
cat $OUT/bench.json
