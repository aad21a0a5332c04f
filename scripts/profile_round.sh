#!/bin/bash
# Collect the round's profile evidence on the GPU box: bench line, rocprofv3 kernel-trace stats of the
# same command, and PMC passes (lane utilisation, HBM traffic).  usage: scripts/profile_round.sh <tag>
# Outputs under gpurun_out/<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_valu -- python3 $R/scripts/pmc_c1.py 0 > $OUT/pmc_valu.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/pmc_c1.py 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/pmc_c1.py 0 > $OUT/pmc_write.log 2>&1
cat $OUT/bench.json
