#!/bin/bash
# L2 -> fabric write bytes of the C1 trace kernel with pixels claimed n at a time (-DWF_CLAIM=n through rt_tuning.jit_flags).
# usage: scripts/pmc_claim.sh <tag> [C1|C1strip8|C2]
set -u
TAG=$1; CFG=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in 0 8 16; do
  OUT=$R/gpurun_out/$TAG/claim$n
  mkdir -p $OUT
  RT_JIT_FLAGS="-DWF_CLAIM=$n" rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_write.log 2>&1
  echo "== WF_CLAIM=$n ($CFG)"; python3 $R/scripts/summarize_pmc.py $OUT $CFG | grep -A4 "pmc_write: mean per launch of rt_trace"
done
