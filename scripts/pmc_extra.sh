#!/bin/bash
# Extra PMC passes: instruction cache, branches, LDS conflicts.  usage: scripts/pmc_extra.sh <tag> [C1|C2|C3]
set -u
TAG=${1:-extra}
CFG=${2:-C1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_icache -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_icache.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_lds -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES --output-format csv -d $OUT/pmc_sq -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_icache", "pmc_lds", "pmc_sq"):
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for k in ("rt_trace", "rt_primary_pass"):
                if k in r["Kernel_Name"]: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            print("== %s: mean per launch of %s* ($CFG)" % (d, k))
            for c in sorted(agg[k]): print("%-32s %.6g" % (c, sum(agg[k][c]) / len(agg[k][c])))
PY
