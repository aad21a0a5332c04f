#!/bin/bash
# PMC passes of the culled trace on a synthetic large scene.  usage: scripts/pmc_large.sh <tag> [L256|L512|L1024]
set -u
TAG=${1:-pmc_large}; CFG=${2:-L1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_valu -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_valu.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_lds -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $OUT/pmc_busy -- python3 $R/scripts/render_cfg.py $CFG > $OUT/pmc_busy.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_valu", "pmc_lds", "pmc_sq", "pmc_busy"):
    for f in glob.glob("$OUT/%s/**/*_counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for k in ("rt_trace", "rt_primary_pass"):
                if k in r["Kernel_Name"]: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            print("== %s: mean per launch of %s* ($CFG)" % (d, k))
            for c in sorted(agg[k]): print("%-32s %.6g" % (c, sum(agg[k][c]) / len(agg[k][c])))
PY
