#!/bin/bash
# usage: pmc_icache.sh <lib.so> <tag>   -> instruction-cache counters of the trace kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RT_LIB_FILE=$R/$1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/ic_$2 -- python3 $R/scripts/pmc_lib.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/ic_$2/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("rt_trace"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$2", {k: "%.4g" % (sum(v)/len(v)) for k,v in acc.items()})
PY
