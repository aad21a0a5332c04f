#!/bin/bash
# Development aid: bench.py C1 with different rt_tuning.spare_workgroups on ONE box (does the frame's copy to the host overlap the next render?)
for s in "$@"; do
  python3 bench.py --no-cpu-baseline --no-extras --spare-workgroups $s 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('spare', $s, 'ms_per_step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'kernel', d['roofline']['avg_kernel_ms'])"
done
