timeout 600 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_progressive.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | head
for c in C1 C2 C3; do bash scripts/profile_round.sh r02_j_$c $c > /dev/null 2>&1; done
ls gpurun_out/
