for c in C1 C2 C3; do bash scripts/profile_round.sh r02_k_$c $c > /dev/null 2>&1; done
ls gpurun_out/
