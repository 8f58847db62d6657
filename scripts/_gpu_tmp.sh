cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 1100 python scripts/parity_sweep.py 384 5000 wide > $O/sweep_wide3b.txt 2>&1; tail -1 $O/sweep_wide3b.txt | cut -c1-600; grep " BAD " $O/sweep_wide3b.txt | cut -c1-400
