#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3g; mkdir -p $O
for v in h2 fp4 s2 s2fp2; do
  OCC_HIP_LIB=$PWD/build/ab/libocc_$v.so timeout -k 10 200 python -m pytest tests/test_gpu_env_api.py -x -q -m gpu -k "shapenetcore_directory" > $O/t_$v.log 2>&1; echo "$v rc $?"; tail -2 $O/t_$v.log
done
timeout -k 10 200 python -m pytest tests/test_gpu_env_api.py -x -q -m gpu -k "shapenetcore_directory" > $O/t_default.log 2>&1; echo "default rc $?"; tail -2 $O/t_default.log
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "all rc $?"; tail -6 $O/pytest_all.log
