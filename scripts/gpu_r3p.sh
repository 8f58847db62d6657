#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3p
timeout -k 10 300 python scripts/dbg/kbuf_time.py > gpurun_out/r3p/kbuf.txt 2>&1; cat gpurun_out/r3p/kbuf.txt
