#!/bin/bash
# hipvoc-worker end to end over TCP (tools/worker_load.py), the legs of profiles/r01_worker_load.txt -> gpurun_out/r02/r02_worker_load.txt
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r02
mkdir -p $D
cd $R
O=$D/r02_worker_load.txt
: > $O
timeout -k 10 200 python3 tools/worker_load.py 8 40 4 f32 >> $O 2>&1 || exit 1
timeout -k 10 200 python3 tools/worker_load.py 12 60 6 bf16 >> $O 2>&1 || exit 1
timeout -k 10 200 python3 tools/worker_load.py 10 50 5 bf16x3 >> $O 2>&1 || exit 1
timeout -k 10 200 python3 tools/worker_load.py 1 40 1 f32 >> $O 2>&1 || exit 1
timeout -k 10 200 python3 tools/worker_load.py 1 40 1 bf16 >> $O 2>&1 || exit 1
timeout -k 10 200 python3 tools/worker_load.py 1 40 1 bf16x3 >> $O 2>&1 || exit 1
echo worker load done
