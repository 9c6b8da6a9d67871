#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
for cfg in "8 64 2048 20" "8 3 2048 20" "4 64 8192 40"; do python tools/knn_nominate_stamps.py $cfg; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_knn5_stamps.log
