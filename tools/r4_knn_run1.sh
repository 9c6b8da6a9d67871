#!/bin/bash
# round 4: first run of the two-launch graph build (correctness, kNN tests, per-kernel trace)
set -e
cd $GRAFT_REPO_ROOT
python tools/knn_split_check.py > gpurun_out/r4_knn1_check.log 2>&1 || echo "CHECK FAILED" >> gpurun_out/r4_knn1_check.log
tail -30 gpurun_out/r4_knn1_check.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn" > gpurun_out/r4_knn1_tests.log 2>&1 || echo "TESTS FAILED"
tail -5 gpurun_out/r4_knn1_tests.log
cd /tmp; export TMPDIR=/tmp
for cfg in "8 64 2048 20" "8 3 2048 20" "4 64 8192 40"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_knn1_trace_$tag -- python3 $GRAFT_REPO_ROOT/tools/knn_split_prof.py $cfg > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/r4_knn1_trace_$tag knn
done
