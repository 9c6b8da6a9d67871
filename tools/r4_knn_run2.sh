#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
python tools/knn_split_check.py > gpurun_out/r4_knn8_check.log 2>&1 || echo "CHECK FAILED" >> gpurun_out/r4_knn8_check.log
grep -E "MISMATCH|ALL EQUAL|!=|FAILED|Error|error" gpurun_out/r4_knn8_check.log | head -20
grep -E "^time|^stats" gpurun_out/r4_knn8_check.log
for cfg in "8 64 2048 20" "8 3 2048 20" "4 64 8192 40"; do python tools/knn_nominate_stamps.py $cfg; done 2>&1 | tee gpurun_out/r4_knn8_stamps.log
python tools/knn_nominate_stamps.py 8 64 2048 20 134217728 2>&1 | tee -a gpurun_out/r4_knn8_stamps.log
cd /tmp; export TMPDIR=/tmp
for cfg in "8 64 2048 20" "8 3 2048 20" "4 64 8192 40"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_knn8_trace_$tag -- python3 $GRAFT_REPO_ROOT/tools/knn_split_prof.py $cfg > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/r4_knn8_trace_$tag knn
done
