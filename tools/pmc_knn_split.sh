#!/bin/bash
# kernel trace + SQ counter passes over the coarse-sweep kNN kernel; usage: tools/pmc_knn_split.sh <tag> B C N k [flags]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_split_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/knn_split_prof.py "$@" > $out/trace.log 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/knn_split_prof.py "$@" > $out/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarise_knn_pmc.py $out knn_split > $out/summary.txt
find $out/trace -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-160 >> $out/summary.txt
