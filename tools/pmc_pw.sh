#!/bin/bash
# SQ counter passes over the row product of csrc/pointwise.hip; usage: tools/pmc_pw.sh <tag> M N K tile
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_pw_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/pw_prof.py "$@" > $out/trace.log 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/pw_prof.py "$@" > $out/p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarise_knn_pmc.py $out pw_rowgemm > $out/summary.txt
find $out/trace -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-200 | grep pw_rowgemm >> $out/summary.txt
cat $out/summary.txt
