#!/bin/bash
# PMC passes over a few eager training steps of a bench workload; usage: tools/pmc_step.sh <tag> [bench args]
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcstep_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --eager --steps 3 --warmup 3 --no-cpu-baseline "$@" > $out/p$i.log 2>&1
done
