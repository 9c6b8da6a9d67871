#!/bin/bash
# SQ counter passes over a few EAGER steps of bench.py (every kernel of the step, as launched by the modules):
# gpurun -- tools/pmc_step.sh TAG [bench.py flags]  ->  gpurun_out/pmc_step_TAG.txt (tools/summarise_step_pmc.py)
R=$GRAFT_REPO_ROOT
tag=$1; shift
out=$R/gpurun_out/pmc_step_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $R/bench.py --steps 3 --warmup 2 --eager --no-cpu-baseline --min-seconds 0 "$@" > $out/p$i.log 2>&1
done
python3 $R/tools/summarise_step_pmc.py $out > $R/gpurun_out/pmc_step_$tag.txt
