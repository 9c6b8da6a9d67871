#!/bin/bash
# Runs on the GPU box (gpurun -- tools/refresh_profiles.sh [tag]): everything profiles/ is built from, into gpurun_out/refresh/.
# Afterwards, in the repo:  python tools/refresh_profiles_collect.py <tag>
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $R
python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
# config 4 WITH its CPU baseline beside it (one 8192-point cloud per CPU step: bench.py says so in `sample`)
python3 bench.py --workload c4 --steps 20 --warmup 3 >> $out/bench_other.jsonl 2>> $out/bench_other.err
for w in c3 c5 c2s c3f c3b; do python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline >> $out/bench_other.jsonl 2>> $out/bench_other.err; done
python3 bench.py --steps 20 --warmup 3 --inputs surface --no-cpu-baseline >> $out/bench_other.jsonl 2>> $out/bench_other.err
python3 bench.py --workload c4 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline >> $out/bench_other.jsonl 2>> $out/bench_other.err
python3 bench.py --workload c3 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline >> $out/bench_other.jsonl 2>> $out/bench_other.err
python3 tools/bench_infer.py >> $out/bench_other.jsonl 2>> $out/bench_other.err
cd /tmp; export TMPDIR=/tmp
# (the raw traces are hundreds of MB and gpurun copies at most 64 MiB back: only the statistics are kept)
slim() { find $1 -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --min-seconds 0.1 > $out/stats.log 2>&1
slim $out/stats
for w in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --min-seconds 0.1 > $out/stats_$w.log 2>&1
  slim $out/stats_$w
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 3 --eager --no-cpu-baseline --min-seconds 0 > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 3 --eager --no-cpu-baseline --min-seconds 0 > $out/pmc_write.log 2>&1
slim $out/pmc_fetch; slim $out/pmc_write
# SQ counters of the dominant kernel alone (kNN at the config-2 shape, 64 channels): MFMA busy, VALU / LDS activity, waits
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_knn/p$i -- python3 $R/tools/knn_split_prof.py 8 64 2048 20 > $out/pmc_knn_p$i.log 2>&1
done
cd $R
python3 tools/summarise_knn_pmc.py $out/pmc_knn knn_nominate > $out/knn_sq_counters.txt
python3 tools/summarise_knn_pmc.py $out/pmc_knn knn_refine >> $out/knn_sq_counters.txt
rm -rf $out/pmc_knn
python3 tools/knn_nominate_stamps.py 8 64 2048 20 2>/dev/null | grep -v amdgpu.ids > $out/knn_phase_stamps.txt
python3 tools/knn_nominate_stamps.py 8 3 2048 20 2>/dev/null | grep -v amdgpu.ids >> $out/knn_phase_stamps.txt
python3 tools/knn_split_check.py --time-only 2>/dev/null | grep -v amdgpu.ids > $out/knn_split_timing.txt
bash tools/pmc_step.sh refresh > /dev/null 2>&1; rm -rf $R/gpurun_out/pmc_step_refresh; cp $R/gpurun_out/pmc_step_refresh.txt $out/step_sq_counters.txt
grep -h ms_per_step $out/stats.log | cut -c1-200
