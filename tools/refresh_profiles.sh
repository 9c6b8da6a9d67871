#!/bin/bash
# Runs on the GPU box (gpurun -- tools/refresh_profiles.sh): everything profiles/ is built from, into gpurun_out/refresh/.
# Afterwards, in the repo:  python tools/refresh_profiles_collect.py
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/refresh
rm -rf $out; mkdir -p $out
cd $R
python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
for w in c4 c3 c5; do python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline >> $out/bench_other.jsonl 2>> $out/bench_other.err; done
python3 tools/bench_infer.py >> $out/bench_other.jsonl 2>> $out/bench_other.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 3 --eager --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 3 --eager --no-cpu-baseline > $out/pmc_write.log 2>&1
grep -h ms_per_step $out/stats.log | cut -c1-200
