#!/bin/bash
# gpurun -- tools/quick_prof.sh TAG [bench.py flags]: rocprofv3 kernel statistics of a short bench.py run, condensed per step
# into gpurun_out/qp_TAG.txt (steps = calls of adam_flat_kernel)
R=$GRAFT_REPO_ROOT
tag=$1; shift
out=$R/gpurun_out/qp_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --steps 60 --warmup 3 --no-cpu-baseline --min-seconds 0.05 "$@" > $out/log.txt 2>&1
cd $R
f=$(ls $out/*/*kernel_stats.csv | head -1)
steps=$(python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(max(int(r["Calls"]) for r in rows if "adam_flat" in r["Name"]))
PY
)
python3 tools/prof_summary.py $f $steps 45 > $R/gpurun_out/qp_$tag.txt
grep -h ms_per_step $out/log.txt | cut -c1-160 >> $R/gpurun_out/qp_$tag.txt
