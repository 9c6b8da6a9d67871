#!/bin/bash
# round 4: EdgeConv / DGCNN tests, config-2 bench line, per-kernel trace of the replayed step.  usage: tools/r4_step_run.sh TAG [pytest -k expr]
set -e
TAG=$1; KEXPR=${2:-"edgeconv or dgcnn or reproducible or seg_head or knn_gather or prepared"}
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$KEXPR" > gpurun_out/${TAG}_tests.log 2>&1 || echo "TESTS FAILED"
tail -4 gpurun_out/${TAG}_tests.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || (tail -5 gpurun_out/${TAG}_bench.err; exit 1)
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("roofline", {k:d["roofline"][k] for k in ("bound","frac","ceilings","avg_us","nominees_per_query")})
g=d["roofline_group_hbm"]; print("group us", g["us_per_step_hip_events"], g["us_per_step_graph_replay"], "frac", g["frac_vs_reference_bytes"])
print("head", d["roofline_head_product"])
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:45]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.2f}%")
PY
# the raw trace is hundreds of MB: keep the statistics only (gpurun copies at most 64 MiB back)
find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -type f ! -name "*kernel_stats.csv" -delete
