#!/bin/bash
# round 4: PointTransformer tests, config-3 bench line, launch census of one step.  usage: tools/r4_pt_run.sh TAG [pytest -k expr]
set -e
TAG=$1; KEXPR=${2:-"pt_ or pointtransformer or PointTransformer or gemm_small or linear or forward_step or bn_rows or interp or group or transition"}
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$KEXPR" > gpurun_out/${TAG}_tests.log 2>&1 || echo "TESTS FAILED"
tail -4 gpurun_out/${TAG}_tests.log
python bench.py --workload c3 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || (tail -5 gpurun_out/${TAG}_bench.err; exit 1)
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
PY
python tools/diag_pt_launches.py > gpurun_out/${TAG}_launches.txt 2>&1 || true
grep "launches in the step" gpurun_out/${TAG}_launches.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline --min-seconds 0.1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("calls", sum(int(r['Calls']) for r in rows), "total ms", sum(float(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:50]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.2f}%")
PY
find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -type f ! -name "*kernel_stats.csv" -delete
