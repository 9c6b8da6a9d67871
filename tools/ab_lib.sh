#!/bin/bash
# A/B of two builds of libfsg_hip.so on ONE box, alternating.  usage: tools/ab_lib.sh WORKLOAD VARIANT.so [reps]
WL=$1; VAR=$2; REPS=${3:-3}
cd $GRAFT_REPO_ROOT
for r in $(seq $REPS); do
  for v in base var; do
    if [ $v = var ]; then export FSG_HIP_LIB=$GRAFT_REPO_ROOT/$VAR; else unset FSG_HIP_LIB; fi
    python bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done
