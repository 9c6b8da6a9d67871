#!/bin/bash
# Two ranks on ONE MI355X (gloo between them, every rank on device 0): the real bench.py N>1 launch path -- fwd/bwd graph,
# in-place all-reduce of FlatAdam's flat gradient buffer, optimizer graph.  RCCL itself needs >= 2 GPUs; this rehearses
# everything around it.  usage: tools/ddp_rehearsal.sh OUT_PREFIX [extra bench.py flags]
set -e
OUT=${1:-gpurun_out/ddp}; shift || true
export FSG_SHARE_GPU0=1 FSG_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 FSG_DUMP_MEMMAP=${OUT}_memmap
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 2 --steps 20 --warmup 3 --dump-check ${OUT}_check.npz "$@" > ${OUT}.json 2> ${OUT}.err
