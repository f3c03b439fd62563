#!/bin/bash
# bench.py N = 2 plumbing on ONE GPU (gloo transport, tiny model): sequence-parallel video, replicas, context-parallel C5
mkdir -p gpurun_out
export PF_DIST_BACKEND=gloo
out=gpurun_out/r2_bench_n2_plumbing.log
: > $out
run() { echo "== $*" >> $out; ( timeout 600 "$@" 2>&1 | tail -3 | cut -c1-900 ) >> $out; }
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --workload smoke_128p_17f --tiny-model --no-cpu-baseline
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 0 --workload smoke_128p_17f --tiny-model --no-cpu-baseline --parallelism replicas
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 1 --warmup 0 --workload c5_vae_768p_241f --tiny-model --no-cpu-baseline
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 4 --steps 1 --warmup 0 --workload smoke_128p_17f --tiny-model --no-cpu-baseline
run python bench.py --workload smoke_128p_17f --tiny-model --no-cpu-baseline
cat $out
