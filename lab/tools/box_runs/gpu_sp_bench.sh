#!/bin/bash
# bench.py N=2 plumbing on ONE GPU (gloo transport, tiny model) + N=1 reference of the same tiny workload
mkdir -p gpurun_out
export PF_DIST_BACKEND=gloo
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --workload smoke_128p_17f --tiny-model --no-cpu-baseline 2>&1 | tail -4 ) > gpurun_out/bench_sp2_tiny.log
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 0 --workload smoke_128p_17f --tiny-model --no-cpu-baseline --parallelism replicas 2>&1 | tail -2 ) > gpurun_out/bench_rep2_tiny.log
( timeout 600 python bench.py --workload smoke_128p_17f --tiny-model --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_1_tiny.log
cat gpurun_out/bench_sp2_tiny.log gpurun_out/bench_rep2_tiny.log gpurun_out/bench_1_tiny.log | cut -c1-700
