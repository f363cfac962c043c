#!/bin/bash
# rehearsal of bench.py's N>1 path on one GPU: two gloo ranks share cuda:0
PG_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --reads 20000
