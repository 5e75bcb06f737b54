export JDET_BENCH_FORCE_DIST=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-300
