export JDET_BENCH_FORCE_DIST=1
for sg in 1 0; do

JDET_DDP_STATIC_GRAPH=$sg timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$sg bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done
