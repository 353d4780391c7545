export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06f; mkdir -p $OUT; cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
ZKIR_BENCH_BACKEND=gloo ZKIR_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 > $OUT/bench_n2.json 2> $OUT/bench_n2.err
ZKIR_BENCH_BACKEND=gloo ZKIR_BENCH_SHARE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 8 --steps 5 --warmup 2 > $OUT/bench_n8.json 2> $OUT/bench_n8.err
ZKIR_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_w1.json 2> $OUT/bench_w1.err
for f in bench bench_n2 bench_n8 bench_w1; do echo $f; tail -1 $OUT/$f.json | head -c 400; echo; done
