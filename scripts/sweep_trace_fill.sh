mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo smoke_rc=$?; tail -2 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo pytest_rc=$?; tail -5 gpurun_out/pytest_gpu.log
for k in 20 24; do for T in 256 512 1024 2048; do for TH in 256 512; do for NTS in 0 1; do
  echo "k=$k T=$T TH=$TH NT=$NTS $(ZKIR_TF_THREADS=$TH ZKIR_TF_NT=$NTS python bench.py --steps 30 --warmup 3 --log2-rows $k --tile-rows $T --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["achieved"]), round(d["value"]/1e9,2))')"
done; done; done; done > gpurun_out/sweep1.log 2>&1
cat gpurun_out/sweep1.log
