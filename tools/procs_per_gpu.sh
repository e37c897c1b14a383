for n in 2 3 4 6; do
  SHG_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['n_gpus'], d['value'], d['ms_per_step'])"
done
