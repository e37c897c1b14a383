python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "two_ranks" 2>&1 | tail -5
SHG_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --mode sharded --no-e2e --no-cpu-baseline --no-extra --repeats 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 ranks on one GPU (gloo), sharded series: ms/step', d['ms_per_step'], d['repeats']['ms_per_step'])"
