#!/bin/bash
# One GPU-box visit: the GPU suite (PARITY lines kept), then bench.py with the driver's flags.  Output under gpurun_out/$1/.
tag=${1:-check}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $out/pytest.log 2>&1
echo "pytest rc=$?" > $out/summary.txt
grep -a "^PARITY" $out/pytest.log > $out/parity.txt
tail -n 15 $out/pytest.log >> $out/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
