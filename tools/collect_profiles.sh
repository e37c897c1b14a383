#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench lines, rocprofv3 kernel trace + stats (default bench and --workers 1), two separate PMC passes (FETCH_SIZE,
#   WRITE_SIZE), per-scan kernel tables for C2 / C4 / C5.  Results land in gpurun_out/; tools/summarise_profiles.py turns
#   them into profiles/*.  Under rocprofv3 the program itself follows `--` (no env / bash -c hop).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> /dev/null
python3 $R/bench.py --workers 1 --no-e2e --no-extra > $O/${TAG}_bench_w1.json 2> /dev/null
for cfg in "c1 --frames 200 --width 800 --height 120 --bits 8" "c3_one_gpu --frames 4000" "c4 --shifts=-10:10:1" "c5_per_gpu --frames 4000 --width 2560 --height 256" "u8 --bits 8" "n500 --frames 500"; do
  set -- $cfg; t=$1; shift
  python3 $R/bench.py --no-cpu-baseline --no-e2e --no-extra "$@" > $O/${TAG}_bench_$t.json 2> /dev/null
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --no-e2e --no-extra --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_w1 -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 20 --warmup 5 > $O/${TAG}_bench_w1_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --workers 1 --no-cpu-baseline --no-e2e --no-extra --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
# per-scan kernel tables of the per-file flow alone (no probes, no generator)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_step_trace_c4 -- python3 $R/tools/step_loop.py 10 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_step_trace_c5 -- python3 $R/tools/step_loop.py 10 0 4000 2560 256 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_c4 -- python3 $R/tools/step_loop.py 3 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_c4 -- python3 $R/tools/step_loop.py 3 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch_c5 -- python3 $R/tools/step_loop.py 3 0 4000 2560 256 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write_c5 -- python3 $R/tools/step_loop.py 3 0 4000 2560 256 16 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/${TAG}_step_trace 20 > gpurun_out/${TAG}_step_kernel_table.txt
python3 tools/kernel_table.py gpurun_out/${TAG}_step_trace_c4 10 > gpurun_out/${TAG}_step_kernel_table_c4.txt
python3 tools/kernel_table.py gpurun_out/${TAG}_step_trace_c5 10 > gpurun_out/${TAG}_step_kernel_table_c5.txt
for f in gpurun_out/${TAG}_step_kernel_table*.txt; do tail -n 1 $f; done
