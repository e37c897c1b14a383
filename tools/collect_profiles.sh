#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel trace + stats, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# Results land in gpurun_out/; tools/summarise_profiles.py turns them into profiles/*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
ls $R/gpurun_out/${TAG}_trace/*/ | head
# per-step kernel table of the per-file flow alone (no probes, no generator): 20 steps on a resident C2 stack
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
cd $R && python3 tools/kernel_table.py gpurun_out/${TAG}_step_trace 20 > gpurun_out/${TAG}_step_kernel_table.txt
