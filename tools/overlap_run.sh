# kernel trace of the pipelined bench (driver flags, no extra legs) -> tools/overlap.py; per-scan kernel table of the serial loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/k
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/k/bench_trace -- python3 $R/bench.py --no-cpu-baseline --no-e2e --no-extra --steps 40 --warmup 5 --repeats 3 > $R/gpurun_out/k/bench_trace.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
cd $R
python3 tools/overlap.py gpurun_out/k/bench_trace
python3 tools/kernel_table.py gpurun_out/k/step_trace 20
