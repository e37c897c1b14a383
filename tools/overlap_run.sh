R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/k
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/k/bench_trace -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 40 --warmup 5 > $R/gpurun_out/k/bench_trace.json 2>/dev/null
cd $R
python3 tools/overlap.py gpurun_out/k/bench_trace
tail -c 600 gpurun_out/k/bench_trace.json
