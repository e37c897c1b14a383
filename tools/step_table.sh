R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/k
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/k/step_trace 20 > gpurun_out/k/table.txt
cat gpurun_out/k/table.txt
