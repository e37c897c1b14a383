python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace_c4 -- python3 $R/tools/step_loop.py 10 -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/k/step_trace_c4 10 | head -12; python3 tools/kernel_table.py gpurun_out/k/step_trace_c4 10 | tail -1
python3 tools/kernel_table.py gpurun_out/k/step_trace 20 | grep -E "products|scale_rows|warp|library"
