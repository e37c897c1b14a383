python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused_limb or limb" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/step_trace -- python3 $R/tools/step_loop.py 20 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/k/step_trace 20 | grep -E "limb|library"
