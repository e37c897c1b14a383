# per-scan kernel table of a C4 scan (21 disks), one scan at a time under rocprofv3: tools/c4_table.sh [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-10}
mkdir -p $R/gpurun_out/k
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/k/c4_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k/c4_trace -- python3 $R/tools/step_loop.py $K -10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10 > /dev/null 2>&1
cd $R
python3 tools/kernel_table.py gpurun_out/k/c4_trace $K > gpurun_out/k/c4_table.txt
rm -rf $R/gpurun_out/k/c4_trace
cat gpurun_out/k/c4_table.txt
