#!/bin/bash
# Per-scan kernel tables (C2 and C4, one scan at a time under rocprofv3) for the current build, optionally with an environment
# setting to compare: tools/ab_tables.sh <tag> [VAR=value ...]  ->  gpurun_out/ab/<tag>_{c2,c4}.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/ab
C4=-10,-9,-8,-7,-6,-5,-4,-3,-2,-1,0,1,2,3,4,5,6,7,8,9,10
cd /tmp && export TMPDIR=/tmp
CFGS=${AB_CFGS:-c2 c4}
for cfg in "c2 20 0" "c4 10 $C4"; do
  case " $CFGS " in *" ${cfg%% *} "*) ;; *) continue;; esac
  set -- $cfg; name=$1; steps=$2; shift; shift
  rm -rf /tmp/ab_trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_trace -- python3 $R/tools/step_loop.py $steps "$@" > /dev/null 2>&1
  python3 $R/tools/kernel_table.py /tmp/ab_trace $steps > $R/gpurun_out/ab/${tag}_$name.txt
  rm -rf /tmp/ab_trace
done
cd $R
