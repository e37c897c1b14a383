# PMC comparison of the extraction kernels at S = 21 (C4 shape): general against the dense variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_extract
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in general:0 dense0:1 dense1:1 dense3:1; do
  name=${v%%:*}; dense=${v##*:}
  export SHG_EXT_DENSE=$dense
  export SHG_EXT_DENSE_SHAPE=${name#dense}
  [ "$name" = general ] && export SHG_EXT_DENSE_SHAPE=0
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/${name}_sq -- python3 $R/tools/bench_extract.py 0 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -- python3 $R/tools/bench_extract.py 0 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -- python3 $R/tools/bench_extract.py 0 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O = 'gpurun_out/pmc_extract'
for name in ('general', 'dense0', 'dense1', 'dense3'):
    out = {}
    for kind in ('sq', 'fetch', 'write'):
        files = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, name, kind), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            k = r.get('Kernel_Name', '')
            if 'k_extract' not in k:
                continue
            grid = int(r.get('Grid_Size', 0) or 0)
            acc[(r['Counter_Name'], grid)].append(float(r['Counter_Value']))
        for (c, grid), v in acc.items():
            out.setdefault(grid, {})[c] = sum(v) / len(v)
    for grid, d in sorted(out.items()):
        print(name, 'grid', grid, ' '.join('%s=%.4g' % kv for kv in sorted(d.items())))
PY
