for cfg in "--n 2000 --w 2000 --h 200 --bits 16" "--n 4000 --w 2000 --h 200 --bits 16" "--n 4000 --w 2560 --h 256 --bits 16" "--n 500 --w 2000 --h 200 --bits 16" "--n 2000 --w 2000 --h 200 --bits 8" "--n 200 --w 120 --h 800 --bits 8" "--n 2000 --w 200 --h 2000 --bits 16" "--n 3000 --w 1936 --h 150 --bits 16" "--n 2500 --w 3096 --h 120 --bits 16"; do
  echo "== $cfg"; python tools/bench_kernels.py $cfg 2>&1 | grep "pass"
done
