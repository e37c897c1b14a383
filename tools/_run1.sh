python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 tools/bench_extract.py 2 2>&1 | tail -3
