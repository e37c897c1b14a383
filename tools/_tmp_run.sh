timeout 900 python3 tools/soak_workers.py 100 4 2>&1 | tail -5
timeout 600 python3 tools/soak.py 2000 2>&1 | tail -6
