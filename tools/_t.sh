cd /root/repo
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_precise_engine.py -x -q -m gpu -k "3-1-0-20-80-1-5-7 or random_shapes" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-300; done > gpurun_out/x3_full.txt 2>&1
timeout 900 python -m pytest tests/test_precise_engine.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/x3_full.txt
timeout 600 python tools/bench_precise.py x3 2>&1 | grep -v "^/opt" >> gpurun_out/x3_full.txt
timeout 600 python tools/soak_precise.py 12 9 2>&1 | tail -1 >> gpurun_out/x3_full.txt
