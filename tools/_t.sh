cd /root/repo
timeout 900 python -m pytest tests/test_precise_engine.py -x -q -m gpu -k random_shapes 2>&1 | tail -15 > gpurun_out/fuzz.txt
