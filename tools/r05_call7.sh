#!/bin/bash
# round 5, GPU call 7: 20 states beyond the LDS capacity, the two-stage build of large trees, bench
O=gpurun_out/r05g; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "20_state or large_tree or size_limits or golden or random_models" > $O/t_engine.log 2>&1; echo engine rc=$?
timeout 300 python tools/big_tree_compile_probe.py 192 > $O/big_tree.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
tail -n 5 $O/t_engine.log; cat $O/big_tree.txt | tail -2
