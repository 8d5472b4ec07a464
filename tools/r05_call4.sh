#!/bin/bash
# round 5, GPU call 4: the whole GPU suite on the tree as it stands + bench
O=gpurun_out/r05d; mkdir -p $O; cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo tests rc=$?
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
tail -n 25 $O/gpu_tests.txt | cut -c1-300
