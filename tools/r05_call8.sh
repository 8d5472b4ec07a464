#!/bin/bash
# round 5, GPU call 8: trees beyond 207 tips on the per-tree kernel (two half blocks of tip codes per tile)
O=gpurun_out/r05h; mkdir -p $O; cd /root/repo
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "size_limits or large_tree or spills" > $O/t_engine.log 2>&1; echo engine rc=$?
PAML_AMD_JIT_ONE_STAGE= timeout 700 python tools/big_tree_compile_probe.py 400 > $O/big_tree_400.txt 2>&1
tail -n 4 $O/t_engine.log; tail -n 2 $O/big_tree_400.txt
