#!/bin/bash
# round 5, final GPU call: the whole GPU suite, the driver's bench command, smoke, the rocprof evidence, the binding's timings
O=gpurun_out/r05z; mkdir -p $O; R=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo tests rc=$?
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo smoke rc=$?
bash tools/collect_profiles_r05.sh > $O/collect.txt 2>&1; echo collect rc=$?
bash tools/r05_call13.sh > /dev/null 2>&1; cp gpurun_out/r05n/codeml_gpu_timing.txt $O/codeml_gpu_timing.txt
tail -n 6 $O/gpu_tests.txt | cut -c1-300; tail -c 600 $O/bench.json; tail -3 $O/smoke.txt; tail -20 $O/collect.txt | cut -c1-250
