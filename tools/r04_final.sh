#!/bin/bash
# round 4, second session: the evidence of the final build -> gpurun_out/r04f/
out=$PWD/gpurun_out/r04f
rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
R=$PWD
python bench.py --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench_driver_command.err
python bench.py > $out/bench_1gpu.json 2> $out/bench_1gpu.err
(cd /tmp && PAML_AMD_DUAL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python $R/bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/stats.err)
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; ; rm -rf $out/stats
for c in hiv_m0 hiv_m8 stewart brown; do
  (cd /tmp && rm -rf /tmp/tr_$c && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; tail -1 /tmp/tr_$c.log; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -5)
done > $out/small_timeline_final.txt 2>&1
timeout 600 python -m pytest tests/test_reference_binding_gpu.py -m gpu -q -s 2>&1 | grep -E "through|passed|failed" > $out/reference_binding.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
grep -E "passed|failed" $out/gpu_tests.txt | tail -2; grep -B30 "Error" $out/gpu_tests.txt | head -60
python - <<'PY'
import json
for f in ("bench_driver_command", "bench_1gpu"):
    try:
        d = json.loads(open("gpurun_out/r04f/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms_per_step %.4f value %.4g kernel_ms %.4f frac %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]),
              "c5 mle total", d.get("c5", {}).get("mle_seconds_total"), "c3", d.get("c3", {}).get("ms_per_eval_back_to_back"), d.get("c3", {}).get("kernel"))
    except Exception as e:
        print(f, "failed", e)
PY
head -4 $out/kernel_stats.csv | cut -c1-150; cat $out/reference_binding.txt | head -14
