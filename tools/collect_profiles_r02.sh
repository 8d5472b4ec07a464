#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun): kernel-trace stats of the default bench command, then the
# HBM counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in separate --pmc runs, no other trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2prof
mkdir -p $OUT
CMD="python bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 3"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- $CMD > /dev/null 2> $OUT/pmc_$c.err
  find $OUT/pmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_$c.csv
done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
find $OUT/pmc_sq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_sq.csv
# C2 (4-state) the same way
CMD2="python tools/c2_probe.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace_c2 -o c2 -- $CMD2 > $OUT/c2_under_rocprof.jsonl 2> $OUT/trace_c2.err
find $OUT/trace_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c2_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_c2_$c -o pmc -- $CMD2 > /dev/null 2> $OUT/pmc_c2_$c.err
  find $OUT/pmc_c2_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_c2_$c.csv
done
rm -rf $OUT/trace $OUT/trace_c2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/pmc_c2_FETCH_SIZE $OUT/pmc_c2_WRITE_SIZE
ls -la $OUT
head -5 $OUT/kernel_stats.csv
