#!/bin/bash
# Round-5 evidence, run on the GPU box through gpurun:  tools/collect_profiles_r05.sh  -> gpurun_out/r5prof/
# kernel-trace statistics and counters (each counter group in its own --pmc pass with --kernel-trace only: MI355X_MICROARCH.md) of the
# bench's headline command; kernel-trace statistics of the keep-partials per-tree kernel (tools/keep_probe.py) and of the HIV M0 / M8
# evaluations on the cooperative per-tree kernel (tools/small_timeline.py).
out=$PWD/gpurun_out/r5prof
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
export PAML_AMD_DUAL=0      # launches one after the other: per-kernel durations and per-dispatch counters (see collect_profiles_r03.sh)
R=$PWD
B="python $R/bench.py --no-cpu-baseline --no-extras"
cd /tmp
prof() { d=$1; shift; rocprofv3 "$@" > $out/$d.out 2>$out/$d.err; }
prof stats      --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B
cp $out/stats.out $out/bench_under_rocprof.json
prof pmc_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B
prof pmc_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B
prof pmc_mfma   --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $B
prof stats_keep --kernel-trace --stats --output-format csv -d $out/stats_keep -o s -- python $R/tools/keep_probe.py 10
prof pmc_keep_w --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_keep_w -o w -- python $R/tools/keep_probe.py 4
cd - > /dev/null
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/stats_keep -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_keep.csv \;
{
  echo "# bench.py headline (16 taxa x 1e6 codon patterns, M0), PAML_AMD_DUAL=0: FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes), MFMA"
  for d in pmc_fetch pmc_write pmc_mfma; do python tools/pmc_summary.py $out/$d; done
  echo "# tools/keep_probe.py (the same workload, PAML_AMD_KEEP_PARTIALS: every internal node's partial stored): WRITE_SIZE per dispatch"
  python tools/pmc_summary.py $out/pmc_keep_w
} > $out/pmc_summary.txt 2>&1
python - "$out" <<'PY'
# HBM bytes per launch, corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE counts the 128-byte requests of wide streaming
# reads at 64 bytes on gfx950 -> doubled; WRITE_SIZE as reported (KiB)
import ast, json, sys
out = sys.argv[1]
sect, vals = None, {}
for ln in open(out + "/pmc_summary.txt"):
    if ln.startswith("#"):
        sect = "bench" if "bench.py" in ln else "keep"
        continue
    name, _, rest = ln.partition(" {")
    try:
        vals.setdefault(sect, {}).setdefault(name.strip(), {}).update(ast.literal_eval("{" + rest.strip()))
    except (SyntaxError, ValueError):
        pass
b = vals.get("bench", {}).get("prune_jit")
if b and "FETCH_SIZE" in b and "WRITE_SIZE" in b:
    json.dump({"kernel": "prune_jit", "workload": "bench.py headline (16 taxa x 1e6 codon patterns, M0)", "fetch_size_kib": b["FETCH_SIZE"], "write_size_kib": b["WRITE_SIZE"],
               "hbm_bytes_per_launch": (2 * b["FETCH_SIZE"] + b["WRITE_SIZE"]) * 1024.0, "correction": "2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes"},
              open(out + "/pmc.json", "w"), indent=1)
PY
for c in hiv_m0 hiv_m8 stewart; do
  timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1
  (cd /tmp && rm -rf /tmp/tr_$c && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -6)
done > $out/small_timeline.txt 2>&1
rm -rf $out/stats $out/stats_keep $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_keep_w
head -5 $out/kernel_stats.csv | cut -c1-160; head -5 $out/kernel_stats_keep.csv | cut -c1-160; cat $out/pmc_summary.txt | cut -c1-300; cat $out/small_timeline.txt
