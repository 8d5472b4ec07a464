#!/bin/bash
# Round-3 evidence, run on the GPU box through gpurun:  tools/collect_profiles_r03.sh  -> gpurun_out/r3prof/
# kernel-trace stats of the bench's headline command, the HBM traffic counters in separate --pmc passes (MI355X_MICROARCH.md:
# FETCH_SIZE / WRITE_SIZE each in its own run, with --kernel-trace only), MFMA / LDS counters, and the same for the 4-state probe.
out=$PWD/gpurun_out/r3prof
mkdir -p $out
export TMPDIR=/tmp
# Per-kernel figures: the launches one after the other.  By default consecutive evaluations alternate between two pruning streams
# and their kernels overlap — a kernel's "duration" in a trace then includes its wait for the CUs of the one before (about twice
# the evaluation period) and the per-dispatch counters of overlapping dispatches mix.  profiles/r03_dual_stream.txt has the
# overlapped timeline (tools/collect_dual_r03.sh).
export PAML_AMD_DUAL=0
B="python $PWD/bench.py --no-cpu-baseline --no-extras"      # bench.py with its default step counts, headline only
C2="python $PWD/tools/c2_probe.py"
M20="python $PWD/tools/m20_probe.py"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/bench_under_rocprof.json 2>$out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/pmc_lds -o l -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o s -- $C2 > $out/c2_under_rocprof.jsonl 2>$out/stats_c2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_c2_fetch -o f -- $C2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_c2_write -o w -- $C2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_m20 -o s -- $M20 > $out/m20_under_rocprof.txt 2>$out/stats_m20.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/pmc_m20_lds -o l -- $M20 > /dev/null 2>&1
cd - > /dev/null
{
  echo "# bench.py headline (16 taxa x 1e6 codon patterns, M0): FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes)"; python tools/pmc_summary.py $out/pmc_fetch; python tools/pmc_summary.py $out/pmc_write
  echo "# MFMA"; python tools/pmc_summary.py $out/pmc_mfma
  echo "# LDS"; python tools/pmc_summary.py $out/pmc_lds
  echo "# tools/c2_probe.py (32 taxa x 1e5 and x 4e6 nucleotide patterns, GTR+G4): FETCH_SIZE / WRITE_SIZE"; python tools/pmc_summary.py $out/pmc_c2_fetch; python tools/pmc_summary.py $out/pmc_c2_write
  echo "# tools/m20_probe.py (20 states): LDS"; python tools/pmc_summary.py $out/pmc_m20_lds
} > $out/pmc_summary.txt 2>&1
python - "$out" <<'PY'
# HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE counts the 128-byte
# requests of wide streaming reads at 64 bytes on gfx950 -> doubled; WRITE_SIZE as reported (KiB)
import ast, json, sys
out = sys.argv[1]
vals = {}
for ln in open(out + "/pmc_summary.txt"):
    if ln.startswith("# tools/c2_probe"):
        break
    if ln.startswith("prune_jit"):
        vals.update(ast.literal_eval(ln[len("prune_jit"):].strip()))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    json.dump({"kernel": "prune_jit", "workload": "bench.py headline (16 taxa x 1e6 codon patterns, M0)", "fetch_size_kib": vals["FETCH_SIZE"],
               "write_size_kib": vals["WRITE_SIZE"], "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
               "correction": "2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes"}, open(out + "/pmc.json", "w"), indent=1)
PY
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/stats_c2 -name "*kernel_stats.csv" -exec cp {} $out/c2_kernel_stats.csv \;
find $out/stats_m20 -name "*kernel_stats.csv" -exec cp {} $out/m20_kernel_stats.csv \;
rm -rf $out/stats_m20 $out/pmc_m20_lds $out/stats $out/stats_c2 $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_lds $out/pmc_c2_fetch $out/pmc_c2_write
head -6 $out/kernel_stats.csv; head -8 $out/c2_kernel_stats.csv; cat $out/pmc_summary.txt
