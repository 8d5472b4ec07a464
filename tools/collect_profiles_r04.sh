#!/bin/bash
# Round-4 evidence, run on the GPU box through gpurun:  tools/collect_profiles_r04.sh  -> gpurun_out/r4prof/
# kernel-trace statistics and counters (each counter group in its own --pmc pass with --kernel-trace only: MI355X_MICROARCH.md) of
#   the bench's headline command, the branch-local evaluation (tools/branch_probe.py), and ONE workload each of the 4-state and the
#   20-state probes (so that a per-kernel average in the CSV is that workload's).
out=$PWD/gpurun_out/r4prof
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
export PAML_AMD_DUAL=0      # launches one after the other: per-kernel durations and per-dispatch counters (see collect_profiles_r03.sh)
R=$PWD
B="python $R/bench.py --no-cpu-baseline --no-extras"
BR="python $R/tools/branch_probe.py"
C2="python $R/tools/c2_probe.py 100000"
M20="python $R/tools/m20_probe.py 0"
export C2_NOCHECK=1 M20_NOCHECK=1
cd /tmp
prof() { d=$1; shift; rocprofv3 "$@" > $out/$d.out 2>$out/$d.err; }
prof stats      --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B
cp $out/stats.out $out/bench_under_rocprof.json
prof pmc_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B
prof pmc_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B
prof pmc_mfma   --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $B
prof stats_br   --kernel-trace --stats --output-format csv -d $out/stats_br -o s -- $BR
cp $out/stats_br.out $out/branch_probe_under_rocprof.json
prof pmc_br_f   --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_br_f -o f -- $BR
prof pmc_br_w   --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_br_w -o w -- $BR
prof pmc_br_m   --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_br_m -o m -- $BR
prof stats_c2   --kernel-trace --stats --output-format csv -d $out/stats_c2 -o s -- $C2
prof pmc_c2_f   --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_c2_f -o f -- $C2
prof pmc_c2_w   --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_c2_w -o w -- $C2
prof stats_m20  --kernel-trace --stats --output-format csv -d $out/stats_m20 -o s -- $M20
prof pmc_m20_i  --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_m20_i -o i -- $M20
prof pmc_m20_b  --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_m20_b -o b -- $M20
cd - > /dev/null
for k in "" _br _c2 _m20; do find $out/stats$k -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats$k.csv \; ; done
{
  echo "# bench.py headline (16 taxa x 1e6 codon patterns, M0), PAML_AMD_DUAL=0: FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes), MFMA"
  for d in pmc_fetch pmc_write pmc_mfma; do python tools/pmc_summary.py $out/$d; done
  echo "# tools/branch_probe.py (eval_branch at 16 taxa x 1e6 codon patterns): FETCH_SIZE / WRITE_SIZE / MFMA per dispatch"
  for d in pmc_br_f pmc_br_w pmc_br_m; do python tools/pmc_summary.py $out/$d; done
  echo "# tools/c2_probe.py 100000 (32 taxa x 1e5 nucleotide patterns, GTR+G4 — this workload only): FETCH_SIZE / WRITE_SIZE"
  for d in pmc_c2_f pmc_c2_w; do python tools/pmc_summary.py $out/$d; done
  echo "# tools/m20_probe.py 0 (20 states, 32 taxa x 1e5 patterns x 4 classes — this workload only): instruction mix, busy cycles"
  for d in pmc_m20_i pmc_m20_b; do python tools/pmc_summary.py $out/$d; done
} > $out/pmc_summary.txt 2>&1
python - "$out" <<'PY'
# HBM bytes per launch, corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE counts the 128-byte requests of wide streaming
# reads at 64 bytes on gfx950 -> doubled; WRITE_SIZE as reported (KiB)
import ast, json, sys
out = sys.argv[1]
sect, vals = None, {}
for ln in open(out + "/pmc_summary.txt"):
    if ln.startswith("#"):
        sect = "bench" if "bench.py" in ln else "branch" if "branch_probe" in ln else "c2" if "c2_probe" in ln else "m20"
        continue
    name, _, rest = ln.partition(" {")
    try:
        vals.setdefault(sect, {}).setdefault(name.strip(), {}).update(ast.literal_eval("{" + rest.strip()))
    except (SyntaxError, ValueError):
        pass
def hbm(v):
    return (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
b = vals.get("bench", {}).get("prune_jit")
if b and "FETCH_SIZE" in b and "WRITE_SIZE" in b:
    json.dump({"kernel": "prune_jit", "workload": "bench.py headline (16 taxa x 1e6 codon patterns, M0)", "fetch_size_kib": b["FETCH_SIZE"], "write_size_kib": b["WRITE_SIZE"],
               "hbm_bytes_per_launch": hbm(b), "correction": "2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes"}, open(out + "/pmc.json", "w"), indent=1)
for k, v in vals.get("c2", {}).items():
    if "jit" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        json.dump({"kernel": k, "workload": "tools/c2_probe.py 100000 (32 taxa x 1e5 nucleotide patterns, GTR+G4), this workload only", "fetch_size_kib": v["FETCH_SIZE"],
                   "write_size_kib": v["WRITE_SIZE"], "hbm_bytes_per_launch": hbm(v), "source": "profiles/r04_c2_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2 x FETCH + WRITE)"},
                  open(out + "/c2_pmc.json", "w"), indent=1)
json.dump({s: d for s, d in vals.items() if s == "branch"}, open(out + "/branch_pmc.json", "w"), indent=1)
PY
rm -rf $out/stats $out/stats_br $out/stats_c2 $out/stats_m20 $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_br_f $out/pmc_br_w $out/pmc_br_m $out/pmc_c2_f $out/pmc_c2_w $out/pmc_m20_i $out/pmc_m20_b
head -5 $out/kernel_stats.csv | cut -c1-160; head -8 $out/kernel_stats_br.csv | cut -c1-160; head -5 $out/kernel_stats_c2.csv | cut -c1-160; head -5 $out/kernel_stats_m20.csv | cut -c1-160; cat $out/pmc_summary.txt | cut -c1-400
