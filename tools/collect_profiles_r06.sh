#!/bin/bash
# Round-6 evidence, run on the GPU box through gpurun:  tools/collect_profiles_r06.sh  -> gpurun_out/r6prof/
# kernel-trace statistics and counters (each counter group in its own --pmc pass with --kernel-trace only: MI355X_MICROARCH.md) of the
# bench's headline command, and — what round 5's review asked to be re-profiled with the final build — of the 4-state kernel at 10^5 and
# at 10^7 patterns, the 20-state kernel, the branch-local evaluation, the keep-partials kernel and the kernel with more than 64 codes.
out=$PWD/gpurun_out/r6prof
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
export PAML_AMD_DUAL=0      # launches one after the other: per-kernel durations and per-dispatch counters (see collect_profiles_r03.sh)
R=$PWD
B="python $R/bench.py --no-cpu-baseline --no-extras"
cd /tmp
prof() { d=$1; shift; rocprofv3 "$@" > $out/$d.out 2>$out/$d.err; }
stats() { name=$1; shift; prof stats_$name --kernel-trace --stats --output-format csv -d $out/stats_$name -o s -- "$@"; find $out/stats_$name -name "*kernel_stats.csv" -exec cp {} $out/${name}_kernel_stats.csv \; ; grep -v "^W2026\|^E2026" $out/stats_$name.out | tail -3 > $out/${name}_under_rocprof.txt; rm -rf $out/stats_$name $out/stats_$name.out $out/stats_$name.err; }
pmc() { name=$1; ctr=$2; shift 2; prof pmc_${name}_$ctr --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${name}_$ctr -o p -- "$@"; }
prof stats      --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B
cp $out/stats.out $out/bench_under_rocprof.json
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
prof pmc_fetch  --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B
prof pmc_write  --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B
prof pmc_mfma   --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $B
stats keep    python $R/tools/keep_probe.py 10
stats c2      python $R/tools/c2_probe.py 100000
stats c2large python $R/tools/c2large_probe.py 10
stats m20     python $R/tools/aa_probe.py 32
stats m20_60  python $R/tools/aa_probe.py 60
stats branch  python $R/tools/branch_quick.py
stats amb     python $R/tools/amb_probe.py
for w in c2 c2large branch; do
  case $w in c2) C="python $R/tools/c2_probe.py 100000";; c2large) C="python $R/tools/c2large_probe.py 4";; branch) C="python $R/tools/branch_quick.py";; esac
  pmc $w FETCH_SIZE $C; pmc $w WRITE_SIZE $C
done
cd - > /dev/null
# the memory-side ceiling of the forming kernel's traffic mix (2 x 0.5 GB read, 0.5 GB written), and bench.py's branch block alone
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/hbm_mix_peak tools/hbm_mix_peak.hip && /tmp/hbm_mix_peak > $out/hbm_mix_peak.txt 2>&1
python tools/refill_probe.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $out/branch_block.json
{
  echo "# bench.py headline (16 taxa x 1e6 codon patterns, M0), PAML_AMD_DUAL=0: FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes), MFMA"
  for d in pmc_fetch pmc_write pmc_mfma; do python tools/pmc_summary.py $out/$d; done
  for w in c2 c2large branch; do
    echo "# $w: FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes)"
    python tools/pmc_summary.py $out/pmc_${w}_FETCH_SIZE; python tools/pmc_summary.py $out/pmc_${w}_WRITE_SIZE
  done
} > $out/pmc_summary.txt 2>&1
python - "$out" <<'PY'
# HBM bytes per launch, corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE counts the 128-byte requests of wide streaming
# reads at 64 bytes on gfx950 -> doubled; WRITE_SIZE as reported (KiB)
import ast, json, sys
out = sys.argv[1]
sect, vals = None, {}
for ln in open(out + "/pmc_summary.txt"):
    if ln.startswith("#"):
        sect = ln[2:].split(":")[0].split()[0]
        continue
    name, _, rest = ln.partition(" {")
    try:
        vals.setdefault(sect, {}).setdefault(name.strip(), {}).update(ast.literal_eval("{" + rest.strip()))
    except (SyntaxError, ValueError):
        pass
what = {"bench.py": ("pmc.json", "prune_jit", "bench.py headline (16 taxa x 1e6 codon patterns, M0)"),
        "c2": ("c2_pmc.json", "prune_jit", "tools/c2_probe.py 100000 (32 taxa x 1e5 nucleotide patterns, GTR+G4), this workload only"),
        "c2large": ("c2large_pmc.json", "prune_jit", "tools/c2large_probe.py (32 taxa x 1e7 nucleotide patterns, GTR+G4), this workload only"),
        "branch": ("branch_pmc.json", None, "tools/branch_quick.py (eval_branch at 16 taxa x 1e6 codon patterns)")}
for sect, (fn, kernel, wl) in what.items():
    ks = vals.get(sect, {})
    rows = {k: v for k, v in ks.items() if (kernel is None and ("branch" in k or "eig_kernel" in k)) or k == kernel}      # (pmc_summary.py keeps the tail of a long kernel name)
    res = {}
    for k, b in rows.items():
        if "FETCH_SIZE" in b and "WRITE_SIZE" in b:
            res[k] = {"fetch_size_kib": b["FETCH_SIZE"], "write_size_kib": b["WRITE_SIZE"], "hbm_bytes_per_launch": (2 * b["FETCH_SIZE"] + b["WRITE_SIZE"]) * 1024.0}
    if not res:
        continue
    src = "profiles/r06_%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2 x FETCH + WRITE)" % fn
    if kernel:
        d = dict(kernel=kernel, workload=wl, source=src, **res[kernel])
    else:
        d = dict(workload=wl, source=src, kernels=res)
    json.dump(d, open(out + "/" + fn, "w"), indent=1)
PY
rm -rf $out/stats; find $out -maxdepth 1 -type d -name "pmc_*" -exec rm -rf {} +; rm -f $out/pmc_*.out $out/pmc_*.err $out/stats.out $out/stats.err
cat $out/hbm_mix_peak.txt; ls $out; head -4 $out/kernel_stats.csv | cut -c1-160; cat $out/pmc_summary.txt | cut -c1-300; for f in $out/*_under_rocprof.txt; do echo "== $f"; cat $f; done
