mkdir -p gpurun_out/r2c
run() { tag=$1; shift; env "$@" python tools/c2_probe.py > gpurun_out/r2c/$tag.jsonl 2> gpurun_out/r2c/$tag.err; echo "== $tag"; cat gpurun_out/r2c/$tag.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  %-40s sync %.4f dev %.4f prune %.4f reduce %.4f pmat %.4f frac %.3f' % (d['case'], d['ms_eval_sync'], d['ms_eval_device'], d['ms_prune'], d['ms_reduce'], d['ms_pmat'], d['valu_frac']))
"; tail -2 gpurun_out/r2c/$tag.err | cut -c1-300; }
run default A=1
run cw1 PAML_AMD_VF_CW=1 PAML_AMD_VF_R=1
run cw1_nocherry PAML_AMD_VF_CW=1 PAML_AMD_VF_R=1 PAML_AMD_VF_NOCHERRY=1
run cw2 PAML_AMD_VF_CW=2
run cw4 PAML_AMD_VF_CW=4
run cw4_nocherry PAML_AMD_VF_CW=4 PAML_AMD_VF_NOCHERRY=1
run r2 PAML_AMD_VF_CW=1 PAML_AMD_VF_R=2
