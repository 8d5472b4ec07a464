# (PAML_AMD_M20_ABL is read by a library built with -DPAML_AMD_JIT_EXPERIMENTS only: PAML_AMD_LIB=<dir>/libpaml_amd.so PAML_AMD_EXTRA_FLAGS=-DPAML_AMD_JIT_EXPERIMENTS python -c "from paml_amd import engine; engine.build()")
for abl in none notip noa notip,noa; do echo "== $abl"; PAML_AMD_JIT_CACHE=0 PAML_AMD_M20_ABL=$abl python tools/m20_probe.py 2>/dev/null | head -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s prune %.4f ms  %.1f TF' % (d['case'], d['ms_prune'], d['tflops']))"; done
