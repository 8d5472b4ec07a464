for abl in none notip noa notip,noa; do echo "== $abl"; PAML_AMD_JIT_CACHE=0 PAML_AMD_M20_ABL=$abl python tools/m20_probe.py 2>/dev/null | head -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s prune %.4f ms  %.1f TF' % (d['case'], d['ms_prune'], d['tflops']))"; done
