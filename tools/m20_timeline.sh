#!/bin/bash
# workgroup timeline of the 20-state kernel (case of tools/m20_probe.py)
export M20_NOCHECK=1 PAML_AMD_JIT_CACHE=0 PAML_AMD_PROF_TILES=1 PAML_AMD_PROF_OPS=/tmp/m20tl.bin
python tools/m20_probe.py ${1:-0} && python tools/prof_tiles.py /tmp/m20tl.bin
