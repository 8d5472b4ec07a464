#!/bin/bash
# A variant build of libpaml_amd.so for A/B runs (tools/ab.sh, PAML_AMD_LIB=...): one translation unit recompiled with extra flags,
# linked with the default build's other objects.   tools/build_variant.sh <name> <unit> <flags...>
#   e.g. tools/build_variant.sh nt3 engine_branch -DBEIG_STREAM=3   ->  paml_amd/lib/exp/libpaml_amd_nt3.so
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p paml_amd/lib/exp
python -c "from paml_amd import engine; engine.build()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c paml_amd/csrc/$unit.hip -o paml_amd/lib/exp/${unit}_$name.o
objs=""
for u in engine_core engine_comm engine_eval engine_branch engine_beb engine_jitdbg engine_compress; do
  if [ $u = $unit ]; then objs="$objs paml_amd/lib/exp/${unit}_$name.o"; else objs="$objs paml_amd/lib/obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o paml_amd/lib/exp/libpaml_amd_$name.so $objs -lhiprtc -ldl
echo paml_amd/lib/exp/libpaml_amd_$name.so
