#!/bin/bash
export TMPDIR=/tmp
export PAML_AMD_CSRC=$PWD/paml_amd/csrc PAML_AMD_LIB=$PWD/paml_amd/lib/exp_prof/libpaml_amd.so
PAML_AMD_PROF_OPS=/tmp/ops.bin PAML_AMD_PROF_TID=0 python tools/small_timeline.py hiv_m0 30 2>&1 | tail -1
python tools/prof_ops.py /tmp/ops.bin
