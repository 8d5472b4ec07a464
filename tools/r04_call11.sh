#!/bin/bash
export TMPDIR=/tmp PROBE_NOCHECK=1 PAML_AMD_BEIG_CLOCK=1
python tools/branch_probe.py 2>&1 | grep "beig clock" | tail -12
