#!/bin/bash
# tools/pmc_pass.sh NAME "COUNTER COUNTER ..." -- one rocprofv3 --pmc pass over one C3 frame (tools/pmc_frame.py); single-block
# counter sets only.  Output: gpurun_out/NAME/ (CSV).  Diagnostic.
set -e
cd /tmp && export TMPDIR=/tmp
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
timeout -k 10 ${PRT_PMC_TIMEOUT:-180} rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o run -- python3 $GRAFT_REPO_ROOT/tools/pmc_frame.py > $out/log.txt 2>&1
tail -2 $out/log.txt
