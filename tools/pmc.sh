#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters> -- <program args...>   (one --pmc pass, CSV output)
out=$1; ctr=$2; shift 3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $out -- "$@" > $out.log 2>&1
