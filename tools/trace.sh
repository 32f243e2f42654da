#!/bin/bash
# usage: tools/trace.sh <tag> <bench args...>   -- rocprofv3 kernel trace of bench.py; keeps only the text summary
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python3 $R/${SCRIPT:-bench.py} "$@" > /tmp/prof_$TAG.log 2>&1
tail -1 /tmp/prof_$TAG.log | cut -c1-300
mkdir -p $R/gpurun_out
python3 $R/tools/prof_summary.py $(find /tmp/prof_$TAG -name '*.db' | head -1) $R/gpurun_out/trace_$TAG.txt > /dev/null
head -${LINES_SHOWN:-40} $R/gpurun_out/trace_$TAG.txt
