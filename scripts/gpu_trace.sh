#!/bin/bash
# rocprofv3 kernel trace of a python command on the GPU box -> per-kernel summary (scripts/rocpd_stats.py) and, with GAPS=<regex>, the
# idle gap in front of every matching kernel (scripts/rocpd_gaps.py).   usage: bash scripts/gpu_trace.sh <tag> <python args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=$1; shift
mkdir -p $O; export TMPDIR=/tmp; cd /tmp; rm -rf $O/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o t -- python "$@" > $O/${TAG}_run.log 2> $O/${TAG}_run.err
DB=$(ls $O/prof_$TAG/*.db $O/prof_$TAG/*/*.db 2>/dev/null | head -1)
tail -3 $O/${TAG}_run.log
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB > $O/${TAG}_kernel_stats.txt 2>> $O/${TAG}_run.err
[ -n "$DB" ] && [ -n "$GAPS" ] && python $R/scripts/rocpd_gaps.py $DB "$GAPS" > $O/${TAG}_gaps.txt 2>> $O/${TAG}_run.err
rm -rf $O/prof_$TAG
head -${TRACE_LINES:-40} $O/${TAG}_kernel_stats.txt | cut -c1-170
