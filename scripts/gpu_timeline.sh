#!/bin/bash
# rocprofv3 kernel trace of a python command on the GPU box -> one period of its repeating kernel sequence as a timeline
# (scripts/rocpd_timeline.py).   usage: bash scripts/gpu_timeline.sh <tag> <anchor-kernel-substring> <python args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=$1; ANCHOR=$2; shift 2
mkdir -p $O; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_$TAG
[ -f "$R/$1" ] && set -- "$R/$1" "${@:2}"
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o t -- python "$@" > $O/${TAG}_run.log 2>&1
DB=$(ls /tmp/prof_$TAG/*.db /tmp/prof_$TAG/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_timeline.py $DB "$ANCHOR" ${BACK:-3} > $O/${TAG}_timeline.txt 2>> $O/${TAG}_run.log
grep learner $O/${TAG}_run.log; tail -1 $O/${TAG}_timeline.txt
