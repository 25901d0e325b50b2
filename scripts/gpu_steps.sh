#!/bin/bash
# Runs named steps on the GPU box (through gpurun, from the repo root), each under its own timeout, logs under gpurun_out/<tag>/.
# A step that is killed at its limit (exit 124 / 137) ends the run: no further GPU step is started after a hang.
#   scripts/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "== $name ($(date +%T))"
    timeout -k 10 $secs bash -c "$cmd" > $O/$name.log 2> $O/$name.err
    rc=$?
    echo "   rc=$rc"; tail -n 3 $O/$name.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   step $name hit its limit: stopping"; exit 1; fi
done
exit 0
