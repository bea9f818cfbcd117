#!/bin/bash
# Runs GPU steps one after another on the box; a step that hits its time limit (or is killed) ends the whole call
# -- no further GPU step is started after a timeout.  A step that merely FAILS (a red test) does not stop the rest.
#   tools/gpu_steps.sh "<seconds> <command>" "<seconds> <command>" ...
mkdir -p gpurun_out/r2
for spec in "$@"; do
    limit=${spec%% *}
    cmd=${spec#* }
    echo "[step] (limit ${limit}s) $cmd"
    timeout -k 10 "$limit" bash -o pipefail -c "$cmd"
    rc=$?
    echo "[step rc=$rc]"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[steps] time limit hit: stopping the call here"
        exit $rc
    fi
done
exit 0
