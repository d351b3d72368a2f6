#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python command: tools/trace_cmd.sh <outdir> <python args...>   (run on the GPU box from the repo root)
O=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/$O" -o t -- python "$@" > "$REPO/$O.out" 2> "$REPO/$O.err")
python profiles/rocpd_stats.py "$(find "$O" -name '*_results.db' | head -n 1)" > "$O.stats.txt" 2>&1
head -n 25 "$O.stats.txt"
find "$O" -name '*.db' -size +20M -delete
