#!/bin/bash
# usage: tools/rocprof_stats.sh <out-name> <python script and args...>
# rocprofv3 kernel trace + stats of one command (csv), summary printed and kept under gpurun_out/<out-name>/
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name.log 2>&1 || { tail -20 gpurun_out/$name.log; exit 1; }
python3 - "$name" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print("%-86s calls=%6s avg_us=%8.1f pct=%5.2f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
