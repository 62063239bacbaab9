"""Median duration per kernel from a rocprofv3 --kernel-trace csv directory: python tools/kernel_medians.py <dir> [substr]"""
import csv, glob, sys
d = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    v = sorted(v)
    print("%-70s n=%4d median %8.1f us  min %8.1f" % (k[:70], len(v), v[len(v) // 2], v[0]))
