#!/bin/bash
# usage (GPU box): bash scripts/trace_kernel_sequence.sh <out.txt> <cmd...>  -- rocprofv3 kernel trace of a command, printed as the
# SEQUENCE of dispatches (name, start since the first dispatch, duration, gap to the previous end), the last $KSEQ_LAST (150) dispatches
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kseq && rocprofv3 --kernel-trace --output-format csv -d /tmp/kseq -- "$@" > /tmp/kseq.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/kseq/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); prev = None
with open(sys.argv[1], "w") as o:
    for r in rows[-int(__import__("os").environ.get("KSEQ_LAST", "150")):]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%-60s start %10.1f us  dur %8.1f us  gap %7.1f us\n" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
PY
