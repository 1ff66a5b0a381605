#!/bin/bash
# tools/kstats.sh <command ...> -- rocprofv3 --kernel-trace --stats of a command, the per-kernel table printed (GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstats.XXXX)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o p -- "$@" > "$D/log" 2>&1)
grep -v "rocprofv3\|simple_timer\|output_stream" "$D/log" | tail -8
python - "$D" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f: sys.exit("no kernel stats")
for i, r in enumerate(csv.DictReader(open(f[0]))):
    if i < 40: print("%-72s %6s avg %10.1f us  min %9.1f  max %9.1f  %5s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"][:5]))
P
