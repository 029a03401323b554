#!/bin/bash
# dev helper, ON the GPU box: duration of k_cells_fill per pyramid level (one launch per level)
cd "$(dirname "$0")/.." || exit 1
R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pc; (cd /tmp && LSGPU_CELLS_SPLIT=1 rocprofv3 --kernel-trace -d /tmp/pc --output-format csv -- python $R/devtools/select_modes.py > /tmp/pc.out 2>&1)
f=$(find /tmp/pc -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_cells_fill" in r["Kernel_Name"]]
print(len(rows), "launches; durations us:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows][:14], "grid", rows[0]["Grid_Size_X"] if rows else None)
PY
