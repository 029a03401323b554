#!/bin/bash
# dev helper (GPU box): bench.py's value_track section alone, per-scan stages printed (LSGPU_GS_DEBUG=1: the library's diagnostics)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-trk}
timeout 900 python devtools/track_only.py > gpurun_out/${tag}_track.json 2> gpurun_out/${tag}_track.err; echo "track rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_track.json"))
print("value_track %.1f scans/s, median %.3f ms" % (d["value"], d["ms_per_scan_median"]))
print("ms_per_scan", d["ms_per_scan"])
print("iterations", d["icp_iterations"])
print("stages median", d.get("stages_ms_median"))
st=d.get("stages_ms",{})
for k in st: print(k, st[k])
print(d.get("cpu_baseline_track")); print(d.get("gpu_vs_cpu_transform"))
PY
