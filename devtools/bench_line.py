"""dev helper: the few numbers of a bench.py line that a sweep compares.   python devtools/bench_line.py <bench.json>"""
import json, sys
b = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
tr = b.get("value_track", {})
print("value %.1f track %.1f" % (b["value"], tr.get("value", 0.0)),
      {k: round(v["scans_per_s"], 1) for k, v in b.get("compute_variants", {}).items() if isinstance(v, dict) and "scans_per_s" in v},
      "knn it0-2", b["roofline"]["per_iteration_us"][:3], "track tail ms", [round(x, 2) for x in tr.get("ms_per_scan", [])[-8:]])
