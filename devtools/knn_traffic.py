"""dev helper: profiles/knn_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: knn_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <tag>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
skipped = [0]
per_kernel_all = {}
def mean_counter(path, name):
    vals = {}
    per_kernel = per_kernel_all.setdefault(name, {})
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == name and ("k_knn_tile" in k or "k_knn_cone" in k or "k_knn_fallback" in k or "k_knn_rowq" in k):
            # launches enqueued behind the end of an alignment exit at once (a few microseconds, no traffic): not launches
            # of the search, left out of the mean like bench.py leaves them out of the launch time
            main = "k_knn_tile" in k or "k_knn_cone" in k
            if main and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < 12e3:
                skipped[0] += 1
                continue
            vals.setdefault("tile" if main else "fallback", []).append(float(r["Counter_Value"]))
            if main: per_kernel.setdefault("cone" if "k_knn_cone" in k else "tile", []).append(float(r["Counter_Value"]))
    tile = vals.get("tile", [])
    fb = vals.get("fallback", [])
    # per kNN launch = one k_knn_tile dispatch (+ the wave-per-query pass where one follows)
    return (sum(tile) + sum(fb)) / max(len(tile), 1), len(tile), len(fb)
f, nt, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
w, _, _ = mean_counter(sys.argv[2], "WRITE_SIZE")
pk = {kk: {n: (sum(v) / max(len(v), 1), len(v)) for n, v in per_kernel_all[kk].items()} for kk in per_kernel_all}
out = {"n_az": 16384, "csrc_sha": csrc_digest(), "kernel": "one kNN launch = k_knn_cone (iterations >= 2) or k_knn_tile + the wave-per-query pass that follows it (iterations 0-1)",
       "per_kernel_kb": {"k_knn_cone": {"fetch": pk.get("FETCH_SIZE", {}).get("cone"), "write": pk.get("WRITE_SIZE", {}).get("cone")},
                         "k_knn_tile": {"fetch": pk.get("FETCH_SIZE", {}).get("tile"), "write": pk.get("WRITE_SIZE", {}).get("tile")}},
       "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
       "hbm_bytes_per_launch": (2 * f + w) * 1024,
       "dispatches": {"k_knn_cone + k_knn_tile": nt, "k_knn_fallback + k_knn_rowq": nf, "launches that exited at once (left out)": skipped[0] // 2},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace, mean over the "
                 "kNN launches (those that exit at once behind the end of an alignment excluded) of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compute-e2e` (the timed compute steps + the profiled loop steps of configs[1]); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                 "(gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated)",
       "source": ["profiles/%s_pmc_fetch.csv.gz" % sys.argv[4], "profiles/%s_pmc_write.csv.gz" % sys.argv[4]]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
