"""dev helper: profiles/knn_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: knn_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <tag>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest
skipped = [0]
def mean_counter(path, name):
    vals = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == name and ("k_knn_tile" in k or "k_knn_fallback" in k or "k_knn_rowq" in k):
            # launches enqueued behind the end of an alignment exit at once (a few microseconds, no traffic): not launches
            # of the search, left out of the mean like bench.py leaves them out of the launch time
            if "k_knn_tile" in k and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) < 20e3:
                skipped[0] += 1
                continue
            vals.setdefault("tile" if "k_knn_tile" in k else "fallback", []).append(float(r["Counter_Value"]))
    tile = vals.get("tile", [])
    fb = vals.get("fallback", [])
    # per kNN launch = one k_knn_tile dispatch (+ the wave-per-query pass where one follows)
    return (sum(tile) + sum(fb)) / max(len(tile), 1), len(tile), len(fb)
f, nt, nf = mean_counter(sys.argv[1], "FETCH_SIZE")
w, _, _ = mean_counter(sys.argv[2], "WRITE_SIZE")
out = {"n_az": 16384, "csrc_sha": csrc_digest(), "kernel": "k_knn_tile (+ the wave-per-query / row-per-query pass that follows it)",
       "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
       "hbm_bytes_per_launch": (2 * f + w) * 1024,
       "dispatches": {"k_knn_tile": nt, "k_knn_fallback + k_knn_rowq": nf, "k_knn_tile that exited at once (left out)": skipped[0] // 2},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace, mean over the "
                 "kNN launches (those that exit at once behind the end of an alignment excluded) of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compute-e2e` (the timed compute steps + the profiled loop steps of configs[1]); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                 "(gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated)",
       "source": ["profiles/%s_pmc_fetch.csv.gz" % sys.argv[4], "profiles/%s_pmc_write.csv.gz" % sys.argv[4]]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
