#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel stats of devtools/filter_time.py and the timeline of one reference filter
# (IDX: which k_gs_init to start from -- 3: a 1 M-point filter, 9: a 3.1 M-point one; NROWS: kernels to print)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-fprof}
rm -rf gpurun_out/prof_${tag}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -- python $OLDPWD/devtools/filter_time.py > /dev/null 2> $OLDPWD/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name "*results.db" | head -1)
[ -n "$db" ] && python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_filter.stats.txt && grep "k_gs_\|k_ssn\|k_seg\|k_scan" gpurun_out/${tag}_filter.stats.txt | head -30
python - <<PY
import sqlite3, re, glob
db = sqlite3.connect(glob.glob("gpurun_out/prof_${tag}/**/*results.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# first occurrence of k_gs_init -> timeline of one filter at 1M
names=[re.sub(r"^void |lsgpu::","",re.sub(r"\(.*","",r[0]))[:34] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith("k_gs_init")]
import os; i0=idx[int(os.environ.get("IDX","3"))]
t0=rows[i0][1]
prev=rows[i0][2]
for j in range(i0, min(i0+int(os.environ.get("NROWS","75")), len(rows))):
    print("%8.1f +%6.1f gap %5.1f %s" % ((rows[j][1]-t0)/1e3, (rows[j][2]-rows[j][1])/1e3, (rows[j][1]-prev)/1e3, names[j]))
    prev=rows[j][2]
PY
rm -rf gpurun_out/prof_${tag}
