#!/bin/bash
# dev helper, run ON the GPU box through gpurun:  bash devtools/gpu_check.sh <tag> [steps...]
# steps: tests bench bench0 prof pmc sq batch cfg3 rows2 (default: tests bench prof)
# Everything lands under gpurun_out/<tag>_*; every step has its own timeout so a hung kernel cannot eat the box.
tag=${1:-chk}; shift
steps=${@:-tests bench prof}
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for s in $steps; do
  case $s in
    tests)   timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log ;;
    rows2)   LSGPU_SO=$PWD/devtools/liblsgpu_exp.so LSGPU_KNN_ROWS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/${tag}_rows2.log 2>&1; echo "rows2 rc=$?"; tail -3 gpurun_out/${tag}_rows2.log ;;
    bench)   timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/${tag}_bench.json ;;
    bench0)  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compute-e2e > gpurun_out/${tag}_bench0.json 2> gpurun_out/${tag}_bench0.err; echo "bench0 rc=$?"; cut -c1-600 gpurun_out/${tag}_bench0.json ;;
    prof)    rm -rf gpurun_out/prof_${tag}
             (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_${tag} -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-compute-e2e > $OLDPWD/gpurun_out/${tag}_profbench.json 2> $OLDPWD/gpurun_out/${tag}_prof.err)
             db=$(find gpurun_out/prof_${tag} -name "*results.db" | head -1)
             [ -n "$db" ] && python profiles/summarize_rocpd.py $db > gpurun_out/${tag}_bench.stats.txt && head -16 gpurun_out/${tag}_bench.stats.txt ;;
    pmc)     for c in FETCH_SIZE WRITE_SIZE; do
               rm -rf gpurun_out/pmc_${tag}_$c
               (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/gpurun_out/pmc_${tag}_$c --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compute-e2e > /dev/null 2> $OLDPWD/gpurun_out/${tag}_pmc_$c.err)
             done
             f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
             python devtools/knn_traffic.py $f $w gpurun_out/${tag}_knn_traffic.json ${tag} 2>> gpurun_out/${tag}_pmc.err
             gzip -c $f > gpurun_out/${tag}_pmc_fetch.csv.gz; gzip -c $w > gpurun_out/${tag}_pmc_write.csv.gz ;;
    batch)   timeout 600 python devtools/batch_bench.py 3125 32 gpurun_out/${tag}_batch200k.json 2>&1 | tail -6
             timeout 600 python bench.py --batch --steps 3 --warmup 1 > gpurun_out/${tag}_bench_batch.json 2> gpurun_out/${tag}_bench_batch.err; echo "bench --batch rc=$?"; cut -c1-900 gpurun_out/${tag}_bench_batch.json ;;
    split)   timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --split --steps 5 --warmup 1 2> gpurun_out/${tag}_bench_split.err | grep '^{' > gpurun_out/${tag}_bench_split.json; echo "bench --split rc=$?"; cut -c1-900 gpurun_out/${tag}_bench_split.json ;;
    cfg3)    timeout 600 python devtools/config4_shape.py 16384 - gpurun_out/${tag}_config3_1gpu.json 2>&1 | tail -3 ;;
    sq)      rm -rf gpurun_out/sq_${tag}
             (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OLDPWD/gpurun_out/sq_${tag} --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compute-e2e > /dev/null 2> $OLDPWD/gpurun_out/${tag}_sq.err)
             python devtools/pmc_summary.py $(find gpurun_out/sq_${tag} -name "*counter_collection.csv" | head -1) k_knn_ > gpurun_out/${tag}_knn_sq_counters.txt 2>> gpurun_out/${tag}_sq.err; cat gpurun_out/${tag}_knn_sq_counters.txt ;;
  esac
done
