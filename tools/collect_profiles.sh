#!/bin/bash
# Round profiles on the GPU box: kernel stats of the bench command, then the PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs).
#   bash tools/collect_profiles.sh <out dir under gpurun_out>
set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_profiler.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
   timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/spmv_$c -o pmc -- python3 $GRAFT_REPO_ROOT/tools/pmc_spmv.py > $OUT/pmc_spmv_$c.log 2>&1
   timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/cycle_$c -o pmc -- python3 $GRAFT_REPO_ROOT/tools/pmc_cycle.py > $OUT/pmc_cycle_$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_to_json.py $OUT/spmv_FETCH_SIZE/pmc_counter_collection.csv $OUT/spmv_WRITE_SIZE/pmc_counter_collection.csv $OUT/spmv_pmc.json
python3 $GRAFT_REPO_ROOT/tools/pmc_cycle_to_json.py $OUT/cycle_FETCH_SIZE/pmc_counter_collection.csv $OUT/cycle_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_cycle_FETCH_SIZE.log $OUT/cycle_pmc.json
rm -f $OUT/stats/bench_kernel_trace.csv          # 85 MB; the per-kernel summary is bench_kernel_stats.csv
for d in spmv_FETCH_SIZE spmv_WRITE_SIZE cycle_FETCH_SIZE cycle_WRITE_SIZE; do rm -f $OUT/$d/pmc_kernel_trace.csv; done
du -sh $OUT
tail -1 $OUT/bench_under_profiler.log | cut -c1-600
