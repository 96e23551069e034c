set -x
mkdir -p gpurun_out/r2j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2j/pmc_fetch -o fetch -- python3 tools/pmc_spmv.py > gpurun_out/r2j/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2j/pmc_write -o write -- python3 tools/pmc_spmv.py > gpurun_out/r2j/pmc_write.log 2>&1
find gpurun_out/r2j -name "*counter_collection.csv"
python tools/pmc_to_json.py $(find gpurun_out/r2j/pmc_fetch -name "*counter_collection.csv") $(find gpurun_out/r2j/pmc_write -name "*counter_collection.csv") gpurun_out/r2j/r2_spmv_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2j/prof -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --rhs-batch 0 --round1-steps 0 > gpurun_out/r2j/bench_prof.log 2>&1
tail -n 1 gpurun_out/r2j/bench_prof.log | cut -c1-1500
python bench.py > gpurun_out/r2j/bench.log 2> gpurun_out/r2j/bench.err
tail -n 1 gpurun_out/r2j/bench.log | cut -c1-3000
