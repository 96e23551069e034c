set -x
mkdir -p gpurun_out/r4o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( while true; do sleep 50; echo "alive $(date +%T)" >> gpurun_out/r4o/heartbeat.log; done ) &
HB=$!
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r4o/pmc_fetch -o fetch -- python3 tools/pmc_cycle.py > gpurun_out/r4o/pmc_fetch.log 2>&1 || { kill $HB; tail -n 20 gpurun_out/r4o/pmc_fetch.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r4o/pmc_write -o write -- python3 tools/pmc_cycle.py > gpurun_out/r4o/pmc_write.log 2>&1 || { kill $HB; tail -n 20 gpurun_out/r4o/pmc_write.log; exit 1; }
kill $HB
python tools/pmc_cycle_to_json.py $(find gpurun_out/r4o/pmc_fetch -name "*counter_collection.csv") $(find gpurun_out/r4o/pmc_write -name "*counter_collection.csv") gpurun_out/r4o/pmc_fetch.log gpurun_out/r4o/r2_cycle_pmc.json
# keep only the rows of the kernels of interest (the full counter files are large)
for w in fetch write; do f=$(find gpurun_out/r4o/pmc_$w -name "*counter_collection.csv"); (head -n 1 $f; grep -E "csr_spmv_pipe_kernel<1, float|colblock_apply_ldsres_kernel<2, float|scale_to_kernel" $f) > gpurun_out/r4o/r2_cycle_pmc_${w}_rows.csv; done
find gpurun_out/r4o/pmc_fetch gpurun_out/r4o/pmc_write -type f -delete
