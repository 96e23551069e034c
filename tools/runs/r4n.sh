mkdir -p gpurun_out/r4n
( while true; do sleep 50; echo "alive $(date +%T)" >> gpurun_out/r4n/heartbeat.log; done ) &
HB=$!
cd /tmp && export TMPDIR=/tmp
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4n/prof -o quarter -- python3 $GRAFT_REPO_ROOT/tools/probe_gpu.py --grid 1440x720x80 --restart 60 > $GRAFT_REPO_ROOT/gpurun_out/r4n/probe.log 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
kill $HB
find gpurun_out/r4n/prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/r4n/quarter_deg_kernel_stats.csv
find gpurun_out/r4n/prof -type f ! -name "*stats*" -delete
head -n 7 gpurun_out/r4n/quarter_deg_kernel_stats.csv | cut -c1-200
tail -n 1 gpurun_out/r4n/probe.log | cut -c1-400
exit $rc
