mkdir -p gpurun_out/r4b
( while true; do sleep 50; echo "alive $(date +%T)" >> gpurun_out/r4b/heartbeat.log; done ) &
HB=$!
timeout -k 10 900 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --verbose 1 > gpurun_out/r4b/probe_quarter_deg.log 2>&1
rc=$?
kill $HB
tail -n 1 gpurun_out/r4b/probe_quarter_deg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('0.25 deg', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'], 'setup_s', d['setup_s'], 'spmv_GBs', d['spmv_GBs'])"
exit $rc
