mkdir -p gpurun_out/r4t
( while true; do sleep 50; echo "alive $(date +%T)" >> gpurun_out/r4t/heartbeat.log; done ) &
HB=$!
timeout -k 10 600 python tools/probe_gpu.py --grid 320x384x60 --tracers 4 --verbose 1 > gpurun_out/r4t/probe_c4.log 2>&1 || { kill $HB; tail -n 20 gpurun_out/r4t/probe_c4.log; exit 1; }
kill $HB
grep -h "multilevel setup:\|nkp_create: n =" gpurun_out/r4t/probe_c4.log | cut -c1-460
tail -n 1 gpurun_out/r4t/probe_c4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', 'setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'cycle_ms', d['precond_ms'])"
