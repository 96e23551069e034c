mkdir -p gpurun_out/r4c
run2() {
  env "$@" timeout -k 10 300 python tools/probe_gpu.py --grid 720x360x80 --restart 60 > gpurun_out/r4c/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4c/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4c/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720x360x80 $*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run2 NKP_COLSTREAM_MIN=50000
run2 NKP_COLSTREAM_MIN=20000
run2 NKP_COLSTREAM_MIN=8000
run2 NKP_COLSTREAM_MIN=8000 NKP_COLWAVE_MAX=3000
