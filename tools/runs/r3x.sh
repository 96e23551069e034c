mkdir -p gpurun_out/r3x
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r3x/tmp.log 2>&1 || exit 1
  tail -n 1 gpurun_out/r3x/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run NKP_X=0
run NKP_COLSTREAM_MIN=20000
run NKP_COLSTREAM_MIN=5000
run NKP_COLWAVE_MAX=30000
run NKP_COLWAVE_MAX=3000
run NKP_COLSTREAM_MIN=20000 NKP_COLSTREAM_GW=16
