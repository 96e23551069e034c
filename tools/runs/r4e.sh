mkdir -p gpurun_out/r4e
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r4e/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4e/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4e/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run NKP_X=0
run NKP_COL_LDSRES=2 NKP_COLSTREAM_MIN=20000
run NKP_COL_LDSRES=2 NKP_COLSTREAM_MIN=5000
run NKP_COL_LDSRES=2 NKP_COLSTREAM_MIN=5000 NKP_COLWAVE_MAX=3000
