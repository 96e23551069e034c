mkdir -p gpurun_out/r4l
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r4l/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4l/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4l/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run NKP_DEFAULT=1
run NKP_ML_SMOOTH_COARSE=2 NKP_ML_COARSE_FROM=2
run NKP_ML_SMOOTH_COARSE=2 NKP_ML_COARSE_FROM=3
run NKP_ML_SMOOTH_COARSE=4 NKP_ML_COARSE_FROM=2
run NKP_ML_SMOOTH_COARSE=2 NKP_ML_COARSE_FROM=1
run NKP_ML_SMOOTH_COARSE=1 NKP_ML_COARSE_FROM=3
