mkdir -p gpurun_out/r4h
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r4h/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4h/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4h/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
for rep in 1 2; do
run NKP_DEFAULT=1
run NKP_COL_LDSRES=1
run NKP_COL_LDSRES=2 NKP_COLSTREAM_MIN=20000
done
