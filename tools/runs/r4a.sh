set -x
mkdir -p gpurun_out/r4a
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_resident or stream_kernel or wave_per_column" > gpurun_out/r4a/pytest_kernel.log 2>&1 || { tail -n 40 gpurun_out/r4a/pytest_kernel.log; exit 1; }
tail -n 3 gpurun_out/r4a/pytest_kernel.log
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r4a/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4a/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4a/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run NKP_X=0
run NKP_COL_LDSRES=2
python - <<'PY'
# 80-level columns at a size that fits a short run: 360x180x80 (1 degree x 80 levels)
PY
run2() {
  env "$@" timeout -k 10 300 python tools/probe_gpu.py --grid 720x360x80 --restart 60 > gpurun_out/r4a/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4a/tmp.log; exit 1; }
  tail -n 1 gpurun_out/r4a/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720x360x80 $*', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'], 'n', d['n'])"
}
run2 NKP_COL_LDSRES=0
run2 NKP_COL_LDSRES=1
