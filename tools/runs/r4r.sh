mkdir -p gpurun_out/r4r
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 > gpurun_out/r4r/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4r/tmp.log; exit 1; }
  echo "== $*"; grep -h "multilevel setup:\|nkp_create: n =" gpurun_out/r4r/tmp.log | cut -c1-420
  tail -n 1 gpurun_out/r4r/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
run NKP_DEFAULT=1
run NKP_DEFAULT=1


