set -x
mkdir -p gpurun_out/r3s
run () {
  timeout -k 10 300 python tools/probe_gpu.py --grid $1 --ml-smooth $2 --max-iters 3000 > gpurun_out/r3s/tmp.log 2>&1
  tail -n 1 gpurun_out/r3s/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
for cfg in "3 2" "3 1" "2 3" "4 2" "2 2" "4 1"; do set -- $cfg; export NKP_ML_POST=$2 TAG="V($1,$2)"; run 320x384x60 $1; run 640x768x60 $1; done
