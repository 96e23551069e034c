set -x
mkdir -p gpurun_out/r3q
run () {
  timeout -k 10 300 python tools/probe_gpu.py --grid $1 --refine $2 --ml-smooth $3 --max-iters 3000 > gpurun_out/r3q/tmp.log 2>&1
  tail -n 1 gpurun_out/r3q/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'refine', d['refine'], 'levels', d['levels'], 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
export TAG="nu=3"; run 100x116x60 1 3
for nu in 2 4; do export TAG="nu=$nu"; run 320x384x60 1 $nu; run 100x116x60 1 $nu; done
for cr in 300 700 3000 5000; do export NKP_ML_COARSEST_ROWS=$cr TAG="coarsest_rows=$cr"; run 320x384x60 1 3; run 100x116x60 1 3; done
