set -x
mkdir -p gpurun_out/r3m
run () {
  timeout -k 10 300 python tools/probe_gpu.py --grid $1 --refine $2 --max-iters 3000 > gpurun_out/r3m/tmp.log 2>&1
  tail -n 1 gpurun_out/r3m/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'refine', d['refine'], 'levels', d['levels'], 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
for om in 0.9 1.1 1.2 1.35; do
export NKP_ML_OMEGA=$om TAG="omega=$om"
run 320x384x60 1
run 100x116x60 12
done
unset NKP_ML_OMEGA
for bf in -1 1 2 4; do
export NKP_ML_BIG_FROM=$bf TAG="big_from=$bf"
run 320x384x60 1
run 640x768x60 1
done
