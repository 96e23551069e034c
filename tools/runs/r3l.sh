set -x
mkdir -p gpurun_out/r3l
run () {
  timeout -k 10 500 python tools/probe_gpu.py --grid $1 --refine $2 --k33 $3 --tracers $4 --restart $5 --max-iters 6000 > gpurun_out/r3l/tmp.log 2>&1
  tail -n 1 gpurun_out/r3l/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'refine', d['refine'], 'k33', d['k33'], 'n', d['n'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'status', d['status'])"
}
for pk in 4 16; do
export NKP_ML_POCKET=$pk TAG="pocket=$pk"
run 100x116x60 12 0 1 200
run 320x384x60 1 1 4 100
run 640x768x60 1 1 1 100
run 1440x720x80 1 1 1 60
done
