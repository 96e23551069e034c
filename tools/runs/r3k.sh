set -x
mkdir -p gpurun_out/r3k
run () {
  timeout -k 10 200 python tools/probe_gpu.py --grid $1 --refine $2 --k33 $3 --max-iters 3000 > gpurun_out/r3k/tmp.log 2>&1
  tail -n 1 gpurun_out/r3k/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'refine', d['refine'], 'k33', d['k33'], 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
for pk in 1 2 3 4 6 8 12; do
export NKP_ML_POCKET=$pk TAG="pocket=$pk"
run 320x384x60 1 1
run 100x116x60 1 1
run 100x116x60 12 1
run 320x384x60 1 0
done
