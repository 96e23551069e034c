set -x
mkdir -p gpurun_out/r3n
run () {
  timeout -k 10 400 python tools/probe_gpu.py --grid $1 --refine 1 --k33 $2 --restart $3 --max-iters 3000 > gpurun_out/r3n/tmp.log 2>&1
  tail -n 1 gpurun_out/r3n/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'k33', d['k33'], 'levels', d['levels'], 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
}
for cfg in "3 1.1" "-1 1.1" "-1 1.0" "4 1.1"; do
set -- $cfg
export NKP_ML_BIG_FROM=$1 NKP_ML_OMEGA=$2 TAG="big_from=$1 omega=$2"
run 1440x720x80 1 60
run 640x768x60 1 100
run 320x384x60 0 200
done
