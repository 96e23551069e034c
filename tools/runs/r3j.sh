set -x
mkdir -p gpurun_out/r3j
run () {
  timeout -k 10 200 python tools/probe_gpu.py --grid $1 --refine $2 > gpurun_out/r3j/tmp.log 2>&1
  tail -n 1 gpurun_out/r3j/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'refine', d['refine'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'levels', d['levels'], 'ml_rows', d['ml_rows'])"
}
for cfg in "0 16 0.01" "0.1 16 0.01" "0.25 16 0.01" "0.5 16 0.01" "0 4 0.01" "0 64 0.01" "0 256 0.01" "0 16 0.001" "0 16 0.1" "0 16 0.5"; do
set -- $cfg
export NKP_ML_THETA=$1 NKP_ML_POCKET=$2 NKP_ML_TAU=$3 TAG="theta=$1 pocket=$2 tau=$3"
run 320x384x60 1
run 100x116x60 12
done
