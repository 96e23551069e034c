set -x
mkdir -p gpurun_out/r3r
run () {
  timeout -k 10 300 python tools/probe_gpu.py --grid $1 --basis-f32 $2 --reorth $3 --max-iters 3000 > gpurun_out/r3r/tmp.log 2>&1
  tail -n 1 gpurun_out/r3r/tmp.log | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ.get('TAG'), d['grid'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'relres', d['relres'], 'j100_ms', round(d['arnoldi_ms_j100'],2))"
}
for cfg in "0 0" "1 0" "1 1" "0 1"; do set -- $cfg; export TAG="basis_f32=$1 reorth=$2"; run 320x384x60 $1 $2; run 100x116x60 $1 $2; run 640x768x60 $1 $2; done
