mkdir -p gpurun_out/r4q
nproc
run() {
  env "$@" timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 --solve 0 > gpurun_out/r4q/tmp.log 2>&1 || { tail -n 20 gpurun_out/r4q/tmp.log; exit 1; }
  echo "== $*"; grep -h "multilevel setup:\|nkp_create: n =" gpurun_out/r4q/tmp.log | cut -c1-330
}
run NKP_SETUP_THREADS=16
run NKP_SETUP_THREADS=32
run NKP_SETUP_THREADS=64
run NKP_SETUP_THREADS=8
