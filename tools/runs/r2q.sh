set -x
mkdir -p gpurun_out/r2q
for st in 1 0; do
NKP_COLSTREAM=$st timeout -k 10 400 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --solve 0 > gpurun_out/r2q/probe_quarter_stream$st.log 2>&1
tail -n 1 gpurun_out/r2q/probe_quarter_stream$st.log | cut -c1-700
done
