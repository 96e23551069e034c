mkdir -p gpurun_out/r4s
( while true; do sleep 50; echo "alive $(date +%T)" >> gpurun_out/r4s/heartbeat.log; done ) &
HB=$!
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_ml_plan.py -x -q -m gpu > gpurun_out/r4s/pytest.log 2>&1 || { kill $HB; tail -n 30 gpurun_out/r4s/pytest.log; exit 1; }
tail -n 2 gpurun_out/r4s/pytest.log
for g in 320x384x60 1440x720x80; do
timeout -k 10 600 python tools/probe_gpu.py --grid $g --restart 60 --verbose 1 --solve 0 > gpurun_out/r4s/probe_$g.log 2>&1 || { kill $HB; tail -n 20 gpurun_out/r4s/probe_$g.log; exit 1; }
grep -h "multilevel setup:\|nkp_create: n =" gpurun_out/r4s/probe_$g.log | cut -c1-460
done
kill $HB
