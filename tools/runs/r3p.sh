set -x
mkdir -p gpurun_out/r3p
for eq in 1 0; do
NKP_TEST_EQUIL=$eq timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "gen_A_to_solve_pipeline and coupled_pair" > gpurun_out/r3p/pytest_eq$eq.log 2>&1
grep -h "coupled_pair:\|passed\|failed" gpurun_out/r3p/pytest_eq$eq.log
done
