#!/bin/bash
# Developer check (CPU only): build gen_A + the host library with AddressSanitizer / UBSan and run it over the
# committed golden option files and over every option file the gen_A tests generate.  GPU sanitizers are not
# available on the pool; this covers the host-side C (codec define mode, grid loader, matrix assembly).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/nk_ocn_tracer_jacobian_precond_amd
gcc -fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -fopenmp -ffp-contract=off -std=gnu11 \
    $P/host/nc3_codec.c $P/host/tracer_file_io.c $P/host/slab_alloc.c $P/host/strparse.c $P/host/matrix_file.c \
    $P/host/grid_file.c $P/host/matrix_gen.c $P/cli/gen_A_main.c -o /tmp/gen_A_asan -lm
rm -rf /tmp/pt_genA
(cd $ROOT && python -m pytest tests/test_gen_A.py -q --basetemp=/tmp/pt_genA > /dev/null)
n=0; bad=0
for f in $ROOT/tests/golden/gen_A_*.opt $(find /tmp/pt_genA -name "*.opt"); do
  (cd $(dirname $f) && ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=2 /tmp/gen_A_asan -o $f /tmp/asan_out.nc > /tmp/asan_run.log 2>&1) || true
  n=$((n+1))
  if grep -qi "runtime error\|AddressSanitizer" /tmp/asan_run.log; then bad=$((bad+1)); echo "ISSUE in $f"; grep -i "runtime error\|AddressSanitizer" /tmp/asan_run.log | head -3; fi
done
echo "ran $n option files, $bad with sanitizer findings"
[ $bad -eq 0 ]
