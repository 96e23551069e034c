#!/bin/bash
# Developer helper: run tests/dist_worker.py with N ranks on this machine (all ranks share GPU 0).
N=${1:-2}; MODE=${2:-gpu-solve}; GRID=${3:-40x46x20}; OUT=${4:-/tmp/nkp_dist}; EXTRA=${5:-}
PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 python tests/dist_worker.py --mode $MODE --grid $GRID --out $OUT $EXTRA > $OUT.log.$r 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for r in $(seq 0 $((N-1))); do echo "--- rank $r"; tail -3 $OUT.log.$r; cat $OUT.$r 2>/dev/null; echo; done
exit $rc
