#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace --stats`: N applications of the multilevel cycle on the 1 degree bench workload
(kernel time per kernel name = where a V-cycle spends its time).  --ab VAR=a,b runs the same with an environment knob at
two settings in ONE process (same box, same clocks) and prints the cycle time of each."""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="320x384x60")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--ab", action="append", default=[], help="NAME=v1,v2,...: one solver per value of the environment variable")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
runs = [({}, "default")]
for spec in a.ab:
    name, vals = spec.split("=")
    runs += [({name: v}, f"{name}={v}") for v in vals.split(",")]
for env, label in runs:
    for k, v in env.items():
        os.environ[k] = v
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
    out = dict(run=label, cycle_ms=s.time_kernel(1, reps=a.reps), smoother_ms=s.time_kernel(3, reps=a.reps), column_ms=s.time_kernel(4, reps=a.reps),
               spmv_ms=s.time_kernel(0, reps=a.reps), cycle_bytes=s.get_int("cycle_bytes"))
    print(json.dumps(out), flush=True)
    s.close()
    for k in env:
        del os.environ[k]
