#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace`: one warm-up and one timed batched solve of 4 right-hand sides on the 1 degree bench workload."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
import torch
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
g = torch.Generator(device="cuda"); g.manual_seed(1)
B = torch.randn((R, p.flat_len), dtype=torch.float64, device="cuda", generator=g)
X = torch.zeros_like(B)
torch.cuda.synchronize()
s.solve_batch_device(B.data_ptr(), X.data_ptr(), R, p.flat_len)
torch.cuda.synchronize()
t0 = time.perf_counter()
info = s.solve_batch_device(B.data_ptr(), X.data_ptr(), R, p.flat_len)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(R=R, ms_total=dt * 1e3, ms_per_solve=dt / R * 1e3, iters=[i["iters"] for i in info])))
