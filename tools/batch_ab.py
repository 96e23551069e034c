#!/usr/bin/env python3
"""A/B of tuning knobs on the 1 degree bench workload: one solve at a time and 4 right-hand sides per sweep, same process.
usage: batch_ab.py name=knob:value,knob:value ..."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
import torch
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
R = int(os.environ.get("NKP_AB_R", "4"))
B = torch.randn((R, p.flat_len), dtype=torch.float64, device="cuda", generator=g)
X = torch.zeros_like(B)
for arg in sys.argv[1:]:
    name, _, spec = arg.partition("=")
    tune = {}
    for kv in filter(None, spec.split(",")):
        k, _, v = kv.partition(":")
        tune[k] = float(v) if "." in v else int(v)
    t0 = time.perf_counter()
    with solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, tuning=tune) as s:
        out = dict(name=name, tune=tune, create_s=time.perf_counter() - t0, levels=s.get_int("levels"), precond_ms=s.time_kernel(1, reps=20))
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            its = []
            for k in range(R):
                info = s.solve_device(B[k].data_ptr(), X[k].data_ptr())
                its.append(info["iters"])
            torch.cuda.synchronize(); out["single_ms"] = (time.perf_counter() - t0) / R * 1e3
            torch.cuda.synchronize(); t0 = time.perf_counter()
            info = s.solve_batch_device(B.data_ptr(), X.data_ptr(), R, p.flat_len)
            torch.cuda.synchronize(); out["batch_ms_per_solve"] = (time.perf_counter() - t0) / R * 1e3
        out["iters"] = its
        print(json.dumps(out), flush=True)
