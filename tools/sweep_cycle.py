#!/usr/bin/env python3
"""Developer sweep: iterations and solve time of the 1 degree bench workload over cycle parameters (one matrix, one process)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="320x384x60")
ap.add_argument("--adv", default="upwind3")
ap.add_argument("--hmix", default="isop")
ap.add_argument("--sets", default="", help="semicolon-separated settings, each 'key=val,key=val' (nkp_tuning fields, or ml_smooth / restart)")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
b = np.random.default_rng(1).standard_normal(p.flat_len)
import torch
bd = torch.from_numpy(b).cuda()
xd = torch.zeros_like(bd)
torch.cuda.synchronize()
for spec in [""] + [s for s in a.sets.split(";") if s]:
    kv = dict(t.split("=") for t in spec.split(",") if t)
    opts = {k: int(kv.pop(k)) for k in ("ml_smooth", "restart") if k in kv}
    tune = {k: (float(v) if "." in v else int(v)) for k, v in kv.items()}
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, tuning=tune, **opts)
    s.solve_device(bd.data_ptr(), xd.data_ptr())
    t0 = time.perf_counter()
    info = s.solve_device(bd.data_ptr(), xd.data_ptr(), raise_on_fail=False)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(set=spec or "default", iters=info["iters"], solve_ms=round(dt * 1e3, 1), cycle_ms=round(s.time_kernel(1, reps=10), 3), status=info["status"])), flush=True)
    s.close()
