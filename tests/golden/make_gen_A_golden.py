#!/usr/bin/env python3
"""Generate the gen_A fixtures in this directory (build container only; commits data, never runs on the GPU box).

The reference ships no input or output files for gen_A (test/test_gen_A.csh points at /glade paths), so
the inputs are synthetic POP-style history files (nk_ocn_tracer_jacobian_precond_amd.circ) and the expected
matrices come from the numpy restatement oracle/gen_A_oracle.py -- NOT from the C implementation under test.

  gen_A_circ_12x10x6.nc         float32 circulation file (all fields every option family reads)
  gen_A_sources_12x10x6.nc      tracer-source file (sink rates, d_J_*, piston velocity, d_SF_*)
  gen_A_<case>.opt              option file in the reference's grammar (paths relative to this directory)
  gen_A_<case>_expected.npz     rowptr, colind, nzval, KMT, index maps the restatement produced
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from nk_ocn_tracer_jacobian_precond_amd import circ, nc3  # noqa: E402

spec = importlib.util.spec_from_file_location("gen_A_oracle", os.path.join(ROOT, "oracle", "gen_A_oracle.py"))
ora = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ora)

CIRC, SRC = "gen_A_circ_12x10x6.nc", "gen_A_sources_12x10x6.nc"


def options(**kw):
    o = ora.default_options()
    o.update(kw)
    return o


CASES = {
    "shipped_job": options(adv="upwind3", hmix="isop_file", vmix="file", per_tracer=[dict(sink=("const_shallow", 365.0, 10.0e2))]),
    "cent_hor_file": options(adv="cent", hmix="hor_file", vmix="const", day_cnt=30.0, per_tracer=[dict(sink=("file", "SINK_RATE"), pv="PV")]),
    "pair_po4_dop": options(adv="donor", hmix="const", vmix="file", coupled_tracer_cnt=2, coupled_type="OCMIP_BGC_PO4_DOP",
                            per_tracer=[dict(sink=("generic_tracer", "ABIO_DIC14", 3)), dict(sink=("const", 0.5), sf="D_SF")]),
}


def opt_text(o):
    L = [f"circ_fname {CIRC}", f"tracer_fname {SRC}", f"day_cnt {o['day_cnt']!r}", f"adv_type {o['adv']}",
         f"l_adv_enforce_divfree {int(o['divfree'])}", f"hmix_type {o['hmix']}", f"vmix_type {o['vmix']}"]
    if o["coupled_tracer_cnt"] != 1:
        L.append(f"coupled_tracer_cnt {o['coupled_tracer_cnt']}")
    for t, p in enumerate(o["per_tracer"]):
        L.append(f"tracer_ind {t}")
        L.append("sink_type " + " ".join(repr(x) if isinstance(x, float) else str(x) for x in p["sink"]))
        for key in ("pv", "sf"):
            if p.get(key):
                L.append(f"{key} {p[key]}")
    L.append(f"coupled_tracer_type {o['coupled_type']}")
    return "\n".join(L) + "\n"


def read_back(path):
    f = nc3.NcFile(path)
    return ({nm: f.get(nm) for nm in f.vars},
            {nm: v.atts["_FillValue"][0] for nm, v in f.vars.items() if "_FillValue" in v.atts})


def main():
    F, fills = circ.make_circulation(12, 10, 6, seed=21)
    circ.write_circ_file(os.path.join(HERE, CIRC), F, fills, nc_type="float32")
    T = circ.make_tracer_sources(F, seed=21)
    circ.write_tracer_source_file(os.path.join(HERE, SRC), F, T)
    G, gf = read_back(os.path.join(HERE, CIRC))
    Tb = read_back(os.path.join(HERE, SRC))[0]
    for name, o in CASES.items():
        with open(os.path.join(HERE, f"gen_A_{name}.opt"), "w") as fh:
            fh.write(opt_text(o))
        w = ora.gen_A(G, gf, o, Tb)
        np.savez_compressed(os.path.join(HERE, f"gen_A_{name}_expected.npz"), rowptr=w["rowptr"], colind=w["colind"], nzval=w["nzval"],
                            KMT=w["KMT"], int3_to_tracer_state_ind=w["int3_to_tracer_state_ind"], ind_i=w["ind_i"], ind_j=w["ind_j"],
                            ind_k=w["ind_k"])
        print(name, "n =", w["flat_len"], "nnz =", len(w["nzval"]))


if __name__ == "__main__":
    main()
