import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

GOLDEN_CASES = {
    "tri_12x10x6": ["IAGE"],
    "penta_12x10x6": ["IAGE", "TRACER2"],
    "cent_10x9x5": ["IAGE"],
    "pair_8x8x5": ["OCMIP_BGC_PO4", "OCMIP_BGC_DOP"],
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the host library, the HIP library and the oracle once (no-op when up to date)."""
    from nk_ocn_tracer_jacobian_precond_amd import build
    need = [os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "host", "libnkp_host.so"),
            os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "csrc", "libnkp_hip.so"),
            os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "bin", "solve_ABglobal"),
            os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "bin", "gen_A"),
            os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "bin", "nc_convert"),
            os.path.join(ROOT, "oracle", "libnkp_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        build.build_all()


class GoldenCase:
    def __init__(self, name):
        from nk_ocn_tracer_jacobian_precond_amd import nc3
        self.name = name
        self.varnames = GOLDEN_CASES[name]
        self.matrix_path = os.path.join(GOLDEN, f"{name}_matrix.nc")
        self.tracer_path = os.path.join(GOLDEN, f"{name}_tracers.nc")
        self.gold = np.load(os.path.join(GOLDEN, f"{name}_gold.npz"))
        f = nc3.NcFile(self.matrix_path)
        self.rowptr = f.get("rowptr")
        self.colind = f.get("colind")
        self.val = f.get("nzval_row_wise")
        self.cnt = int(f.get("coupled_tracer_cnt"))
        self.ind_i, self.ind_j, self.ind_k = (f.get(f"tracer_state_ind_to_{c}") for c in "ijk")
        self.imt, self.jmt, self.km = f.dims["nlon"], f.dims["nlat"], f.dims["z_t"]
        self.tsl = f.dims["tracer_state_len"]
        self.n = self.rowptr.size - 1
        starts = np.flatnonzero(self.ind_k == 0)
        self.col_start = np.concatenate([starts, [self.tsl]]).astype(np.int32)
        self.blk_start = np.concatenate([self.col_start[:-1] + t * self.tsl for t in range(self.cnt)] + [[self.n]]).astype(np.int32)
        t = nc3.NcFile(self.tracer_path)
        self.fields = {v: t.get(v) for v in self.varnames}

    def rhs(self, group_first):
        g = self.varnames.index(group_first)
        return np.concatenate([self.fields[v][self.ind_k, self.ind_j, self.ind_i] for v in self.varnames[g:g + self.cnt]])

    def groups(self):
        return [self.varnames[g] for g in range(0, len(self.varnames), self.cnt)]


@pytest.fixture(scope="session", params=list(GOLDEN_CASES))
def golden(request):
    return GoldenCase(request.param)


@pytest.fixture(scope="session")
def golden_by_name():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = GoldenCase(name)
        return cache[name]
    return get
