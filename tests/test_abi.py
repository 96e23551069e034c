"""The C-ABI library loads on a CPU-only host, exports every symbol include/nkp.h declares,
and fails LOUDLY (no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from nk_ocn_tracer_jacobian_precond_amd import solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nkp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nkp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(solver.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = solver.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", solver.HIP_LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in declared_symbols():
        assert re.search(rf"\bT {name}\b", out), name


def test_default_options():
    o = solver.default_options()
    assert o.struct_size == C.sizeof(solver.NkpOptions)
    assert (o.precond, o.krylov, o.rtol) == (solver.PRECOND_MULTILEVEL, solver.KRYLOV_FGMRES, 1e-10)


def test_argument_validation_needs_no_gpu():
    rp = np.array([0, 1, 3], np.int32)
    with pytest.raises(solver.NkpError) as e:
        solver.NkpSolver(rp, np.array([0, 0, 5], np.int32), np.ones(3))       # column 5 out of range
    assert e.value.code == -1 and "out of range" in str(e.value)
    with pytest.raises(solver.NkpError) as e:
        solver.NkpSolver(rp, np.array([0, 0, 1], np.int32), np.ones(3), blk_start=np.array([0, 1], np.int32))
    assert e.value.code == -1


@pytest.mark.skipif(solver.device_count() > 0, reason="only meaningful on a host without a GPU")
def test_no_cpu_fallback():
    rp = np.array([0, 1, 2], np.int32)
    with pytest.raises(solver.NkpError) as e:
        solver.NkpSolver(rp, np.array([0, 1], np.int32), np.ones(2))
    assert e.value.code == -3 and "no HIP device" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """Nothing shipped may import, link or mention the oracle (only tests/, smoke() and bench's cpu_baseline)."""
    pkg = os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".cpp", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if f == "build.py":
                    continue          # build.py only *compiles* the oracle next to the product
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    for so in ("csrc/libnkp_hip.so", "host/libnkp_host.so"):
        needed = subprocess.run(["readelf", "-d", os.path.join(pkg, so)], capture_output=True, text=True).stdout
        assert "oracle" not in needed


def test_create64_refuses_what_one_gpu_cannot_index():
    """nkp_create64 (64-bit row pointers): more than 2^31 - 1 entries are refused with the way out (row-partition with
    nkp_create_dist) before any device is touched, so this runs without a GPU."""
    import ctypes as C
    import numpy as np
    from nk_ocn_tracer_jacobian_precond_amd import solver
    lib = solver.load_library()
    rp = np.array([0, 2 ** 31 + 5], np.int64)
    h = C.c_void_p()
    rc = lib.nkp_create64(C.byref(h), None, 1, rp.ctypes.data_as(C.POINTER(C.c_int64)), None, None, None, 0, 1)
    assert rc == -1 and "nkp_create_dist" in solver.last_error() and not h.value
    rp = np.array([0, 3, 2], np.int64)
    rc = lib.nkp_create64(C.byref(h), None, 2, rp.ctypes.data_as(C.POINTER(C.c_int64)), None, None, None, 0, 1)
    assert rc == -1 and "out of range" in solver.last_error()


def test_no_ablation_kernels_in_the_shipped_library():
    """The SpMV's timing-only bodies (NKP_SPMV_VARIANT 5..8: results wrong by design) are compiled only into the separate
    `make ablation` library, never into libnkp_hip.so / the executables."""
    out = subprocess.run(["nm", "-C", solver.HIP_LIB_PATH], capture_output=True, text=True, check=True).stdout
    kernels = [ln for ln in out.splitlines() if "csr_spmv_stream_kernel<" in ln]
    assert kernels, "the SpMV kernels should be visible to nm"
    assert not [ln for ln in kernels if re.search(r"csr_spmv_stream_kernel<\d+, [5-8],", ln)]


def test_kernels_and_cycle_do_not_read_the_environment():
    """Tuning knobs reach the kernels through nkp_tuning, resolved once per nkp_create (csrc/solver.hip: nkp_default_tuning);
    no launcher, cycle or setup routine calls getenv."""
    csrc = os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "csrc")
    for name in ("spmv.hip", "colblock.hip", "multilevel.hip", "mlsetup.hip", "mltail.hip", "blas1.hip"):
        assert "getenv" not in open(os.path.join(csrc, name)).read(), name
    text = open(os.path.join(csrc, "solver.hip")).read()
    body = text[text.index('extern "C" int nkp_default_tuning'):text.index("static int resolve_tuning")]
    assert text.count("getenv") == body.count("getenv")


def test_tuning_struct_round_trip():
    t = solver.default_tuning(ml_pocket=6)
    assert t.struct_size == C.sizeof(solver.NkpTuning) and t.ml_pocket == 6 and t.ml_omega == 1.1 and t.spmv_variant == 4
