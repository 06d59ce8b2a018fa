"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/lsm.h declares (no compute calls without a GPU), the ctypes mirrors match the C structs,
and the host-side mirror of the reference interface behaves like the reference
(test/test-boundaryconditions.jl, test/test-meshes.jl, test/test-meshfield.jl)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


def test_library_exports_every_declared_symbol(lsm):
    hdr = open(os.path.join(ROOT, "include", "lsm.h")).read()
    declared = sorted(set(re.findall(r"\b(lsm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    lib = lsm._lib.lib()          # raises if libhiplsm.so is missing: no fallback
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lsm.h but not exported"
    assert sorted(lsm._lib.EXPORTS) == declared, "ctypes binding and header disagree"
    assert lib.lsm_version().decode().startswith("hiplsm")


def test_ctypes_structs_match_the_header(lsm):
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "lsm.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(LsmGrid), sizeof(LsmBc), sizeof(LsmSlab), sizeof(LsmLayout), sizeof(LsmCoeff), sizeof(LsmTerm));
  printf("%zu %zu %zu %zu\n", offsetof(LsmGrid,n), offsetof(LsmGrid,lc), offsetof(LsmCoeff,field), offsetof(LsmTerm,s0));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    L = lsm._lib
    sizes = [C.sizeof(x) for x in (L.LsmGrid, L.LsmBc, L.LsmSlab, L.LsmLayout, L.LsmCoeff, L.LsmTerm)]
    offs = [L.LsmGrid.n.offset, L.LsmGrid.lc.offset, L.LsmCoeff.field.offset, L.LsmTerm.s0.offset]
    assert [int(x) for x in out] == sizes + offs


def test_no_gpu_means_loud_failure(lsm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    grid = lsm.CartesianGrid((0, 0), (1, 1), (16, 16))
    ic = lsm.MeshField(lambda x: x[0] + x[1], grid)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0, 0.0)),), ic=ic, bc=lsm.NeumannBC())
    # the raw ABI reports, never aborts
    L = lsm._lib
    h = C.c_void_p()
    from lsm_amd.api import _bc_c, _normalize_bc
    code = L.lib().lsm_create(C.byref(grid._c()), _bc_c(_normalize_bc(lsm.NeumannBC(), 2), 2), None, 0, 0, 0, C.byref(h))
    assert code == L.ERR_NO_DEVICE and b"no HIP device" in L.lib().lsm_last_error(None)


def test_create_validates_arguments(lsm):
    """Argument validation happens before any device call, so it is testable without a GPU."""
    L = lsm._lib
    from lsm_amd.api import _bc_c, _normalize_bc
    h = C.c_void_p()
    g = lsm.CartesianGrid((0, 0), (1, 1), (16, 3))._c()      # < 4 nodes
    assert L.lib().lsm_create(C.byref(g), _bc_c(_normalize_bc(lsm.NeumannBC(), 2), 2), None, 0, 0, 0, C.byref(h)) == L.ERR_INVALID
    g = lsm.CartesianGrid((0, 0), (1, 1), (16, 16))._c()
    bad = _bc_c(_normalize_bc(lsm.NeumannBC(), 2), 2)
    bad[0][0].kind = L.BC_PERIODIC                            # periodic on one face only
    assert L.lib().lsm_create(C.byref(g), bad, None, 0, 0, 0, C.byref(h)) == L.ERR_INVALID
    assert b"periodic" in L.lib().lsm_last_error(None)
    assert L.lib().lsm_create(C.byref(g), _bc_c(_normalize_bc(lsm.NeumannBC(), 2), 2), None, 7, 0, 0, C.byref(h)) == L.ERR_INVALID  # neither LSM_DTYPE_F64 nor LSM_DTYPE_F32


def test_normalize_bc(lsm):
    """test/test-boundaryconditions.jl:4-22"""
    from lsm_amd.api import _normalize_bc
    P, N = lsm.PeriodicBC, lsm.NeumannBC
    r = _normalize_bc(P(), 2)
    assert all(isinstance(b, P) for pair in r for b in pair)
    r = _normalize_bc((P(), N()), 2)
    assert isinstance(r[0][0], P) and r[1][0].degree == 0 and r[1][1].degree == 0
    ebc = lsm.ExtrapolationBC(2)
    r = _normalize_bc([P(), (ebc, N())], 2)
    assert r[1][0] is ebc and r[1][1].degree == 0
    with pytest.raises(ValueError):
        _normalize_bc([(P(), ebc), (ebc, N())], 2)
    with pytest.raises(ValueError):
        _normalize_bc([P()], 2)
    with pytest.raises(ValueError):
        lsm.ExtrapolationBC(-1)


def test_grid_geometry(lsm):
    """test/test-meshes.jl: corner nodes are exact; spacing (hc-lc)/(n-1)."""
    g = lsm.CartesianGrid((-1, 0), (1, 3), (5, 7))
    assert g.getnode((0, 0)) == (-1.0, 0.0)
    assert g.getnode((4, 6)) == (1.0, 3.0)
    assert g.meshsize() == (0.5, 0.5)
    with pytest.raises(ValueError):
        g.getnode((5, 0))
    with pytest.raises(ValueError):
        lsm.CartesianGrid((0,), (1, 1), (3, 3))


@pytest.mark.parametrize("bcname", ["periodic", "neumann", "symmetry", "extrap3"])
def test_meshfield_getindex_matches_oracle(lsm, orc, bcname):
    """The host MeshField's ghost resolution (src/meshfield.jl:213-260) against the C oracle."""
    rng = np.random.default_rng(0)
    shape = (7, 6, 5)
    vals = rng.standard_normal(shape)
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), shape)
    bc = {"periodic": lsm.PeriodicBC(), "neumann": lsm.NeumannBC(), "symmetry": lsm.SymmetryBC(), "extrap3": lsm.ExtrapolationBC(3)}[bcname]
    obc = orc.make_bc({"periodic": "periodic", "neumann": "neumann", "symmetry": "symmetry", "extrap3": ("extrapolation", 3)}[bcname], 3)
    mf = lsm.MeshField(vals, grid, bc=bc)
    og = orc.Grid((0, 0, 0), (1, 1, 1), shape)
    for I in [(-1, 2, 2), (7, 2, 2), (-2, -1, 3), (8, 7, -3), (3, 3, 3), (-3, 6, 5)]:
        assert mf[I] == orc.get(og, obc, np.asfortranarray(vals), I)
    with pytest.raises(ValueError):
        lsm.MeshField(vals, grid)[(-1, 0, 0)]


def test_terms_and_equation_argument_checks(lsm):
    grid = lsm.CartesianGrid((0, 0), (1, 1), (8, 8))
    ic = lsm.MeshField(lambda x: x[0], grid, bc=lsm.NeumannBC())
    with pytest.raises(ValueError, match="terms must be"):
        lsm.LevelSetEquation(terms=[1, 2], ic=ic)
    with pytest.raises(ValueError):
        lsm.MeshField(np.zeros((3, 3)), grid)
    assert repr(lsm.RK3()) == "RK3 (3rd order TVD Runge-Kutta)\n  └─ cfl: 0.5"       # test/test-show.jl integrator strings
    assert repr(lsm.ForwardEuler(cfl=0.25)).endswith("cfl: 0.25")
    assert repr(lsm.AdvectionTerm((1.0, 0.0))) == "𝐮 ⋅ ∇ ϕ" and repr(lsm.CurvatureTerm(1.0)) == "b κ|∇ϕ|"
    assert repr(lsm.EikonalReinitializationTerm()) == "sign(ϕ) (|∇ϕ| - 1)"



def test_set_operations_on_level_sets():
    """src/levelsetops.jl:246-325 and docs/src/example-zalesak.md:21-26: union = min, intersection = max, complement = -ϕ,
    setdiff = max(ϕ₁, -ϕ₂); the in-place forms mutate and return their first argument."""
    import lsm_amd as lsm
    grid = lsm.CartesianGrid((-1.5, -1.5), (1.5, 1.5), (61, 61))
    disk = lsm.MeshField(lambda x: np.hypot(x[0] + 0.75, x[1]) - 0.5, grid)
    rec = lsm.MeshField(lambda x: np.maximum(np.abs(x[0] + 0.75) - 0.1, np.abs(x[1] + 0.25) - 0.5), grid)
    zal = disk.setdiff(rec)
    assert np.array_equal(zal.vals, np.maximum(disk.vals, -rec.vals)) and zal is not disk
    assert np.array_equal(disk.union(rec).vals, np.minimum(disk.vals, rec.vals))
    assert np.array_equal((disk & rec).vals, np.maximum(disk.vals, rec.vals))
    assert np.array_equal((-disk).vals, -disk.vals) and np.array_equal((disk - rec).vals, zal.vals)
    keep = disk.vals.copy()
    assert disk.union_(rec) is disk and np.array_equal(disk.vals, np.minimum(keep, rec.vals))
    assert disk.vals.flags.f_contiguous
    with pytest.raises(ValueError):
        disk.union(lsm.MeshField(lambda x: x[0], lsm.CartesianGrid((-1, -1), (1, 1), (11, 11))))
