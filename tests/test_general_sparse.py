"""SURVEY §8 (f)2 — general sparse, complex operators on the device (csrc/csr.hip behind multigridcmt_amd/general.py):
the reference's own vcycle / smoothers run on its k.p Hamiltonian (ThesisProblem.py:26-40,62-101; fixtures written by
oracle/gen_golden.py from PotWellSolver.makeMatrix through the loader) against the HIP path and against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, rel_err
from multigridcmt_amd import MGCMTSolver, MGCMTStencilMaker, _lib
from oracle.sparse_ref import RefSolver, RefStencilMaker

TAGS = ("n64_k0.5", "n128_k0")


def _matrix(gold, tag):
    n = len(gold[tag + "_indptr"]) - 1
    return sp.csr_matrix((gold[tag + "_data"], gold[tag + "_indices"], gold[tag + "_indptr"]), shape=(n, n))


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_reference_on_kp_hamiltonian(tag):
    """The oracle's generic-sparse restatement, complex numbers included, against the reference's outputs."""
    gold = load_golden("kp_well")
    A, f, shift = _matrix(gold, tag), gold[tag + "_f"], float(gold[tag + "_shift"])
    ref, rsm = RefSolver(), RefStencilMaker()
    N = A.shape[0]
    for name, smo in (("gs", ref.gseidel), ("wj", ref.wjacobi), ("sor", lambda v, f, A, nu=4: ref.sor(v, f, A, nu=nu, omega=1.2))):
        y = ref.vcycle(np.zeros(N, dtype=complex), f, A, rsm, shift=shift, lowest_level=32, smoother=smo)
        assert rel_err(y, gold["%s_vcycle_%s" % (tag, name)]) < 1e-10, name
    assert rel_err(ref.gseidel(0.5 * f, f, A, nu=3), gold[tag + "_gseidel3"]) < 1e-12


@pytest.mark.parametrize("tag", TAGS)
def test_device_galerkin_product(backend, tag):
    """R*A*P (MGCMTSolver.py:318) computed on the device equals the reference's sparse triple product."""
    from multigridcmt_amd.general import CsrPlan
    gold = load_golden("kp_well")
    A = _matrix(gold, tag)
    plan = CsrPlan(A, 32)
    try:
        assert plan.num_levels == int(np.log2(A.shape[0] // 32)) + 1
        C = plan.matrix(1).toarray()
        want = gold[tag + "_rap_dense"]
        assert np.abs(C - want).max() <= 1e-13 * np.abs(want).max()
        n, nnz, chunk = plan.level_info(0)
        assert n == A.shape[0] and nnz == A.nnz and chunk >= A.shape[0] // 4        # at least one chunk per band block
        c1 = plan.level_info(1)[2]
        assert c1 >= 1 and c1 & (c1 - 1) == 0                                       # (the Galerkin blocks couple across their edges)
    finally:
        plan.close()


@pytest.mark.parametrize("tag", TAGS)
def test_kp_vcycle_matches_reference(backend, tag):
    """ThesisProblem.py:101: solver.vcycle(w, v, hamiltonian_sparse, stencil_maker, shift=guess, lowest_level=2**5,
    smoother=solver.gseidel) on the complex 4n x 4n matrix — through the drop-in class, on the device."""
    gold = load_golden("kp_well")
    A, f, shift = _matrix(gold, tag), gold[tag + "_f"], float(gold[tag + "_shift"])
    solver, sm = MGCMTSolver(), MGCMTStencilMaker()
    N = A.shape[0]
    for name, smo in (("gs", solver.gseidel), ("wj", solver.wjacobi), ("sor", __import__("functools").partial(solver.sor, omega=1.2))):
        x = solver.vcycle(np.zeros(N, dtype=complex), f.copy(), sp.csc_matrix(A), sm, shift=shift, lowest_level=2 ** 5, smoother=smo)
        assert x.shape == (N,) and np.iscomplexobj(x)
        assert rel_err(x, gold["%s_vcycle_%s" % (tag, name)]) < 1e-10, name
    x = solver.gseidel((0.5 * f).reshape(N, 1), f.copy().reshape(N, 1), A, nu=3)
    assert x.shape == (N, 1) and rel_err(x.ravel(), gold[tag + "_gseidel3"]) < 1e-12
    x = solver.wjacobi(0.5 * f, f.copy(), A, nu=3)
    assert rel_err(x.ravel(), gold[tag + "_wjacobi3"]) < 1e-12


def test_general_path_on_unstructured_real_matrix(backend):
    """A real matrix recognise() cannot map (1-D Laplacian plus a few long-range couplings: the sequential chunk of the
    lexicographic sweep) still cycles on the device; checked against the oracle; a 2-D cycle of such a matrix raises."""
    from multigridcmt_amd.operators import UnrecognisedOperator
    solver, sm = MGCMTSolver(), MGCMTStencilMaker()
    ref, rsm = RefSolver(), RefStencilMaker()
    n = 128
    rng = np.random.RandomState(5)
    A = sp.lil_matrix(((-1 / np.pi ** 2) * sm.laplacian(n)).toarray())
    for _ in range(40):
        i, j = rng.randint(0, n, 2)
        if abs(i - j) > 1:
            A[i, j] = A[j, i] = 0.05 * rng.rand()
    A = sp.csr_matrix(A)
    f = rng.rand(n)
    for smo, rsmo in ((solver.gseidel, ref.gseidel), (solver.wjacobi, ref.wjacobi)):
        x = solver.vcycle(np.zeros(n), f.copy(), A, sm, shift=0.2, lowest_level=16, smoother=smo)
        y = ref.vcycle(np.zeros(n), f, A, rsm, shift=0.2, lowest_level=16, smoother=rsmo)
        assert not np.iscomplexobj(x) and rel_err(x, y) < 1e-11
    with pytest.raises(UnrecognisedOperator):
        solver.vcycle(np.zeros(64), np.ones(64), sp.random(64, 64, density=0.2, random_state=0) + sp.eye(64), sm, dimension="2d", lowest_level=4)
