"""The CPU oracle against golden vectors produced by the reference's own kernels (not gpu)."""
import numpy as np
import pytest

from conftest import assert_close_scaled, vm_inputs

# The oracle repeats the reference's statements in the same order; what is left is BLAS-internal
# summation order inside np.dot / `@` (a few ulp).
ORACLE_RTOL = 2e-15


@pytest.mark.parametrize("name,d", [("von_mises_d4.npz", 4), ("von_mises_d6.npz", 6)])
def test_von_mises_oracle_matches_reference_golden(oracle, golden, name, d):
    g = golden(name)
    E, nu, sigma_0, H = g["params"]
    C, s, dp = oracle.von_mises(g["deps"], g["sigma_n"], g["p"], E=E, nu=nu, sigma_0=sigma_0, H=H)
    assert g["deps"].shape[-1] == d
    assert_close_scaled(C, g["C_tang"], ORACLE_RTOL, "C_tang")
    assert_close_scaled(s, g["sigma"], ORACLE_RTOL, "sigma")
    assert_close_scaled(dp, g["dp"], ORACLE_RTOL, "dp")
    # the all-zero special point is the reference's 0/0 case (demo_plasticity_von_mises.py:318-319)
    assert np.isnan(C[4]).all() and np.isnan(s[4]).all() and dp[4] == 0.0


@pytest.mark.parametrize("d", [4, 6])
def test_von_mises_oracle_elastic_points_return_c_elas(oracle, golden, d):
    g = golden(f"von_mises_d{d}.npz")
    C, s, dp = oracle.von_mises(g["deps"], g["sigma_n"], g["p"])
    elastic = (dp == 0.0) & np.isfinite(s).all(axis=1)
    assert elastic.sum() > 10
    assert np.array_equal(C[elastic], np.broadcast_to(g["C_elas"], C[elastic].shape))


@pytest.mark.parametrize("d", [4, 6])
def test_von_mises_oracle_threads_agree(oracle, d):
    deps, sigma_n, p = vm_inputs(5000, d, seed=7)
    a = oracle.von_mises(deps, sigma_n, p, nthreads=1)
    b = oracle.von_mises(deps, sigma_n, p, nthreads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("d", [4, 6])
def test_von_mises_oracle_plastic_points_land_on_yield_surface(oracle, d):
    """Independent check with the pure-UFL statement of the same model
    (demo_plasticity_von_mises_pure_ufl.py:105-124): f(sigma_new, p + dp) = 0 on plastic points."""
    deps, sigma_n, p = vm_inputs(20000, d, seed=11)
    C, s, dp = oracle.von_mises(deps, sigma_n, p)
    E, sigma_0 = 70e3, 250.0
    H = E * (E / 100.0) / (E - E / 100.0)
    dev = s.copy()
    dev[:, :3] -= s[:, :3].mean(axis=1, keepdims=True)
    seq = np.sqrt(1.5 * np.sum(dev * dev, axis=1))
    plastic = dp > 0
    assert plastic.mean() > 0.5 and (~plastic).sum() > 100
    f = seq - sigma_0 - H * (p + dp)
    assert np.max(np.abs(f[plastic])) < 1e-9 * sigma_0
    assert np.all(f[~plastic] <= 1e-9)
    # tangent is symmetric
    assert np.max(np.abs(C - np.transpose(C, (0, 2, 1)))) < 1e-9 * E


def test_heat_oracle_matches_reference_golden_bitwise(oracle, golden):
    g = golden("heat_c1.npz")
    q, dqdT, dqds = oracle.heat(g["T"], g["sigma"], A=float(g["A"]), B=float(g["B"]), gdim=2)
    assert g["T"].size == 6144  # BASELINE config 1: 32x32 unit square, 2048 triangles x 3 points
    assert np.array_equal(q.reshape(-1), g["q"])
    assert np.array_equal(dqdT.reshape(-1), g["dqdT"])
    assert np.array_equal(dqds.reshape(-1), g["dqdsigma"])
    # -k * 0 is a negative zero in the reference (part2.py:260); the oracle keeps the sign
    assert np.array_equal(np.signbit(dqds.reshape(-1)), np.signbit(g["dqdsigma"]))


def test_von_mises_load_history_golden(oracle, golden):
    """Six load steps of the reference's return_mapping with the demo's state update (p += dp, sigma_n = sigma,
    demo_plasticity_von_mises.py:564-565) in between: the oracle, driven by its OWN accumulated state, has to stay
    on the reference's trajectory (errors would compound across steps)."""
    g = golden("von_mises_history_d4.npz")
    n_steps = int(g["n_steps"])
    d = 4
    sigma_n = np.zeros_like(g["sigma_0"]).reshape(-1, d)
    p = np.zeros(sigma_n.shape[0])
    for k in range(n_steps):
        C, s, dp = oracle.von_mises(g[f"deps_{k}"].reshape(-1, d), sigma_n, p)
        assert_close_scaled(C, g[f"C_tang_{k}"], 1e-13, f"C_tang step {k}")
        assert_close_scaled(s, g[f"sigma_{k}"], 1e-13, f"sigma step {k}")
        assert_close_scaled(dp, g[f"dp_{k}"], 1e-13, f"dp step {k}")
        p = p + dp.reshape(-1)
        sigma_n = s.reshape(-1, d).copy()
        assert_close_scaled(p, g[f"p_after_{k}"], 1e-13, f"p after step {k}")
    assert (g["dp_2"] > 0).mean() > 0.5 and (g["dp_3"] > 0).sum() == 0       # loading, then elastic unloading


def test_conductivity_oracle_matches_reference_golden(oracle, golden):
    """k_impl / dkdT_impl of the part-1 heat demo executed by the generator (demo_nonlinear_heat_equation_part1.py:251-271):
    the C restatement is bit-identical, including the pole A + B T = 0 (inf) and huge |T|."""
    g = golden("conductivity_p1.npz")
    k, dk = oracle.conductivity(g["T"], A=float(g["A"]), B=float(g["B"]))
    assert np.array_equal(k, g["k"]) and np.array_equal(dk, g["dkdT"])
    with np.errstate(all="ignore"):
        k, dk = oracle.conductivity(g["T_rand"], A=float(g["A"]), B=float(g["B"]))
    assert np.array_equal(k, g["k_rand"], equal_nan=True) and np.array_equal(dk, g["dkdT_rand"], equal_nan=True)


def test_icnn_c_port_matches_golden_and_numpy_oracle(oracle, golden):
    """oracle/icnn_oracle_c.c (per-point jets, OpenMP) — the compiled form of icnn_oracle.py that bench.py times as the CPU baseline
    of BASELINE config 5 — against the golden produced by the reference's own classes under torch and against the NumPy oracle."""
    from oracle.icnn_oracle import h_correction, icnn_stress_tangent

    g = golden("icnn_isihara.npz")
    w = dict(golden("icnn_isihara_weights.npz"))
    dP, P, H = oracle.icnn(g["F"], w, nthreads=4)
    sP, sdP = np.abs(g["P"]).max(), np.abs(g["dP"]).max()
    assert np.abs(P - g["P"]).max() <= 2e-6 * sP and np.abs(dP - g["dP"]).max() <= 2e-6 * sdP      # the GPU parity tolerance
    dPn, Pn = icnn_stress_tangent(g["F"], w)
    assert np.abs(P - Pn).max() <= 5e-7 * sP and np.abs(dP - dPn).max() <= 5e-7 * sdP              # same jets, other summation order
    # H = -P_NN(I) is fp32 rounding noise of the reference's evaluation at the identity (2^-22 on its diagonal); both restatements get 0
    assert np.allclose(H, h_correction(w), rtol=0, atol=1e-9) and np.allclose(H, g["H"], rtol=0, atol=1e-6)
    one = oracle.icnn(g["F"], w, nthreads=1)
    assert np.array_equal(one[0], dP) and np.array_equal(one[1], P)                                  # threads do not change the result


@pytest.mark.parametrize("cell,n", [("triangle", (6, 5)), ("quadrilateral", (4, 4)), ("tetrahedron", (2, 3, 2)), ("hexahedron", (3, 2, 3))])
@pytest.mark.parametrize("degree", [1, 2])
def test_compiled_consumer_oracle_equals_the_numpy_one(cell, n, degree):
    """oracle/operand_oracle_c.c (the threaded CPU baseline of the device-resident Newton iteration) against oracle/operand_oracle.py:
    strain at the points, internal force, matrix-free tangent action — same sums in another order."""
    from oracle import load_oracle
    from oracle.operand_oracle import EPS_MANDEL, eval_operand, operand_adjoint, tangent_apply
    from tools.synthetic import structured_mesh

    o = load_oracle()
    m = structured_mesh(cell, n, degree, distort=0.2, seed=6)
    G, nn = m.gdim, m.node_x.shape[0]
    d = 4 if G == 2 else 6
    rng = np.random.Generator(np.random.PCG64(1))
    u = rng.normal(size=nn * G)
    args = (m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
    for nt in (1, 3):
        e, e_ref = o.operand_eps(m, u, nthreads=nt), eval_operand(EPS_MANDEL, G, u, *args)
        assert np.abs(e - e_ref).max() <= 1e-13 * np.abs(e_ref).max()
        S = rng.normal(size=e.shape)
        f, f_ref = o.operand_eps_adjoint(m, S, nn, nthreads=nt), operand_adjoint(EPS_MANDEL, G, S, m.weights, *args, nn)
        assert np.abs(f - f_ref).max() <= 1e-13 * np.abs(f_ref).max()
        A = rng.normal(size=(e.shape[0], e.shape[1], d, d))
        Ct = A @ A.transpose(0, 1, 3, 2)
        k, k_ref = o.tangent_apply(m, Ct, u, nn, nthreads=nt), tangent_apply(Ct, u, m.weights, *args, nn)
        assert np.abs(k - k_ref).max() <= 1e-13 * np.abs(k_ref).max()
    # a prefix of the cells: what the bench's bounded sample uses
    half = m.num_cells // 2
    e_half = o.operand_eps(m, u, cells=half, nthreads=2)
    assert np.array_equal(e_half, o.operand_eps(m, u, nthreads=1)[:half])


def test_compiled_isihara_oracle_equals_the_numpy_one_and_the_golden(golden):
    """oracle/icnn_oracle_c.c::oracle_isihara (the threaded CPU baseline of the Isihara leg) against icnn_oracle.isihara_stress_tangent and
    against the torch-differentiated golden (tests/golden/isihara_analytic.npz)."""
    from oracle import load_oracle
    from oracle.icnn_oracle import isihara_stress_tangent

    o = load_oracle()
    g = golden("isihara_analytic.npz")
    dP, P = o.isihara(g["F"], nthreads=2)
    assert np.abs(P - g["P"].reshape(P.shape)).max() <= 1e-12 * np.abs(g["P"]).max()
    assert np.abs(dP - g["dP"].reshape(dP.shape)).max() <= 1e-11 * np.abs(g["dP"]).max()
    rng = np.random.Generator(np.random.PCG64(3))
    F = rng.normal(size=(3000, 4)) * 0.1 + np.array([1.0, 0.0, 0.0, 1.0])
    F[7] = [1.0, 2.0, 3.0, 4.0]                       # det F < 0: NaN in both
    dP, P = o.isihara(F, nthreads=3)
    dPr, Pr = isihara_stress_tangent(F)
    ok = ~np.isnan(Pr[:, 0])
    assert np.isnan(P[~ok]).all() and np.isnan(dP[~ok]).all() and (~ok).sum() == 1
    assert np.abs(P[ok] - Pr[ok]).max() <= 1e-13 * np.abs(Pr[ok]).max() and np.abs(dP[ok] - dPr[ok]).max() <= 1e-13 * np.abs(dPr[ok]).max()
