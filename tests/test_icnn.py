"""ICNN hyperelastic operator: oracle vs reference goldens (CPU) and HIP kernel vs both (gpu)."""
import numpy as np
import pytest

from oracle.icnn_oracle import features, icnn_stress_tangent

DEFAULT_VARIANT = 2     # ctx option icnn_variant as the library ships it (restored by the tests that switch kernels)

# The reference runs the network in fp32 (`.float()`, demo_hyperelasticity.py:286); two correct fp32
# implementations differ by summation order. Tolerances are relative to max|dP| resp. max|P| of the batch.
RTOL_FP32 = 2e-6


@pytest.fixture(scope="module")
def weights(golden):
    return dict(golden("icnn_isihara_weights.npz"))


def relerr(a, b):
    return np.max(np.abs(np.asarray(a).reshape(-1) - np.asarray(b).reshape(-1))) / np.max(np.abs(b))


def test_weight_fixture_is_the_reference_state_dict(weights):
    assert sum(v.size for v in weights.values()) == 9027          # 9 027 fp32 parameters
    assert weights["layers__1__weights"].shape == (64, 64) and weights["layers__0__weight"].shape == (64, 3)
    assert all(v.dtype == np.float32 for v in weights.values())


def test_oracle_matches_reference_golden(golden, weights):
    g = golden("icnn_isihara.npz")
    dP, P = icnn_stress_tangent(g["F"], weights)
    assert dP.dtype == np.float64
    assert relerr(dP, g["dP"]) <= RTOL_FP32 and relerr(P, g["P"]) <= RTOL_FP32
    dP32, P32 = icnn_stress_tangent(g["F"].astype(np.float32), weights)     # dtype follows the input (:452-456)
    assert dP32.dtype == np.float32
    assert relerr(dP32, g["dP_f32in"]) <= 5e-6 and relerr(P32, g["P_f32in"]) <= 5e-6


def test_feature_derivatives_by_finite_differences():
    rng = np.random.default_rng(0)
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.15 * rng.normal(size=(50, 4))
    K, dK, d2K = features(F)
    h = 1e-6
    for j in range(4):
        Fp, Fm = F.copy(), F.copy()
        Fp[:, j] += h
        Fm[:, j] -= h
        Kp, dKp, _ = features(Fp)
        Km, dKm, _ = features(Fm)
        assert np.max(np.abs((Kp - Km) / (2 * h) - dK[:, :, j])) < 1e-7
        assert np.max(np.abs((dKp - dKm) / (2 * h) - d2K[:, :, :, j])) < 1e-6


def test_stress_vanishes_in_the_undeformed_state(golden, weights):
    # what the H correction is for (:362-381): P(F = I) = 0 up to fp32 noise
    dP, P = icnn_stress_tangent(np.array([[1.0, 0.0, 0.0, 1.0]]), weights)
    assert np.max(np.abs(P)) < 1e-6
    assert np.allclose(dP[0], dP[0].T, atol=1e-5)          # hyperelastic tangent is symmetric


# ------------------------------------------------------------------------------------------ GPU
def state_dict(weights):
    return {k.replace("__", "."): v for k, v in weights.items()}


@pytest.mark.gpu
def test_hip_matches_reference_golden(ctx, golden, weights):
    from dolfinx_external_operator_amd import make_icnn

    g = golden("icnn_isihara.npz")
    ext = make_icnn(state_dict(weights), ctx=ctx)
    dP, P = ext((1,))(g["F"].reshape(-1, 1, 2, 2))
    assert dP.shape == (g["F"].shape[0] * 16,) and P.shape == (g["F"].shape[0] * 4,)
    assert relerr(dP, g["dP"]) <= RTOL_FP32 and relerr(P, g["P"]) <= RTOL_FP32
    assert np.max(np.abs(ext.correction() - np.array([g["H"][0, 0], g["H"][0, 1], g["H"][1, 0], g["H"][1, 1]]))) < 1e-6
    with pytest.raises(NotImplementedError, match="No external function is defined"):
        ext((0,))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 257, 5000])
def test_hip_against_oracle_sizes(ctx, weights, n):
    from dolfinx_external_operator_amd import MEM_HOST

    rng = np.random.default_rng(n)
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    model = ctx.icnn_create(state_dict(weights))
    try:
        dP, P = np.full(n * 16 + 8, -7.0), np.full(n * 4 + 8, -7.0)
        ctx.icnn_eval(model, 0, n, MEM_HOST, F, dP, P)
        assert np.all(dP[n * 16:] == -7.0) and np.all(P[n * 4:] == -7.0)
        if n:
            dPo, Po = icnn_stress_tangent(F, weights)
            assert relerr(dP[: n * 16], dPo) <= RTOL_FP32 and relerr(P[: n * 4], Po) <= RTOL_FP32
    finally:
        ctx.icnn_destroy(model)


@pytest.mark.gpu
def test_mfma_and_valu_kernels_agree(ctx, weights):
    """fp32 network: the two MFMA kernels (variant 1 fp32-input MFMA; variant 2, the default, split-bf16 MFMA) against the
    lane-per-point VALU kernel (variant 0) and the oracle.
    Summation orders differ between the kernels, so they agree to fp32 rounding, not bit for bit."""
    from dolfinx_external_operator_amd import MEM_HOST

    rng = np.random.default_rng(11)
    n = 20_001     # ragged: last wave has one live lane
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    model = ctx.icnn_create(state_dict(weights))
    out = {}
    try:
        for variant in (0, 1, 2):
            ctx.set_option("icnn_variant", variant)
            dP, P = np.full(n * 16 + 4, -7.0), np.full(n * 4 + 4, -7.0)
            ctx.icnn_eval(model, 0, n, MEM_HOST, F, dP, P)
            assert np.all(dP[n * 16:] == -7.0) and np.all(P[n * 4:] == -7.0)
            out[variant] = (dP[: n * 16], P[: n * 4])
    finally:
        ctx.set_option("icnn_variant", DEFAULT_VARIANT)
        ctx.icnn_destroy(model)
    dPo, Po = icnn_stress_tangent(F, weights)
    for variant in (1, 2):
        assert relerr(out[variant][0], out[0][0]) <= RTOL_FP32 and relerr(out[variant][1], out[0][1]) <= RTOL_FP32
    for variant in (0, 1, 2):
        assert relerr(out[variant][0], dPo) <= RTOL_FP32 and relerr(out[variant][1], Po) <= RTOL_FP32


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 129, 20_001, 300_000])
def test_pipelined_and_hybrid_kernels_are_bit_identical_to_the_packed_one(ctx, weights, n, experiments_build):
    """Experiments build only (scripts/exp/icnn_variants.h, -DDXO_EXPERIMENTS; the product library refuses the options, checked here).
    icnn_variant 3 (scalar fp32 arithmetic, phase 1 of the next half-tile issued between the vector instructions of this one's
    phases 2, one wave per SIMD) and 4 (scalar operands inside phase 1's MFMAs, packed phases 2 and 3) run the same operations in
    the same order per accumulator as variant 2 (packed fp32, phases in sequence): same bits, at sizes with one tile per wave,
    ragged tails and several tiles per wave."""
    from dolfinx_external_operator_amd import MEM_HOST

    if not experiments_build:
        for variant in (3, 4):
            with pytest.raises(ValueError, match="icnn_variant"):
                ctx.set_option("icnn_variant", variant)
        assert ctx.get_option("icnn_variant") == DEFAULT_VARIANT
        return
    rng = np.random.default_rng(21)
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    model = ctx.icnn_create(state_dict(weights))
    out = {}
    try:
        for variant in (2, 3, 4):
            ctx.set_option("icnn_variant", variant)
            dP, P = np.full(n * 16 + 4, -7.0), np.full(n * 4 + 4, -7.0)
            ctx.icnn_eval(model, 0, n, MEM_HOST, F, dP, P)
            assert np.all(dP[n * 16:] == -7.0) and np.all(P[n * 4:] == -7.0)
            out[variant] = (dP, P)
    finally:
        ctx.set_option("icnn_variant", DEFAULT_VARIANT)
        ctx.icnn_destroy(model)
    for variant in (3, 4):
        assert np.array_equal(out[2][0], out[variant][0]) and np.array_equal(out[2][1], out[variant][1]), variant
    assert np.all(np.isfinite(out[4][0]))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_non_finite_points_stay_in_their_own_rows(ctx, weights, variant, experiments_build):
    """The MFMA kernels evaluate 64 points per wave through shared matrix products: a point is one COLUMN of every product, so
    a NaN / inf / singular deformation gradient must poison its own 20 outputs and nothing else — the other points come out
    bit for bit as in a batch where the bad points are replaced by the identity."""
    from dolfinx_external_operator_amd import MEM_HOST

    if variant > 2 and not experiments_build:      # scripts/exp/icnn_variants.h: the product library refuses these values
        with pytest.raises(ValueError, match="icnn_variant"):
            ctx.set_option("icnn_variant", variant)
        return
    rng = np.random.default_rng(5)
    n = 1000
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    bad = {3: [np.nan, 0.0, 0.0, 1.0], 64: [np.inf, 0.0, 0.0, 1.0], 130: [1.0, 1.0, 1.0, 1.0],      # det F = 0
           511: [0.0, 0.0, 0.0, 0.0], 999: [1e200, 0.0, 0.0, 1e200]}
    Fb, Fc = F.copy(), F.copy()
    for i, v in bad.items():
        Fb[i], Fc[i] = v, [1.0, 0.0, 0.0, 1.0]
    model = ctx.icnn_create(state_dict(weights))
    try:
        ctx.set_option("icnn_variant", variant)
        out = []
        for Fx in (Fb, Fc):
            dP, P = np.zeros(n * 16), np.zeros(n * 4)
            ctx.icnn_eval(model, 0, n, MEM_HOST, Fx, dP, P)
            out.append((dP.reshape(n, 16), P.reshape(n, 4)))
    finally:
        ctx.set_option("icnn_variant", DEFAULT_VARIANT)
        ctx.icnn_destroy(model)
    good = np.ones(n, dtype=bool)
    good[list(bad)] = False
    assert np.array_equal(out[0][0][good], out[1][0][good]) and np.array_equal(out[0][1][good], out[1][1][good])
    assert np.all(np.isfinite(out[1][0])) and np.all(np.isfinite(out[1][1]))
    for i in bad:
        assert not np.all(np.isfinite(out[0][0][i])) or not np.all(np.isfinite(out[0][1][i])), i


@pytest.mark.gpu
def test_large_and_inverted_deformations(ctx, weights):
    """Stretches of 0.3 ... 3, shear, and det F < 0 (the features use |det F| and its sign, demo_hyperelasticity.py:263-283): the
    default kernel's |det F|^(-2/3) comes from a seed and two Newton steps, the other kernels' from pow."""
    from dolfinx_external_operator_amd import MEM_HOST

    rng = np.random.default_rng(8)
    n = 4096
    F = rng.uniform(-1.5, 1.5, size=(n, 4))
    F[:, 0] += np.where(rng.random(n) < 0.5, 1.5, -1.5)
    F[:, 3] += 1.5
    det = F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]
    F = F[np.abs(det) > 0.05]
    n = F.shape[0]
    assert (F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2] < 0).sum() > n // 5
    model = ctx.icnn_create(state_dict(weights))
    try:
        dP, P = np.zeros(n * 16), np.zeros(n * 4)
        ctx.icnn_eval(model, 0, n, MEM_HOST, F, dP, P)
    finally:
        ctx.icnn_destroy(model)
    dPo, Po = icnn_stress_tangent(F, weights)
    # per point: the outputs span many decades over this batch, a batch-wide scale would hide the small ones
    sd = np.abs(dPo).reshape(n, -1).max(axis=1)[:, None] + 1e-30
    sp = np.abs(Po).reshape(n, -1).max(axis=1)[:, None] + 1e-30
    assert np.max(np.abs(dP.reshape(n, -1) - dPo.reshape(n, -1)) / sd) <= 2e-5
    assert np.max(np.abs(P.reshape(n, -1) - Po.reshape(n, -1)) / sp) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed, scale, rtol", [(1, 1.0, RTOL_FP32), (2, 0.2, RTOL_FP32), (3, 3.0, 1e-5)])
def test_kernels_with_random_weights(ctx, weights, seed, scale, rtol):
    """Other networks than the trained one: weights of the same architecture drawn at random (a fifth, once and three times the
    trained magnitudes; the softplus re-parametrisation turns them into very differently scaled positive matrices). The split
    of the GEMM operands into three bf16 parts is exact whatever the magnitudes are, so the default kernel must follow the
    oracle as closely as the fp32-input kernel does. (With three times the magnitudes the outputs reach 1e6 and two correct fp32
    evaluations differ by up to 3e-6 of that — the lane-per-point kernel and the oracle do — hence the wider bound there.)"""
    from dolfinx_external_operator_amd import MEM_HOST

    rng = np.random.default_rng(seed)
    w = {k: (scale * np.abs(v).max() * rng.uniform(-1.0, 1.0, size=v.shape)).astype(np.float32) for k, v in weights.items()}
    n = 3000
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.08 * rng.normal(size=(n, 4))
    model = ctx.icnn_create(state_dict(w))
    out = {}
    try:
        for variant in (0, 1, 2):
            ctx.set_option("icnn_variant", variant)
            dP, P = np.zeros(n * 16), np.zeros(n * 4)
            ctx.icnn_eval(model, 0, n, MEM_HOST, F, dP, P)
            out[variant] = (dP, P)
    finally:
        ctx.set_option("icnn_variant", DEFAULT_VARIANT)
        ctx.icnn_destroy(model)
    dPo, Po = icnn_stress_tangent(F, w)
    assert np.all(np.isfinite(dPo)) and np.max(np.abs(dPo)) > 0
    for variant in (0, 1, 2):
        assert relerr(out[variant][0], dPo) <= rtol and relerr(out[variant][1], Po) <= rtol, variant
    assert relerr(out[2][0], out[1][0]) <= rtol and relerr(out[2][1], out[1][1]) <= rtol      # the two MFMA kernels against each other


@pytest.mark.gpu
def test_fp64_network_variant_tolerance_study(ctx, weights):
    """BASELINE config 5: the fp64 network differs from the fp32 one only by fp32 rounding (~1e-7), and
    matches the fp64-network oracle to fp64 accuracy."""
    from dolfinx_external_operator_amd import make_icnn

    rng = np.random.default_rng(5)
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(4000, 4))
    dP32, P32 = make_icnn(state_dict(weights), ctx=ctx, precision="fp32")((1,))(F)
    dP64, P64 = make_icnn(state_dict(weights), ctx=ctx, precision="fp64")((1,))(F)
    dPo, Po = icnn_stress_tangent(F, weights, net_dtype=np.float64)
    assert relerr(dP64, dPo) <= 1e-11 and relerr(P64, Po) <= 1e-11
    assert 1e-9 < relerr(dP32, dP64) <= RTOL_FP32


@pytest.mark.gpu
def test_device_pointers_and_validation(ctx, weights):
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE

    n = 10_000
    rng = np.random.default_rng(1)
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    dev = torch.device("cuda:0")
    Ft = torch.from_numpy(F).to(dev)
    dP = torch.empty(n * 16, dtype=torch.float64, device=dev)
    P = torch.empty(n * 4, dtype=torch.float64, device=dev)
    model = ctx.icnn_create(state_dict(weights))
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.icnn_eval(model, 0, n, MEM_DEVICE, Ft.data_ptr(), dP.data_ptr(), P.data_ptr())
        torch.cuda.synchronize()
        dPo, Po = icnn_stress_tangent(F, weights)
        assert relerr(dP.cpu().numpy(), dPo) <= RTOL_FP32
        with pytest.raises(ValueError, match="OPTION"):
            ctx.icnn_eval(model, 2, n, MEM_DEVICE, Ft.data_ptr(), dP.data_ptr(), P.data_ptr())
    finally:
        ctx.icnn_destroy(model)
    bad = state_dict(weights)
    bad["layers.1.weights"] = bad["layers.1.weights"][:32]
    with pytest.raises(ValueError, match="shape"):
        ctx.icnn_create(bad)


# ------------------------------------------------------------------ analytic Isihara model (demo_hyperelasticity.py:686-703)
def test_isihara_oracle_matches_torch_differentiation(golden):
    from oracle.icnn_oracle import isihara_stress_tangent

    g = golden("isihara_analytic.npz")
    dP, P = isihara_stress_tangent(g["F"])
    assert np.abs(P - g["P"]).max() <= 1e-13 * np.abs(g["P"]).max()
    assert np.abs(dP - g["dP"]).max() <= 1e-13 * np.abs(g["dP"]).max()
    assert np.abs(P[0]).max() == 0.0                      # W has its minimum at F = I
    assert np.abs(dP - dP.transpose(0, 2, 1)).max() <= 1e-13 * np.abs(dP).max()   # a Hessian
    dPn, Pn = isihara_stress_tangent(np.array([[1.0, 0.0, 0.0, -1.0]]))
    assert np.isnan(Pn).all() and np.isnan(dPn).all()     # det F < 0: J^(-2/3) has no real value


def test_network_approximates_the_analytic_model_it_was_trained_on(golden, weights):
    """End-to-end sanity of features + weights + chain rule: the shipped ICNN (trained on noisy Isihara data, :314)
    reproduces the analytic stress within a few per cent on the BASELINE config-5 distribution (the reference
    compares the two through the FEM solution, :806-817)."""
    from oracle.icnn_oracle import isihara_stress_tangent

    g = golden("isihara_analytic.npz")
    _, P_a = isihara_stress_tangent(g["F"])
    _, P_n = icnn_stress_tangent(g["F"], weights)
    assert np.abs(P_n - P_a).max() <= 0.06 * np.abs(P_a).max()
    assert np.median(np.abs(P_n - P_a)) <= 0.02 * np.median(np.abs(P_a))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 1000, 4097])
def test_isihara_hip_against_golden_and_oracle(ctx, golden, n):
    from dolfinx_external_operator_amd import MEM_HOST, IsiharaParams
    from oracle.icnn_oracle import isihara_stress_tangent

    g = golden("isihara_analytic.npz")
    rng = np.random.Generator(np.random.PCG64(n))
    F = g["F"][:n] if n <= 1000 else np.array([1.0, 0, 0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    if n > 1000:
        F[5] = [1.0, 0.0, 0.0, -1.0]                      # det F < 0 -> NaN row
    F = np.ascontiguousarray(F)
    dP, P = np.empty((n, 4, 4)), np.empty((n, 4))
    ctx.isihara(IsiharaParams(0.5, 1.0, 1.0, 1.5), n, MEM_HOST, F, dP, P)
    if n == 0:
        return
    with np.errstate(all="ignore"):
        dPo, Po = isihara_stress_tangent(F)
    assert np.array_equal(np.isnan(dP), np.isnan(dPo)) and np.array_equal(np.isnan(P), np.isnan(Po))
    ok = ~np.isnan(Po).any(axis=1)
    # fp64 throughout; the kernel groups the chain rule by (t, D) partials, the oracle by einsum: rounding only
    assert np.abs(P[ok] - Po[ok]).max() <= 1e-12 * np.abs(Po[ok]).max()
    assert np.abs(dP[ok] - dPo[ok]).max() <= 1e-12 * np.abs(dPo[ok]).max()
    if n <= 1000:
        assert np.abs(P - g["P"][:n]).max() <= 1e-12 * np.abs(g["P"]).max()
        assert np.abs(dP - g["dP"][:n]).max() <= 1e-12 * np.abs(g["dP"]).max()


@pytest.mark.gpu
def test_isihara_factory_contract_and_device_pointers(ctx, golden):
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE, IsiharaParams, make_isihara

    g = golden("isihara_analytic.npz")
    fn = make_isihara(ctx=ctx)
    dP, P = fn((1,))(g["F"].reshape(-1, 1, 2, 2))
    assert dP.shape == (g["F"].shape[0] * 16,) and P.shape == (g["F"].shape[0] * 4,)
    assert np.abs(P.reshape(-1, 4) - g["P"]).max() <= 1e-12 * np.abs(g["P"]).max()
    with pytest.raises(NotImplementedError):
        fn((0,))
    n = 100_003
    Ft = torch.randn(n, 4, dtype=torch.float64, device="cuda") * 0.05 + torch.tensor([1.0, 0, 0, 1.0], dtype=torch.float64, device="cuda")
    dPt = torch.empty(n * 16, dtype=torch.float64, device="cuda")
    Pt = torch.empty(n * 4, dtype=torch.float64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.isihara(IsiharaParams(0.5, 1.0, 1.0, 1.5), n, MEM_DEVICE, Ft.data_ptr(), dPt.data_ptr(), Pt.data_ptr())
    torch.cuda.synchronize()
    from oracle.icnn_oracle import isihara_stress_tangent
    dPo, Po = isihara_stress_tangent(Ft.cpu().numpy())
    assert np.abs(dPt.cpu().numpy().reshape(-1, 4, 4) - dPo).max() <= 1e-12 * np.abs(dPo).max()
    assert np.abs(Pt.cpu().numpy().reshape(-1, 4) - Po).max() <= 1e-12 * np.abs(Po).max()
