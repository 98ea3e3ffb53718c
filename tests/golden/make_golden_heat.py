#!/usr/bin/env python3
"""Generate the golden vectors of BASELINE config 1 (nonlinear heat, 32x32 unit square, 6 144 points).

Run ONLY in the build container (needs /root/reference). The reference kernels are not copied:
`k`, `q_impl`, `dqdT_impl`, `dqdsigma_impl` and the constants `A`, `B`, `Id` are pulled out of
doc/demo/demo_nonlinear_heat_equation_part2.py (:209-261) with `ast` and executed on operand arrays
built here without DOLFINx:

  * mesh: unit square, 32 x 32 squares each split into two triangles (2 048 cells);
  * T = x^2 + y interpolated at the vertices (P1, part2.py:148), evaluated at the three points of the
    degree-2 triangle rule (1/6,1/6), (1/6,2/3), (2/3,1/6) (part2.py:158-160);
  * sigma = grad(T) (part2.py:137), constant per cell for P1.

The kernels are pointwise maps, so the ordering of cells/points is irrelevant to parity.
Output: tests/golden/heat_c1.npz
"""
import ast
import pathlib

import numpy as np

REF = pathlib.Path("/root/reference/doc/demo/demo_nonlinear_heat_equation_part2.py")
OUT = pathlib.Path(__file__).resolve().parent

WANT_FUNCS = {"k", "q_impl", "dqdT_impl", "dqdsigma_impl"}
WANT_CONSTS = {"A", "B", "Id"}


def _extract():
    tree = ast.parse(REF.read_text())
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANT_FUNCS:
            body.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in WANT_CONSTS:
            body.append(node)
    return ast.Module(body=body, type_ignores=[])


def _operands(n=32):
    xs = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    vid = lambda i, j: i * (n + 1) + j  # noqa: E731
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    cells = []
    for i in range(n):
        for j in range(n):
            v00, v10, v01, v11 = vid(i, j), vid(i + 1, j), vid(i, j + 1), vid(i + 1, j + 1)
            cells.append((v00, v10, v11))
            cells.append((v00, v01, v11))
    cells = np.array(cells)
    nodal = coords[:, 0] ** 2 + coords[:, 1]
    qp = np.array([[1 / 6, 1 / 6], [1 / 6, 2 / 3], [2 / 3, 1 / 6]])
    P = coords[cells]                      # (nc, 3, 2)
    Tn = nodal[cells]                      # (nc, 3)
    # P1 basis on the reference triangle: 1-x-y, x, y
    phi = np.stack([1 - qp[:, 0] - qp[:, 1], qp[:, 0], qp[:, 1]], axis=1)   # (nq, 3)
    T = Tn @ phi.T                          # (nc, nq)
    # gradient: solve J^T g = dT/dxi
    J = np.stack([P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]], axis=2)  # (nc, 2, 2) columns = edge vectors
    dT_ref = np.stack([Tn[:, 1] - Tn[:, 0], Tn[:, 2] - Tn[:, 0]], axis=1)  # (nc, 2)
    g = np.linalg.solve(np.transpose(J, (0, 2, 1)), dT_ref[..., None])[..., 0]
    sigma = np.repeat(g[:, None, :], qp.shape[0], axis=1)          # (nc, nq, 2)
    return T, sigma


def main():
    ns = {"np": np, "gdim": 2}
    exec(compile(_extract(), str(REF), "exec"), ns)
    T, sigma = _operands(32)
    nc, nq = T.shape
    sigma_flat = sigma.reshape(nc, nq * 2)  # the layout described at part2.py:222
    q = ns["q_impl"](T, sigma_flat)
    dqdT = ns["dqdT_impl"](T, sigma_flat)
    dqds = ns["dqdsigma_impl"](T, sigma_flat)
    assert q.shape == (nc * nq * 2,) and dqds.shape == (nc * nq * 4,)
    np.savez_compressed(OUT / "heat_c1.npz", A=ns["A"], B=ns["B"], T=T, sigma=sigma, q=q, dqdT=dqdT, dqdsigma=dqds)
    print(f"heat config 1: cells={nc} points={nc * nq} |q|max={np.abs(q).max():.4f}")


if __name__ == "__main__":
    main()
