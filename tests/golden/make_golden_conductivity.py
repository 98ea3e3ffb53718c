#!/usr/bin/env python3
"""Golden vectors of the part-1 heat demo's scalar conductivity operator ("scalar-k variant").

Run ONLY in the build container (needs /root/reference). `k_impl`, `dkdT_impl` and the constants `A`, `B` are pulled out of
doc/demo/demo_nonlinear_heat_equation_part1.py (:247-271) with `ast` and executed here; nothing of the reference's source is
copied. The operator lives on a CG space in the demo (P2 on a 10 x 10 unit square, part1.py:164-166, :212): its operand
is T evaluated at the space's interpolation points, (num_cells, 6) values with shared nodes repeated, and its values
reach the coefficient through the unrolled dofmap (external_operator.py:286-287). The fixture therefore carries
  * T at the interpolation points of a P2 space on a 10 x 10 unit square of triangles built here without DOLFINx
    (T = x^2 + y, the demo's boundary data, part1.py:189), plus seeded values that exercise negative / large T,
  * the P2 dofmap of that mesh (cell -> 6 global dofs),
  * k and dk/dT as the reference's functions return them (flat, cell-major),
  * the coefficient vector that `coefficient.x.array[dofmap] = values` leaves (last writer wins).
Output: tests/golden/conductivity_p1.npz
"""
import ast
import pathlib

import numpy as np

REF = pathlib.Path("/root/reference/doc/demo/demo_nonlinear_heat_equation_part1.py")
OUT = pathlib.Path(__file__).resolve().parent


def _extract():
    tree = ast.parse(REF.read_text())
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in {"k_impl", "dkdT_impl"}:
            body.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in {"A", "B"}:
            body.append(node)
    return ast.Module(body=body, type_ignores=[])


def p2_mesh(n=10):
    """P2 triangles on the unit square: lattice of (2n+1)^2 nodes, two triangles per square, 6 nodes per cell
    (3 vertices, then the 3 edge midpoints)."""
    m = 2 * n + 1
    nid = lambda i, j: i * m + j  # noqa: E731
    xs = np.linspace(0.0, 1.0, m)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    cells = []
    for i in range(0, 2 * n, 2):
        for j in range(0, 2 * n, 2):
            a, b, c, d = (i, j), (i + 2, j), (i, j + 2), (i + 2, j + 2)
            for tri in ((a, b, d), (a, c, d)):
                mids = [((tri[k][0] + tri[(k + 1) % 3][0]) // 2, (tri[k][1] + tri[(k + 1) % 3][1]) // 2) for k in range(3)]
                cells.append([nid(*v) for v in tri] + [nid(*v) for v in mids])
    return coords, np.array(cells, dtype=np.int32)


def main():
    ns = {"np": np}
    exec(compile(_extract(), str(REF), "exec"), ns)
    coords, dofmap = p2_mesh(10)
    T_nodes = coords[:, 0] ** 2 + coords[:, 1]
    T = T_nodes[dofmap]                                   # operand at the interpolation points: (num_cells, 6)
    k = ns["k_impl"](T)
    dk = ns["dkdT_impl"](T)
    coeff_k = np.zeros(coords.shape[0])
    coeff_k[dofmap.reshape(-1)] = k                        # external_operator.py:286-287 with bs = 1
    rng = np.random.Generator(np.random.PCG64(17))
    T_rand = np.concatenate([rng.normal(0.0, 3.0, 4000), [-1.0, 0.0, -0.5, 1e12, -1e12]])   # includes the pole A + B T = 0
    with np.errstate(all="ignore"):
        k_rand, dk_rand = ns["k_impl"](T_rand), ns["dkdT_impl"](T_rand)
    np.savez_compressed(OUT / "conductivity_p1.npz", A=ns["A"], B=ns["B"], T=T, dofmap=dofmap, k=k, dkdT=dk, coeff_k=coeff_k,
                        T_rand=T_rand, k_rand=k_rand, dkdT_rand=dk_rand)
    print(f"conductivity: cells={dofmap.shape[0]} dofs={coords.shape[0]} values={k.size} k in [{k.min():.4f}, {k.max():.4f}]")


if __name__ == "__main__":
    main()
