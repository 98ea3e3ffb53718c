"""Host half of the brick form of the consumer-side scatter (csrc/cell8_brick_host.h; not gpu): cells -> Morton order -> groups of 8 ->
reduction tables + transposed map, compiled on its own with g++ and replayed on the CPU against the plain per-cell sum of element vectors."""
import json
import pathlib
import subprocess

import numpy as np
import pytest

from tools.synthetic import structured_mesh

ROOT = pathlib.Path(__file__).resolve().parents[3]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp("brick") / "brick_host"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", f"-I{ROOT / 'scripts' / 'exp' / 'brick_form'}",
                    str(ROOT / "scripts" / "exp" / "brick_form" / "brick_host.cpp"), "-o", str(exe)], check=True)
    return exe


def run(exe, m, shuffle=None):
    dofmap, geom = m.dofmap, m.geom_dofmap
    if shuffle is not None:
        perm = np.random.Generator(np.random.PCG64(shuffle)).permutation(m.num_cells)
        dofmap, geom = dofmap[perm], geom[perm]
    X = m.x[geom]                                            # (nc, 8, 3), tensor-product vertex order
    xyz = X.mean(axis=1)
    bits = np.array([[(v >> k) & 1 for v in range(8)] for k in range(3)])      # edge vector k: mean of bit-k-set vertices minus the others
    ev = np.stack([X[:, bits[k] == 1].mean(axis=1) - X[:, bits[k] == 0].mean(axis=1) for k in range(3)], axis=1)      # (nc, 3 edges, 3)
    ext = np.abs(ev).max(axis=1).mean(axis=0)                # as dxo_mesh_create computes h_cell_ext
    text = f"{m.num_cells} {dofmap.shape[1]} {m.node_x.shape[0]}\n" + " ".join(map(str, dofmap.reshape(-1))) + "\n" + \
           " ".join(map(str, geom.reshape(-1))) + "\n" + " ".join(f"{v:.9g}" for v in xyz.reshape(-1)) + "\n" + " ".join(f"{v:.17g}" for v in ext) + "\n"
    res = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True)
    return json.loads(res.stdout)


@pytest.mark.parametrize("n, degree, shuffle", [((4, 4, 4), 2, None), ((6, 4, 2), 2, 5), ((4, 4, 4), 1, None), ((5, 3, 3), 2, 1), ((1, 1, 1), 2, None),
                                               ((8, 8, 8), 2, 9)])
def test_bricks_reduce_to_the_per_cell_sum(harness, n, degree, shuffle):
    m = structured_mesh("hexahedron", n, degree, distort=0.2, seed=2)
    r = run(harness, m, shuffle)
    assert r["ok"] and r["bad"] == 0 and r["err"] < 1e-13, r
    nd = (degree + 1) ** 3
    if all(k % 2 == 0 for k in n):
        # even sizes: every Morton run of 8 is a 2 x 2 x 2 brick, whatever order the caller numbered the cells in —
        # 125 (Q2) / 27 (Q1) partials per group instead of 8 x nd entries, ONE reduction table for the whole mesh, at most 8 partials per node
        u = (2 * degree + 1) ** 3
        assert r["min_u"] == r["max_u"] == u and r["tables"] == 1 and r["max_partials_per_node"] <= 8, r
        assert r["slots"] == r["groups"] * u
    else:
        assert r["max_u"] <= 8 * nd and r["slots"] <= m.num_cells * nd
