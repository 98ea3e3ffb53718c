"""DOLFINx-free element tables and structured meshes for the tests and benches of the device operand evaluation.

On a DOLFINx installation the tables come from basix (`element.tabulate(1, points)`) and the arrays from
`V.dofmap.list`, `mesh.geometry.dofmap`, `mesh.geometry.x` (see `dolfinx_external_operator_amd.operand_eval.DeviceMesh.from_dolfinx`). The GPU
box has neither, so this module provides what the synthetic workloads need: Lagrange bases of degree 1 and 2 on
the reference triangle, quadrilateral, tetrahedron and hexahedron (built by inverting a Vandermonde matrix, so no
hand-written shape functions), Gauss rules matching the reference's quadrature degree 2, and structured meshes
with consistent field and geometry dofmaps. Node ORDER inside a cell is this module's own (lattice order), which
is all a self-consistent (tables, dofmap) pair needs; it is not basix's.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass

import numpy as np


def _lattice_nodes(cell: str, degree: int) -> np.ndarray:
    tdim = {"triangle": 2, "quadrilateral": 2, "tetrahedron": 3, "hexahedron": 3}[cell]
    pts = []
    for idx in itertools.product(range(degree + 1), repeat=tdim):
        if cell in ("triangle", "tetrahedron") and sum(idx) > degree:
            continue
        pts.append([i / degree for i in idx[::-1]])      # x fastest
    return np.array(pts, dtype=np.float64)


def _exponents(cell: str, degree: int) -> np.ndarray:
    tdim = {"triangle": 2, "quadrilateral": 2, "tetrahedron": 3, "hexahedron": 3}[cell]
    ex = [e for e in itertools.product(range(degree + 1), repeat=tdim)
          if cell in ("quadrilateral", "hexahedron") or sum(e) <= degree]
    return np.array(ex, dtype=np.int64)


def _monomials(ex: np.ndarray, pts: np.ndarray, deriv: int | None = None) -> np.ndarray:
    """(npts, nmono) values of the monomials x^e, or of their derivative along axis `deriv`."""
    out = np.ones((pts.shape[0], ex.shape[0]))
    for k in range(ex.shape[1]):
        e = ex[:, k]
        if deriv == k:
            out *= e[None, :] * pts[:, k:k + 1] ** np.maximum(e - 1, 0)[None, :]
        else:
            out *= pts[:, k:k + 1] ** e[None, :]
    return out


@dataclass(frozen=True)
class LagrangeElement:
    cell: str
    degree: int

    @property
    def nodes(self) -> np.ndarray:
        return _lattice_nodes(self.cell, self.degree)

    def tabulate(self, points: np.ndarray):
        """phi (npts, ndofs), dphi (npts, ndofs, tdim) at reference `points`."""
        ex = _exponents(self.cell, self.degree)
        coeff = np.linalg.inv(_monomials(ex, self.nodes))          # (nmono, ndofs): phi_a = sum_m coeff[m,a] x^e_m
        phi = _monomials(ex, points) @ coeff
        dphi = np.stack([_monomials(ex, points, k) @ coeff for k in range(points.shape[1])], axis=2)
        return phi, dphi


def quadrature_degree2(cell: str):
    """Points and weights integrating degree 2 exactly: the rules DOLFINx's default scheme gives for
    `quadrature_degree = 2` (3-point triangle rule, demo_plasticity_von_mises.py:230-245; 4-point tetrahedron;
    2-point Gauss per direction on quadrilaterals and hexahedra)."""
    g = 0.5 - 0.5 / np.sqrt(3.0)
    line = np.array([g, 1.0 - g])
    if cell == "triangle":
        return np.array([[1 / 6, 1 / 6], [1 / 6, 2 / 3], [2 / 3, 1 / 6]]), np.full(3, 1 / 6)
    if cell == "tetrahedron":
        a, b = 0.1381966011250105, 0.5854101966249685
        return np.array([[a, a, a], [b, a, a], [a, b, a], [a, a, b]]), np.full(4, 1 / 24)
    if cell == "quadrilateral":
        return np.array([[x, y] for y in line for x in line]), np.full(4, 0.25)
    if cell == "hexahedron":
        return np.array([[x, y, z] for z in line for y in line for x in line]), np.full(8, 0.125)
    raise ValueError(cell)


def gauss_tensor_rule(cell: str, npd: int):
    """npd-point Gauss-Legendre rule per direction on the quadrilateral / hexahedron (x fastest): npd = 3 on the hexahedron is the
    27-point rule DOLFINx picks for `quadrature_degree` 4-5."""
    t, w = np.polynomial.legendre.leggauss(npd)
    t, w = 0.5 * (t + 1.0), 0.5 * w
    gdim = {"quadrilateral": 2, "hexahedron": 3}[cell]
    idx = [c[::-1] for c in itertools.product(range(npd), repeat=gdim)]
    return np.array([[t[i] for i in c] for c in idx]), np.array([np.prod([w[i] for i in c]) for c in idx])


def with_rule(mesh: "SyntheticMesh", points: np.ndarray, weights: np.ndarray) -> "SyntheticMesh":
    """The same mesh tabulated at another quadrature rule."""
    import dataclasses

    phi, dphi = LagrangeElement(mesh.cell, mesh.degree).tabulate(points)
    psi, dpsi = LagrangeElement(mesh.cell, 1).tabulate(points)
    return dataclasses.replace(mesh, points=np.ascontiguousarray(points), phi=phi, dphi=dphi, dpsi=dpsi, weights=np.ascontiguousarray(weights), psi=psi)


@dataclass
class SyntheticMesh:
    """Arrays of a structured mesh in the layout `dxo_mesh_desc` takes."""
    cell: str
    gdim: int
    degree: int
    x: np.ndarray             # (num_geom_nodes, gdim)
    geom_dofmap: np.ndarray   # (num_cells, ngeom) int32, degree-1 coordinate element
    dofmap: np.ndarray        # (num_cells, ndofs) int32, field element of `degree`
    node_x: np.ndarray        # (num_field_nodes, gdim) physical position of every field node (for interpolation)
    points: np.ndarray        # (nq, gdim) reference quadrature points
    phi: np.ndarray
    dphi: np.ndarray
    dpsi: np.ndarray
    weights: np.ndarray | None = None   # (nq,) reference quadrature weights
    psi: np.ndarray | None = None       # (nq, ngeom) values of the coordinate element at the points (the operand `x`, SpatialCoordinate)

    @property
    def num_cells(self) -> int:
        return self.dofmap.shape[0]

    @property
    def nq(self) -> int:
        return self.points.shape[0]

    def physical_points(self) -> np.ndarray:
        """(num_cells, nq, gdim) positions of the quadrature points."""
        geo = LagrangeElement(self.cell, 1)
        psi, _ = geo.tabulate(self.points)
        return np.einsum("qv,cvj->cqj", psi, self.x[self.geom_dofmap])


_MESH_CACHE: dict = {}


def structured_mesh_cached(cell: str, n: tuple[int, ...], degree: int = 2, distort: float = 0.0, seed: int = 0) -> SyntheticMesh:
    """`structured_mesh`, the last few results kept (bench legs that share the 108^3 Q2 mesh build it once: 3 s each time).
    The legs treat a mesh as read-only."""
    key = (cell, tuple(int(k) for k in n), degree, float(distort), seed)
    m = _MESH_CACHE.get(key)
    if m is None:
        if len(_MESH_CACHE) >= 3:
            _MESH_CACHE.pop(next(iter(_MESH_CACHE)))
        m = _MESH_CACHE[key] = structured_mesh(cell, n, degree, distort, seed)
    return m


def structured_mesh(cell: str, n: tuple[int, ...], degree: int = 2, distort: float = 0.0, seed: int = 0) -> SyntheticMesh:
    """Unit square / cube split into n[0] x n[1] (x n[2]) boxes; triangles: 2 per box, tetrahedra: 6 per box.

    `distort` moves the interior VERTICES by up to that fraction of the box size (seeded), so simplices stay
    affine with different Jacobians and quadrilaterals / hexahedra become genuinely non-affine (bi/trilinear
    geometry, J varies inside the cell). Field nodes sit at the images of the reference nodes under the cell map."""
    gdim = len(n)
    if {"triangle": 2, "quadrilateral": 2, "tetrahedron": 3, "hexahedron": 3}[cell] != gdim:
        raise ValueError("len(n) must equal the cell's dimension")
    n = tuple(int(k) for k in n)
    rng = np.random.Generator(np.random.PCG64(seed))
    vshape = tuple(k + 1 for k in n)                       # vertices per direction
    grids = np.meshgrid(*[np.linspace(0.0, 1.0, k + 1) for k in n], indexing="ij")
    vx = np.stack(grids, axis=-1)                          # (..., gdim), index order (i, j[, k])
    if distort:
        h = np.array([1.0 / k for k in n])
        move = rng.uniform(-distort, distort, size=vx.shape) * h
        interior = np.ones(vshape, dtype=bool)
        for ax in range(gdim):
            sl = [slice(None)] * gdim
            sl[ax] = [0, -1]
            interior[tuple(sl)] = False
        vx = vx + move * interior[..., None]
    vid = np.arange(int(np.prod(vshape))).reshape(vshape)
    fshape = tuple(degree * k + 1 for k in n)              # field-node lattice
    fid = np.arange(int(np.prod(fshape))).reshape(fshape)

    B = np.stack(np.meshgrid(*[np.arange(k) for k in n], indexing="ij"), axis=-1).reshape(-1, gdim)   # boxes

    def lattice_ids(offsets, scale, shape):
        """ids of lattice points scale*box + offsets[l] for every box: (nboxes, len(offsets))"""
        idx = scale * B[:, None, :] + np.asarray(offsets)[None, :, :]
        return np.ravel_multi_index(tuple(idx.transpose(2, 0, 1)), shape).astype(np.int32)

    if cell in ("quadrilateral", "hexahedron"):
        # reference lattice order: x fastest  ->  local node (a_x, a_y[, a_z]) with x fastest
        corner = [c[::-1] for c in itertools.product(range(2), repeat=gdim)]
        geom = lattice_ids(corner, 1, vshape)
        lat = [c[::-1] for c in itertools.product(range(degree + 1), repeat=gdim)]
        dm = lattice_ids(lat, degree, fshape)
    else:
        # simplices of a box: one per permutation of the axes (Kuhn triangulation): vertices 0, e_p0, e_p0+e_p1, ...
        ref_lat = _lattice_nodes(cell, degree)             # reference node positions, x fastest
        geom_parts, dm_parts = [], []
        for perm in itertools.permutations(range(gdim)):
            corners = [np.zeros(gdim, dtype=int)]
            for ax in perm:
                nxt = corners[-1].copy()
                nxt[ax] += 1
                corners.append(nxt)
            corners = np.array(corners)                    # (gdim+1, gdim) offsets inside the box
            edges = corners[1:] - corners[0]
            # field node for reference position r: lattice offset = degree*corners[0] + degree * sum_k r_k edges[k]
            offs = degree * corners[0] + np.rint(degree * (ref_lat @ edges)).astype(int)
            geom_parts.append(lattice_ids(corners, 1, vshape))
            dm_parts.append(lattice_ids(offs, degree, fshape))
        geom = np.stack(geom_parts, axis=1).reshape(-1, gdim + 1)        # box-major, permutation inside
        dm = np.stack(dm_parts, axis=1).reshape(-1, ref_lat.shape[0])
        # (half of the Kuhn simplices have det J < 0, as DOLFINx meshes may: the push-forward does not care)
    x = vx.reshape(-1, gdim)
    fe = LagrangeElement(cell, degree)
    geo = LagrangeElement(cell, 1)
    points, weights = quadrature_degree2(cell)
    phi, dphi = fe.tabulate(points)
    psi, dpsi = geo.tabulate(points)
    # physical position of field nodes: image of the reference nodes under each cell's map (shared nodes agree)
    psi_nodes, _ = geo.tabulate(fe.nodes)
    node_x = np.zeros((int(np.prod(fshape)), gdim))
    node_x[dm.reshape(-1)] = np.einsum("av,cvj->caj", psi_nodes, x[geom]).reshape(-1, gdim)
    return SyntheticMesh(cell, gdim, degree, np.ascontiguousarray(x), geom, dm, node_x, points, phi, dphi, dpsi, weights, psi)


# ---------------------------------------------------------------------------------------------- codim-1 (facets)
# Local facets of the reference cells as lists of local VERTEX indices (vertices in this module's lattice order:
# triangle (0,0),(1,0),(0,1); quadrilateral (0,0),(1,0),(0,1),(1,1); tetrahedron origin, e_x, e_y, e_z; hexahedron x
# fastest). Simplices: facet i is opposite vertex i (the DOLFINx convention); tensor cells: one facet per side.
FACETS = {
    "triangle": [(1, 2), (0, 2), (0, 1)],
    "quadrilateral": [(0, 1), (0, 2), (1, 3), (2, 3)],
    "tetrahedron": [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)],
    "hexahedron": [(0, 1, 2, 3), (0, 1, 4, 5), (0, 2, 4, 6), (1, 3, 5, 7), (2, 3, 6, 7), (4, 5, 6, 7)],
}
_FACET_CELL = {"triangle": "interval", "quadrilateral": "interval", "tetrahedron": "triangle", "hexahedron": "quadrilateral"}


def facet_quadrature_degree2(cell: str):
    """Reference points and weights on the facet's own reference cell (interval [0,1], triangle, unit square)."""
    fc = _FACET_CELL[cell]
    if fc == "interval":
        g = 0.5 - 0.5 / np.sqrt(3.0)
        return np.array([[g], [1.0 - g]]), np.full(2, 0.5)
    return quadrature_degree2(fc)


def facet_points_in_cell(cell: str, facet_points: np.ndarray) -> np.ndarray:
    """(n_local_facets, nq, tdim): the facet quadrature points mapped into the reference CELL, per local facet — what
    DOLFINx does before tabulating an Expression on (cell, local_facet) entities."""
    verts = LagrangeElement(cell, 1).nodes                     # reference vertices, lattice order
    out = []
    for f in FACETS[cell]:
        v = verts[list(f)]
        s = np.asarray(facet_points)
        if len(f) == 2:                                        # interval
            pts = v[0] + s[:, :1] * (v[1] - v[0])
        elif len(f) == 3:                                      # triangle: affine
            pts = v[0] + s[:, :1] * (v[1] - v[0]) + s[:, 1:2] * (v[2] - v[0])
        else:                                                  # quadrilateral face: bilinear, lattice order (x fastest)
            a, b = s[:, :1], s[:, 1:2]
            pts = (1 - a) * (1 - b) * v[0] + a * (1 - b) * v[1] + (1 - a) * b * v[2] + a * b * v[3]
        out.append(pts)
    return np.array(out)


def facet_tables(mesh: "SyntheticMesh"):
    """phi_f (nf, nq, ndofs), dphi_f (nf, nq, ndofs, gdim), dpsi_f (nf, nq, ngeom, gdim), ref points (nf, nq, gdim)."""
    fpts, _ = facet_quadrature_degree2(mesh.cell)
    pts = facet_points_in_cell(mesh.cell, fpts)
    fe, geo = LagrangeElement(mesh.cell, mesh.degree), LagrangeElement(mesh.cell, 1)
    tabs = [fe.tabulate(p) for p in pts]
    return (np.array([t[0] for t in tabs]), np.array([t[1] for t in tabs]), np.array([geo.tabulate(p)[1] for p in pts]), pts)


def facet_physical_points(mesh: "SyntheticMesh", entities: np.ndarray, ref_points: np.ndarray) -> np.ndarray:
    """(n_entities, nq, gdim) physical positions of the facet points of the (cell, local_facet) entities."""
    geo = LagrangeElement(mesh.cell, 1)
    out = np.empty((len(entities), ref_points.shape[1], mesh.gdim))
    for i, (c, f) in enumerate(np.asarray(entities)):
        psi, _ = geo.tabulate(ref_points[f])
        out[i] = psi @ mesh.x[mesh.geom_dofmap[c]]
    return out
