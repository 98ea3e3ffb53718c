"""`DeviceMesh.from_dolfinx` with stand-ins for DOLFINx / basix objects (neither is installable here or on the GPU box).

The stand-ins follow the layouts the two libraries document — `basix.finite_element.FiniteElement.tabulate(n, x)` returns
`(n_derivatives, n_points, n_dofs, value_size)`, index 0 the values and 1 + k the derivative along axis k;
`mesh.geometry.x` has THREE columns whatever the geometric dimension; `V.dofmap.list` / `mesh.geometry.dofmap` are
`(num_cells, n)` int32 — and are filled from `tools.synthetic`. What is pinned: the axis shuffling, the coordinate
element reconstruction from `cmap`, the padded coordinates and the refusal of elements with dof transformations.
Parity with a real DOLFINx build stays unpinned."""
import sys
import types

import numpy as np
import pytest

from dolfinx_external_operator_amd import DeviceMesh
from tools.synthetic import LagrangeElement, structured_mesh


class _BasixElement:
    def __init__(self, cell, degree):
        self._el = LagrangeElement(cell, degree)

    def tabulate(self, nderiv, points):
        assert nderiv == 1
        phi, dphi = self._el.tabulate(np.asarray(points))
        tab = np.empty((1 + points.shape[1], points.shape[0], phi.shape[1], 1))
        tab[0, :, :, 0] = phi
        for k in range(points.shape[1]):
            tab[1 + k, :, :, 0] = dphi[:, :, k]
        return tab


@pytest.fixture
def fake_basix(monkeypatch):
    mod = types.ModuleType("basix")
    mod.CellType = types.SimpleNamespace(triangle="triangle", quadrilateral="quadrilateral", tetrahedron="tetrahedron",
                                         hexahedron="hexahedron")
    mod.ElementFamily = types.SimpleNamespace(P="P")
    mod.LagrangeVariant = lambda v: v
    calls = []

    def create_element(family, cell, degree, variant):
        calls.append((family, cell, degree, variant))
        return _BasixElement(cell, degree)

    mod.create_element = create_element
    mod.calls = calls
    monkeypatch.setitem(sys.modules, "basix", mod)
    return mod


def _space(m, transformations=False):
    x3 = np.zeros((m.x.shape[0], 3))
    x3[:, :m.gdim] = m.x
    geometry = types.SimpleNamespace(dim=m.gdim, cmap=types.SimpleNamespace(degree=1, variant=2), dofmap=m.geom_dofmap, x=x3)
    mesh = types.SimpleNamespace(geometry=geometry, topology=types.SimpleNamespace(cell_name=lambda: m.cell))
    n_nodes = m.node_x.shape[0]
    return types.SimpleNamespace(
        mesh=mesh, element=types.SimpleNamespace(basix_element=_BasixElement(m.cell, m.degree), needs_dof_transformations=transformations),
        dofmap=types.SimpleNamespace(list=m.dofmap, index_map=types.SimpleNamespace(size_local=n_nodes - 5, num_ghosts=5)))


@pytest.mark.parametrize("cell,n", [("triangle", (4, 3)), ("hexahedron", (3, 2, 2)), ("tetrahedron", (2, 2, 2)), ("quadrilateral", (3, 3))])
def test_tables_from_dolfinx_match_the_synthetic_tables(fake_basix, cell, n):
    m = structured_mesh(cell, n, 2, distort=0.2, seed=3)
    kw = DeviceMesh.tables_from_dolfinx(_space(m), m.points)
    assert kw["gdim"] == m.gdim and kw["num_field_nodes"] == m.node_x.shape[0]     # owned + ghost nodes
    np.testing.assert_array_equal(kw["phi"], m.phi)
    np.testing.assert_array_equal(kw["dphi"], m.dphi)                              # (nq, ndofs, gdim)
    np.testing.assert_array_equal(kw["dpsi"], m.dpsi)                              # degree-1 coordinate element from cmap
    np.testing.assert_array_equal(kw["psi"], m.psi)                                # its values: the operand `x` (SpatialCoordinate)
    assert fake_basix.calls == [("P", cell, 1, 2)]
    assert kw["x"].shape[1] == 3 and kw["dofmap"] is m.dofmap and kw["geom_dofmap"] is m.geom_dofmap


def test_elements_with_dof_transformations_are_refused(fake_basix):
    m = structured_mesh("triangle", (2, 2), 2)
    with pytest.raises(NotImplementedError, match="dof transformations"):
        DeviceMesh.tables_from_dolfinx(_space(m, transformations=True), m.points)


@pytest.mark.gpu
@pytest.mark.parametrize("cell,n", [("triangle", (9, 7)), ("hexahedron", (4, 3, 5))])
def test_from_dolfinx_evaluates_like_from_synthetic(fake_basix, ctx, cell, n):
    m = structured_mesh(cell, n, 2, distort=0.2, seed=5)
    rng = np.random.Generator(np.random.PCG64(2))
    u = rng.normal(0.0, 1e-3, m.node_x.shape[0] * m.gdim)
    a = DeviceMesh.from_dolfinx(_space(m), m.points, ctx=ctx).evaluate("eps", m.gdim, u)
    b = DeviceMesh.from_synthetic(m, ctx=ctx).evaluate("eps", m.gdim, u)
    np.testing.assert_array_equal(a, b)                                            # 3-column coordinates, same kernel
    xa = DeviceMesh.from_dolfinx(_space(m), m.points, ctx=ctx).coordinate()
    np.testing.assert_allclose(xa, m.physical_points(), rtol=1e-14, atol=1e-16)    # x = SpatialCoordinate from the cmap's values
