"""Operand evaluation on the GPU: the device counterpart of `fem.Expression(operand, points).eval(mesh, entities)`
inside `evaluate_operands` (src/dolfinx_external_operator/external_operator.py:386-402), for operands that are
linear in the value or gradient of one Lagrange field — eps(Du), I + grad u, T, grad T: every operand of the
reference's demos.

    mesh = DeviceMesh(gdim=2, phi=..., dphi=..., dpsi=..., dofmap=V.dofmap.list, geom_dofmap=domain.geometry.dofmap,
                      x=domain.geometry.x, num_field_nodes=...)
    deps = mesh.operand("eps", Du)           # an object with .eval(entities) -> (len(entities), nq, 4)
    evaluated = evaluate_operands([sigma])   # evaluation.py mirror; or call deps.eval(cells) directly

The field holder (`fem.Function`, ndarray or zero-argument callable) is re-read at every evaluation, like the
reference's Expression re-reads the Function's vector.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import MEM_DEVICE, MEM_HOST, Context, DxoError, default_context

KINDS = {"value": 0, "grad": 1, "eps": 2, "F": 3, "value_grad": 4,
         # nonlinear operands of F = I + grad u (the reference's own operand test, test/test_operands_evaluation.py:32-36): forward only
         "C": 5, "I1": 6, "detF": 7,
         "div": 8}        # div u (linear: forward and adjoint), test/test_external_operators_evaluation.py:141


class MeshDesc(C.Structure):
    """dxo_mesh_desc (include/dxo.h)."""
    _fields_ = [("gdim", C.c_int32), ("nq", C.c_int32), ("ndofs", C.c_int32), ("ngeom", C.c_int32),
                ("x_stride", C.c_int32), ("_pad", C.c_int32),
                ("num_cells", C.c_int64), ("num_field_nodes", C.c_int64), ("num_geom_nodes", C.c_int64),
                ("phi", C.c_void_p), ("dphi", C.c_void_p), ("dpsi", C.c_void_p),
                ("dofmap", C.c_void_p), ("geom_dofmap", C.c_void_p), ("x", C.c_void_p)]


def _state(holder):
    if callable(holder) and not hasattr(holder, "x") and not hasattr(holder, "data_ptr"):
        holder = holder()
    xv = getattr(holder, "x", None)
    if xv is not None and hasattr(xv, "array"):
        return xv.array
    return holder


class DeviceMesh:
    """Element tables, dofmaps and coordinates resident on the GPU (uploaded once)."""

    def __init__(self, *, gdim: int, phi, dphi, dpsi, dofmap, geom_dofmap, x, num_field_nodes: int | None = None,
                 ctx: Context | None = None, device: int = 0, psi=None):
        self.ctx = ctx if ctx is not None else default_context(device)
        phi = np.ascontiguousarray(phi, dtype=np.float64)
        dphi = np.ascontiguousarray(dphi, dtype=np.float64)
        dpsi = np.ascontiguousarray(dpsi, dtype=np.float64)
        dofmap = np.ascontiguousarray(dofmap, dtype=np.int32)
        geom_dofmap = np.ascontiguousarray(geom_dofmap, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        nq, ndofs = phi.shape
        if dphi.shape != (nq, ndofs, gdim) or dpsi.shape[0] != nq or dpsi.shape[2] != gdim:
            raise ValueError("table shapes: phi (nq, ndofs), dphi (nq, ndofs, gdim), dpsi (nq, ngeom, gdim)")
        if dofmap.ndim != 2 or dofmap.shape[1] != ndofs or geom_dofmap.shape != (dofmap.shape[0], dpsi.shape[1]):
            raise ValueError("dofmap (num_cells, ndofs) and geom_dofmap (num_cells, ngeom) do not match the tables")
        if num_field_nodes is None:
            num_field_nodes = int(dofmap.max()) + 1 if dofmap.size else 0
        self.gdim, self.nq, self.num_cells, self.num_field_nodes = gdim, nq, dofmap.shape[0], int(num_field_nodes)
        d = MeshDesc(gdim, nq, ndofs, dpsi.shape[1], x.shape[1], 0, self.num_cells, self.num_field_nodes, x.shape[0],
                     phi.ctypes.data, dphi.ctypes.data, dpsi.ctypes.data, dofmap.ctypes.data, geom_dofmap.ctypes.data,
                     x.ctypes.data)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.dxo_mesh_create(self.ctx._h, C.byref(d), C.byref(h)), "dxo_mesh_create")
        self._h = h
        if psi is not None:     # values of the coordinate element at the points: the operand `x` (set_coordinate_values)
            self.set_coordinate_values(psi)

    @classmethod
    def from_synthetic(cls, mesh, **kw):
        """From a `tools.synthetic.SyntheticMesh` (tests, benches, examples): any object with these attributes."""
        dm = cls(gdim=mesh.gdim, phi=mesh.phi, dphi=mesh.dphi, dpsi=mesh.dpsi, dofmap=mesh.dofmap,
                 geom_dofmap=mesh.geom_dofmap, x=mesh.x, num_field_nodes=mesh.node_x.shape[0], **kw)
        if getattr(mesh, "weights", None) is not None:
            dm.set_weights(mesh.weights)
        if getattr(mesh, "psi", None) is not None:
            dm.set_coordinate_values(mesh.psi)
        return dm

    @staticmethod
    def tables_from_dolfinx(V, quadrature_points) -> dict:
        """The constructor arguments for a DOLFINx function space `V` (blocked Lagrange) and the reference quadrature
        points the operator's quadrature element uses (`basix.make_quadrature(...)[0]`), as plain arrays.
        basix's `tabulate(1, points)` returns (1 + tdim, npoints, ndofs, value_size) with index 0 the values and
        1 + k the derivative along reference axis k; field tables and dofmap both come from `V`, geometry tables and
        geometry dofmap both from `mesh.geometry`, so basix's node numbering never has to be known here.
        Elements whose dofs need per-cell transformations (Lagrange degree >= 3 on unordered meshes) are refused: the
        kernels apply none."""
        import basix

        if getattr(V.element, "needs_dof_transformations", False):
            raise NotImplementedError("DeviceMesh: elements that need dof transformations are not supported")
        mesh = V.mesh
        gdim = mesh.geometry.dim
        quadrature_points = np.asarray(quadrature_points, dtype=np.float64)
        tab = V.element.basix_element.tabulate(1, quadrature_points)
        # the coordinate element as a basix element of the same family / degree / variant as mesh.geometry.cmap
        cmap = mesh.geometry.cmap
        cell = getattr(basix.CellType, mesh.topology.cell_name())
        cel = basix.create_element(basix.ElementFamily.P, cell, cmap.degree, basix.LagrangeVariant(int(cmap.variant)))
        ctab = cel.tabulate(1, quadrature_points)
        if tab.shape[3] != 1:
            raise ValueError("DeviceMesh: the space's basix element must be the SCALAR sub-element of a blocked space")
        n_nodes = V.dofmap.index_map.size_local + V.dofmap.index_map.num_ghosts
        return dict(gdim=gdim, phi=tab[0, :, :, 0], dphi=np.moveaxis(tab[1:, :, :, 0], 0, 2),
                    dpsi=np.moveaxis(ctab[1:, :, :, 0], 0, 2), dofmap=V.dofmap.list, geom_dofmap=mesh.geometry.dofmap,
                    x=mesh.geometry.x, num_field_nodes=n_nodes, psi=ctab[0, :, :, 0])

    @classmethod
    def from_dolfinx(cls, V, quadrature_points, **kw):
        """From a DOLFINx function space and reference quadrature points (see `tables_from_dolfinx`). DOLFINx is not
        installed on the GPU box; tests drive this with stand-in objects that follow basix's documented layouts."""
        return cls(**cls.tables_from_dolfinx(V, quadrature_points), **kw)

    def value_size(self, kind: str, bs: int) -> int:
        r = self.ctx.lib.dxo_operand_value_size(self.gdim, int(bs), KINDS[kind])
        if r < 0:
            raise ValueError(f"operand '{kind}' is not defined for gdim={self.gdim}, block size {bs}")
        return r

    def evaluate(self, kind: str, bs: int, u, entities=None, out=None) -> np.ndarray:
        """Host arrays in, host array (n_cells, nq, value_size) out."""
        D = self.value_size(kind, bs)
        u = np.ascontiguousarray(_state(u), dtype=np.float64).reshape(-1)
        if u.size != self.num_field_nodes * bs:
            raise ValueError(f"field vector has {u.size} entries, expected {self.num_field_nodes * bs}")
        cells = None if entities is None else np.ascontiguousarray(entities, dtype=np.int32)
        n = self.num_cells if cells is None else cells.size
        if out is None:
            out = np.empty((n, self.nq, D))
        rc = self.ctx.lib.dxo_eval_operand(self.ctx._h, self._h, KINDS[kind], int(bs), MEM_HOST, u.ctypes.data,
                                           None if cells is None else cells.ctypes.data, n, out.ctypes.data)
        self.ctx.check(rc, "dxo_eval_operand")
        return out

    def set_coordinate_values(self, psi) -> None:
        """Values of the coordinate element's basis at the quadrature points, (nq, ngeom): what the operand `x` (ufl.SpatialCoordinate,
        test/test_nested_ex_op.py:113-118) needs beside the gradients the mesh already holds. basix: `tabulate(0, points)[0]` of
        `mesh.geometry.cmap`."""
        psi = np.ascontiguousarray(psi, dtype=np.float64)
        if psi.ndim != 2 or psi.shape[0] != self.nq:
            raise ValueError("psi must have shape (nq, ngeom)")
        self.ctx.check(self.ctx.lib.dxo_mesh_set_coordinate_values(self.ctx._h, self._h, psi.ctypes.data), "dxo_mesh_set_coordinate_values")

    def coordinate(self, entities=None, out=None) -> np.ndarray:
        """x at every quadrature point of the cells: host array (n_cells, nq, gdim) — `Expression(SpatialCoordinate(mesh), points).eval`."""
        cells = None if entities is None else np.ascontiguousarray(entities, dtype=np.int32)
        n = self.num_cells if cells is None else cells.size
        if out is None:
            out = np.empty((n, self.nq, self.gdim))
        rc = self.ctx.lib.dxo_eval_coordinate(self.ctx._h, self._h, MEM_HOST, None if cells is None else cells.ctypes.data, n, out.ctypes.data)
        self.ctx.check(rc, "dxo_eval_coordinate")
        return out

    def coordinate_device(self, n_cells: int, out_ptr: int, cells_ptr: int | None = None) -> None:
        """The same into device memory, asynchronous on the context's stream."""
        rc = self.ctx.lib.dxo_eval_coordinate(self.ctx._h, self._h, MEM_DEVICE, None if cells_ptr is None else C.c_void_p(cells_ptr), int(n_cells),
                                              C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_eval_coordinate")

    def set_facet_tables(self, phi, dphi, dpsi) -> None:
        """Tables for codim-1 entities, one set per LOCAL facet of the cell, tabulated at the facet quadrature points
        mapped into the reference cell (what Expression.eval does with `(cell, local_facet)` entities,
        external_operator.py:402; test/test_codim_external_operator.py:76-84): phi (nf, nq, ndofs),
        dphi (nf, nq, ndofs, gdim), dpsi (nf, nq, ngeom, gdim)."""
        phi = np.ascontiguousarray(phi, dtype=np.float64)
        dphi = np.ascontiguousarray(dphi, dtype=np.float64)
        dpsi = np.ascontiguousarray(dpsi, dtype=np.float64)
        if phi.ndim != 3 or dphi.shape != phi.shape + (self.gdim,) or dpsi.ndim != 4 or dpsi.shape[:2] != phi.shape[:2] \
                or dpsi.shape[3] != self.gdim:
            raise ValueError("facet tables: phi (nf, nq, ndofs), dphi (nf, nq, ndofs, gdim), dpsi (nf, nq, ngeom, gdim)")
        rc = self.ctx.lib.dxo_mesh_set_facet_tables(self.ctx._h, self._h, phi.shape[0], phi.shape[1], phi.ctypes.data,
                                                    dphi.ctypes.data, dpsi.ctypes.data)
        self.ctx.check(rc, "dxo_mesh_set_facet_tables")
        self.nq_facet = phi.shape[1]

    def evaluate_facets(self, kind: str, bs: int, u, entities, out=None) -> np.ndarray:
        """Host arrays in, host array (n_entities, nq_facet, value_size) out; entities (n, 2) = (cell, local facet)."""
        D = self.value_size(kind, bs)
        u = np.ascontiguousarray(_state(u), dtype=np.float64).reshape(-1)
        if u.size != self.num_field_nodes * bs:
            raise ValueError(f"field vector has {u.size} entries, expected {self.num_field_nodes * bs}")
        ents = np.ascontiguousarray(entities, dtype=np.int32)
        if ents.ndim != 2 or ents.shape[1] != 2:
            raise ValueError("codim-1 entities must have shape (n, 2): (cell, local facet)")
        if out is None:
            out = np.empty((ents.shape[0], getattr(self, "nq_facet", 0), D))
        rc = self.ctx.lib.dxo_eval_operand_facets(self.ctx._h, self._h, KINDS[kind], int(bs), MEM_HOST, u.ctypes.data,
                                                  ents.ctypes.data, ents.shape[0], out.ctypes.data)
        self.ctx.check(rc, "dxo_eval_operand_facets")
        return out

    def evaluate_device(self, kind: str, bs: int, u_ptr: int, n_cells: int, out_ptr: int, cells_ptr: int | None = None) -> None:
        """Raw device pointers, asynchronous on the context's stream."""
        rc = self.ctx.lib.dxo_eval_operand(self.ctx._h, self._h, KINDS[kind], int(bs), MEM_DEVICE, C.c_void_p(u_ptr),
                                           None if cells_ptr is None else C.c_void_p(cells_ptr), int(n_cells),
                                           C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_eval_operand")

    def operand(self, kind: str, field, bs: int | None = None, name: str | None = None, lazy: bool = False,
                snapshot: bool = True) -> "DeviceOperand":
        """lazy=True: `.eval()` over all cells returns a `LazyOperand` — an array-like that a fused consumer
        (`make_von_mises`) evaluates inside its own launch and that turns into the ndarray on `np.asarray`.
        snapshot (lazy only): True copies the field vector when the operand is "evaluated", so the value is the one at
        `evaluate_operands` time exactly as with `Expression.eval` (one host copy of the dof vector per call: 25-40 ms
        at 10^7 Q2 nodes). False keeps a reference to the live array instead — for the reference's own calling
        sequence, where `evaluate_external_operators` follows `evaluate_operands` at once (demo_plasticity_von_mises.py:
        445-456) and the field does not change in between."""
        if kind == "x":     # ufl.SpatialCoordinate(mesh): no field (pass None), codim-0 entities
            return DeviceOperand(self, "x", None, self.gdim, name or "x", False, snapshot)
        return DeviceOperand(self, kind, field, self.gdim if bs is None else bs, name or kind, lazy, snapshot)

    def von_mises(self, prm, u, sigma_n, p, C_tang, sigma, dp, mem: int = MEM_HOST) -> None:
        """dxo_von_mises_field: eps(u) + radial return + tangent in one launch (all cells of the mesh)."""
        def ptr(a):
            return a if isinstance(a, (int, np.integer)) or a is None else a.ctypes.data
        rc = self.ctx.lib.dxo_von_mises_field(self.ctx._h, C.byref(prm), self._h, int(mem), *(C.c_void_p(ptr(a)) for a in
                                              (u, sigma_n, p, C_tang, sigma, dp)))
        self.ctx.check(rc, "dxo_von_mises_field")

    def set_weights(self, weights) -> None:
        """Reference quadrature weights (nq values, `basix.make_quadrature(...)[1]`): needed by adjoint / tangent_apply."""
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != self.nq:
            raise ValueError(f"{w.size} weights for {self.nq} quadrature points")
        self.ctx.check(self.ctx.lib.dxo_mesh_set_weights(self.ctx._h, self._h, w.ctypes.data), "dxo_mesh_set_weights")


    def patch_info(self) -> dict:
        """The patches of the patch form of the internal force — an experiment that is NOT in the product library: the entry point
        (dxo_mesh_patch_info, scripts/exp/adjoint_patch_kernels.h) exists only in a -DDXO_EXPERIMENTS build (scripts/exp/build_variant.py,
        loaded through DXO_HIP_LIBRARY) and is bound here on demand."""
        import numpy as np

        fn = getattr(self.ctx.lib, "dxo_mesh_patch_info", None)
        if fn is None:
            raise DxoError("dxo_mesh_patch_info: the patch form is not part of this build of libdxo_hip.so (-DDXO_EXPERIMENTS only)")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]
        info = np.zeros(8, dtype=np.int64)
        self.ctx.check(fn(self.ctx._h, self._h, info.ctypes.data), "dxo_mesh_patch_info")
        keys = ("patches", "groups_per_wave", "wave_groups", "max_patch_nodes", "patch_nodes", "shared_nodes", "shared_entry_ppm", "waves_per_patch")
        out = dict(zip(keys, (int(x) for x in info)))
        out["max_wave_nodes"] = out["groups_per_wave"] >> 32
        out["groups_per_wave"] &= 0xFFFFFFFF
        out["schedule_fill"] = out["wave_groups"] / max(1, out["patches"] * out["waves_per_patch"] * out["groups_per_wave"])
        return out

    def adjoint(self, kind: str, bs: int, S_ptr: int, out_ptr: int, n_cells: int | None = None, cells_ptr: int | None = None) -> None:
        """out += sum_q w |det J| B^T S (DEVICE pointers): the assembled vector of inner(S, operand(v)) dx."""
        rc = self.ctx.lib.dxo_operand_adjoint(self.ctx._h, self._h, KINDS[kind], int(bs), C.c_void_p(S_ptr),
                                              None if cells_ptr is None else C.c_void_p(cells_ptr),
                                              self.num_cells if n_cells is None else int(n_cells), C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_operand_adjoint")

    def tangent_apply(self, C_tang_ptr: int, v_ptr: int, out_ptr: int) -> None:
        """out += K v with K = sum_q w |det J| B^T C_tang B never formed (DEVICE pointers, eps / Mandel, bs = gdim)."""
        rc = self.ctx.lib.dxo_tangent_apply(self.ctx._h, self._h, C.c_void_p(C_tang_ptr), C.c_void_p(v_ptr), C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_tangent_apply")

    def tangent_diagonal(self, C_tang_ptr: int, out_ptr: int) -> None:
        """out += diag(K) for the same K as tangent_apply (DEVICE pointers): Jacobi preconditioner."""
        rc = self.ctx.lib.dxo_tangent_diagonal(self.ctx._h, self._h, C.c_void_p(C_tang_ptr), C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_tangent_diagonal")

    def tangent_apply_vm(self, prm, sigma_ptr: int, dp_ptr: int, v_ptr: int, out_ptr: int) -> None:
        """out += K v with the von Mises consistent tangent formed per point from the operator's returned (sigma, dp) — no C_tang
        array (dxo_tangent_apply_vm; DEVICE pointers, e.g. VmState.pointers()): 56 instead of 288 bytes per point."""
        rc = self.ctx.lib.dxo_tangent_apply_vm(self.ctx._h, self._h, C.byref(prm), C.c_void_p(sigma_ptr), C.c_void_p(dp_ptr),
                                               C.c_void_p(v_ptr), C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_tangent_apply_vm")

    def tangent_diagonal_vm(self, prm, sigma_ptr: int, dp_ptr: int, out_ptr: int) -> None:
        """out += diag(K) from the returned (sigma, dp) (dxo_tangent_diagonal_vm)."""
        rc = self.ctx.lib.dxo_tangent_diagonal_vm(self.ctx._h, self._h, C.byref(prm), C.c_void_p(sigma_ptr), C.c_void_p(dp_ptr),
                                                  C.c_void_p(out_ptr))
        self.ctx.check(rc, "dxo_tangent_diagonal_vm")

    def von_mises_residual(self, prm, u_ptr: int, sigma_n_ptr: int, p_ptr: int, sigma_ptr: int, dp_ptr: int, R_ptr: int) -> None:
        """(sigma, dp) = von Mises return map of (eps(u), sigma_n, p) and R += sum_q w|J| B^T sigma in one call
        (dxo_von_mises_residual; DEVICE pointers): the residual evaluation of a Newton iteration of a matrix-free solve."""
        rc = self.ctx.lib.dxo_von_mises_residual(self.ctx._h, C.byref(prm), self._h, *(C.c_void_p(a) for a in
                                                 (u_ptr, sigma_n_ptr, p_ptr, sigma_ptr, dp_ptr, R_ptr)))
        self.ctx.check(rc, "dxo_von_mises_residual")

    def heat(self, A: float, B: float, T_dofs, q=None, dqdT=None, dqdsigma=None, mem: int = MEM_HOST) -> None:
        """dxo_heat_field: T and grad T of a scalar field + the heat-flux kernels in one launch (all cells)."""
        def ptr(a):
            return a if isinstance(a, (int, np.integer)) or a is None else a.ctypes.data
        rc = self.ctx.lib.dxo_heat_field(self.ctx._h, float(A), float(B), self._h, int(mem),
                                         *(None if a is None else C.c_void_p(ptr(a)) for a in (T_dofs, q, dqdT, dqdsigma)))
        self.ctx.check(rc, "dxo_heat_field")

    def close(self) -> None:
        if getattr(self, "_h", None) and self.ctx._h:
            self.ctx.lib.dxo_mesh_destroy(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceOperand:
    """Plays the role of a UFL operand in `evaluation.evaluate_operands`: `.eval(entities)` returns what
    `Expression.eval` returns, (len(entities), nq, *shape); for "F" the trailing shape is (gdim, gdim) like the
    tensor operand of the hyperelasticity demo, for "grad" of a vector field (bs, gdim)."""

    def __init__(self, mesh: DeviceMesh, kind: str, field, bs: int, name: str, lazy: bool = False, snapshot: bool = True):
        self.mesh, self.kind, self.field, self.bs, self.name, self.lazy, self.snapshot = mesh, kind, field, bs, name, lazy, snapshot
        self.eval_count = 0

    def eval(self, entities):
        self.eval_count += 1
        if self.lazy and (entities is None or (len(entities) == self.mesh.num_cells
                                               and np.array_equal(entities, np.arange(self.mesh.num_cells)))):
            if self.snapshot:
                u = np.array(_state(self.field), dtype=np.float64).reshape(-1)  # snapshot, like Expression.eval's result
            else:
                u = np.ascontiguousarray(_state(self.field), dtype=np.float64).reshape(-1)   # the live array (no copy if fp64)
            return LazyOperand(self.mesh, self.kind, self.bs, u)
        if self.kind == "x":
            if entities is not None and np.ndim(entities) == 2:
                raise NotImplementedError("the operand `x` is evaluated on cells; codim-1 entities go through the host array")
            return self.mesh.coordinate(entities)                                         # (n, nq, gdim), like any vector operand
        if entities is not None and np.ndim(entities) == 2:
            out = self.mesh.evaluate_facets(self.kind, self.bs, self.field, entities)    # (cell, local facet) pairs
        else:
            out = self.mesh.evaluate(self.kind, self.bs, self.field, entities)
        g = self.mesh.gdim
        if self.kind in ("F", "C"):
            return out.reshape(out.shape[0], out.shape[1], g, g)
        if self.kind == "grad" and self.bs > 1:
            return out.reshape(out.shape[0], out.shape[1], self.bs, g)
        if out.shape[2] == 1:
            return out.reshape(out.shape[0], out.shape[1])       # scalar operand is 2-D (heat demo, part2.py:220-228)
        return out

    def __repr__(self) -> str:
        return f"DeviceOperand({self.name})"


class LazyOperand:
    """The value of an operand over all cells, not yet computed: (mesh, kind, snapshot of the field vector).

    `make_von_mises` recognises it and runs dxo_von_mises_field (operand + return map in one launch, the strain
    never reaches memory). Anything else sees an array: `np.asarray(lazy)`, `.shape`, `.reshape` evaluate it once
    with dxo_eval_operand."""

    def __init__(self, mesh: DeviceMesh, kind: str, bs: int, u: np.ndarray):
        self.mesh, self.kind, self.bs, self.u = mesh, kind, bs, u
        self._value = None

    @property
    def shape(self):
        return (self.mesh.num_cells, self.mesh.nq, self.mesh.value_size(self.kind, self.bs))

    @property
    def dtype(self):
        return np.dtype(np.float64)

    def __array__(self, dtype=None, copy=None):
        if self._value is None:
            self._value = self.mesh.evaluate(self.kind, self.bs, self.u)
        return self._value if dtype is None else self._value.astype(dtype)

    def reshape(self, *shape):
        return np.asarray(self).reshape(*shape)
