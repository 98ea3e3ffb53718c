"""The operator's assigner on DEVICE arrays.

`FEMExternalOperator` picks `_assign_func` once, in its constructor (external_operator.py:195-209): a plain copy for
Quadrature / DG spaces (`_assign_non_mixed_contiguous` :289-290), `coeff.x.array[unrolled_dofmap] = values` for any other
single space (`_assign_non_mixed` :286-287), one scatter per subspace on a mixed space (`_assign_mixed_2d` :292-311,
`_assign_mixed_3d` :313-335); `evaluate_external_operators` then calls it with the kernel's values (:441). When the kernel's
values and the coefficient both live on the GPU, `DeviceAssigner(op, ctx)` is that same choice made once for the device:
one `dxo_assign_plan` per dofmap (the ownership pass of NumPy's last-writer rule done at construction), and

    assigner.apply(values_ptr, coeff_ptr)

leaves in `coeff` the bits `op._assign_func(values)` leaves in `op.ref_coefficient.x.array` — for float32, float64 and
complex128 coefficients (the types `test/test_multiaction.py:15-23` runs). Nothing here computes on the values; a missing
HIP library makes the constructor raise (there is no host fallback: the host assigners are the operator's own).
"""
from __future__ import annotations

import numpy as np

from ._lib import AssignDesc, Context

_COPY_H2D, _COPY_D2D = 0, 2


class DeviceAssigner:
    """`op._assign_func` for device pointers. `op` is a `QuadratureExternalOperator` / `MixedExternalOperator` (or any object
    with the reference's attributes: `ref_coefficient.x.array`, and `_mixed_subspace_info`, `_n_points_total`, `_comp_size`
    on mixed spaces or `unrolled_dofmap` on a dofmap-scattered one)."""

    def __init__(self, op, ctx: Context):
        coeff = op.ref_coefficient.x.array
        self.ctx = ctx
        self.dtype = coeff.dtype
        self.coeff_size = int(coeff.size)
        self.elem_bytes = int(coeff.dtype.itemsize)
        if self.elem_bytes not in (4, 8, 16):
            raise TypeError(f"DeviceAssigner: scalar type {coeff.dtype} is not 4, 8 or 16 bytes wide")
        self._plans = []
        self.kind = "contiguous"
        self.values_size = self.coeff_size
        if getattr(op, "_is_mixed", False):
            self.kind = "mixed_2d" if op._comp_size == 1 else "mixed_3d"
            n_cells = int(op.num_cells)
            self.values_size = n_cells * op._n_points_total * op._comp_size
            for info in op._mixed_subspace_info:
                desc = AssignDesc(n_cells, info["n_pts"], info["val_size"], info["offset"], op._n_points_total, op._comp_size, self.elem_bytes)
                self._plans.append(self._plan(desc, info["flat_dofs"]))
        elif getattr(op, "unrolled_dofmap", None) is not None:
            self.kind = "non_mixed"
            dofs = np.asarray(op.unrolled_dofmap).reshape(-1)
            n_cells, n_pts = int(op.num_cells), int(op.num_points)
            if n_cells * n_pts == 0 or dofs.size % (n_cells * n_pts):
                raise ValueError("DeviceAssigner: the unrolled dofmap does not hold a whole number of entries per (cell, point)")
            bs = dofs.size // (n_cells * n_pts)
            self.values_size = dofs.size
            self._plans.append(self._plan(AssignDesc(n_cells, n_pts, bs, 0, n_pts, bs, self.elem_bytes), dofs))

    def _plan(self, desc: AssignDesc, flat_dofs):
        """The dofmap goes to the device for the plan's construction only (dxo_assign_plan_create does not keep it)."""
        dofs = np.ascontiguousarray(flat_dofs, dtype=np.int32).reshape(-1)
        if dofs.size == 0:
            return self.ctx.assign_plan(desc, None, self.coeff_size)
        d_dofs = self.ctx.device_alloc(dofs.nbytes)
        try:
            self.ctx.copy(d_dofs, dofs, dofs.nbytes, _COPY_H2D)
            return self.ctx.assign_plan(desc, d_dofs, self.coeff_size)
        finally:
            self.ctx.device_free(d_dofs)

    def apply(self, values_ptr: int, coeff_ptr: int) -> None:
        """values: `values_size` scalars of the coefficient's type in the kernel's output layout; coeff: `coeff_size` scalars.
        Contiguous spaces: the operator normally writes in place (values_ptr == coeff_ptr: nothing to do), otherwise one copy."""
        if not self._plans:
            if int(values_ptr) != int(coeff_ptr) and self.coeff_size:
                self.ctx.copy(int(coeff_ptr), int(values_ptr), self.coeff_size * self.elem_bytes, _COPY_D2D)
            return
        for plan in self._plans:
            plan.apply(int(values_ptr), int(coeff_ptr))

    def forms(self) -> list[dict]:
        """dxo_assign_plan_form of every plan (which of its two orders a large plan settled on)."""
        return [p.form() for p in self._plans]

    def close(self) -> None:
        for p in self._plans:
            p.close()
        self._plans = []
