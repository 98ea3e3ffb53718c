"""DOLFINx-free mirror of the reference's evaluation entry points.

The GPU box has no DOLFINx/UFL, and on a DOLFINx installation the reference's own
`evaluate_operands` / `evaluate_external_operators` are used unchanged with the factories of
`operators.py`. This module restates the *value side* of those two functions
(src/dolfinx_external_operator/external_operator.py:338-448 and the assigners :286-335) on plain
NumPy so the boundary semantics — operand de-duplication, nested operators, the tuple rule, the
flat contiguous assignment, the "return everything" rule — can be exercised without UFL.

Names and attribute names follow the reference (`ufl_operands`, `derivatives`, `external_function`,
`ref_coefficient.x.array`, `_assign_func`) so a test written against the reference reads the same here.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def get_unrolled_dofmap(dofmap_list: np.ndarray, bs: int) -> np.ndarray:
    """Blocked dofmap -> flat scalar dof indices (external_operator.py:18-26)."""
    dofmap_list = np.asarray(dofmap_list)
    if dofmap_list.shape[0] == 0:
        return np.empty((0,), dtype=dofmap_list.dtype)
    # blocked node k owns the scalar dofs k*bs .. k*bs + bs - 1, in the node's position in its cell
    components = np.arange(bs, dtype=dofmap_list.dtype)
    return (dofmap_list[:, :, None] * bs + components).reshape(-1)


class _Vector:
    """Stands in for `fem.Function.x`: `.array` plus a `scatter_forward()` hook (:445)."""

    def __init__(self, array: np.ndarray, scatter: Callable[[np.ndarray], None] | None = None):
        self.array = array
        self._scatter = scatter
        self.scatter_count = 0

    def scatter_forward(self) -> None:
        self.scatter_count += 1
        if self._scatter is not None:
            self._scatter(self.array)


class Coefficient:
    """Stands in for the `fem.Function` an operator writes into (`ref_coefficient`, :214-218)."""

    def __init__(self, size: int, dtype=np.float64, name: str | None = None, scatter=None):
        self.x = _Vector(np.zeros(size, dtype=dtype), scatter)
        self.name = name
        self.dtype = np.dtype(dtype)


class Operand:
    """An evaluable operand: the value side of `fem.Expression(operand, eval_points).eval(mesh, entities)`
    (:393-402). `fn(entities) -> ndarray` of shape (len(entities), nq, *shape). Hashable by identity, like
    a UFL expression is by structure."""

    def __init__(self, fn: Callable[[np.ndarray], np.ndarray], name: str = "operand"):
        self.fn = fn
        self.name = name
        self.eval_count = 0

    def eval(self, entities: np.ndarray) -> np.ndarray:
        self.eval_count += 1
        return self.fn(entities)

    def __repr__(self) -> str:
        return f"Operand({self.name})"


class QuadratureExternalOperator:
    """Value-side stand-in of `FEMExternalOperator` (:49-335) for Quadrature/DG (contiguous) spaces and,
    optionally, dofmap-scattered spaces.

    num_cells, num_points : local cells (owned + ghosts) and interpolation points per cell
    value_shape           : shape of the operator value at a point, INCLUDING the derivative axes
                            (shape(N) + shape(operand) per derivative, :108-110)
    unrolled_dofmap       : None -> `_assign_non_mixed_contiguous` (:289-290);
                            index array -> `_assign_non_mixed` (:286-287)
    """

    def __init__(self, *operands, num_cells: int, num_points: int, value_shape: Sequence[int] = (),
                 external_function=None, derivatives: tuple[int, ...] | None = None, name: str | None = None,
                 coefficient: Coefficient | None = None, dtype=np.float64, unrolled_dofmap=None,
                 coefficient_size: int | None = None):
        self.ufl_operands = tuple(operands)
        self.derivatives = tuple(derivatives) if derivatives is not None else (0,) * len(operands)
        self.num_cells = int(num_cells)
        self.num_points = int(num_points)
        self.value_shape = tuple(value_shape)
        self.name = name
        size = self.num_cells * self.num_points * int(np.prod(self.value_shape, dtype=np.int64))
        if coefficient_size is not None:
            size = coefficient_size
        if coefficient is not None:
            if coefficient.x.array.size != size:
                raise TypeError("The provided coefficient must be defined on the same function space as the operator.")
            self.ref_coefficient = coefficient
        else:
            self.ref_coefficient = Coefficient(size, dtype=dtype, name=name)
        self.external_function = external_function
        self.unrolled_dofmap = None if unrolled_dofmap is None else np.asarray(unrolled_dofmap)
        self._assign_func = (
            self._assign_non_mixed_contiguous if self.unrolled_dofmap is None else self._assign_non_mixed
        )
        self._full_cells = None

    def _assign_non_mixed(self, values: np.ndarray) -> None:
        self.ref_coefficient.x.array[self.unrolled_dofmap] = values          # :287

    def _assign_non_mixed_contiguous(self, values: np.ndarray) -> None:
        self.ref_coefficient.x.array[:] = values                             # :290


class MixedExternalOperator(QuadratureExternalOperator):
    """Value-side stand-in of `FEMExternalOperator` on a MIXED function space (external_operator.py:139-198):
    the evaluation points of all subspaces are concatenated, the kernel returns one padded array with
    `comp_size = max(val_sizes)` components per point, and the assigner scatters each subspace's slice through
    that subspace's dofmap (`_assign_mixed_2d` :292-311 when every subspace is scalar, `_assign_mixed_3d`
    :313-335 otherwise).

    subspaces: sequence of dicts {"n_pts": interpolation points of the subspace, "val_size": components per point,
               "dofmap": int array (num_cells, n_pts * val_size) of indices into the mixed coefficient vector}
    """

    def __init__(self, *operands, num_cells: int, subspaces, coefficient_size: int, external_function=None,
                 derivatives: tuple[int, ...] | None = None, name: str | None = None, dtype=np.float64):
        val_sizes = [int(sp["val_size"]) for sp in subspaces]
        self._comp_size = max(val_sizes) if val_sizes else 1                       # :161
        self._mixed_subspace_info = []
        offset = 0
        for i, sp in enumerate(subspaces):
            dofmap = np.asarray(sp["dofmap"])
            info = {"n_pts": int(sp["n_pts"]), "val_size": int(sp["val_size"]), "dofs_per_cell": dofmap.shape[1],
                    "flat_dofs": dofmap.flatten(), "offset": offset}              # :180-190
            if self._comp_size < info["val_size"]:
                raise ValueError(f"Unsupported mixed element layout for subspace {i}")   # :173-178
            self._mixed_subspace_info.append(info)
            offset += info["n_pts"]
        self._n_points_total = offset                                              # :192
        super().__init__(*operands, num_cells=num_cells, num_points=self._n_points_total,
                         value_shape=(self._comp_size,) if self._comp_size > 1 else (), external_function=external_function,
                         derivatives=derivatives, name=name, dtype=dtype, coefficient_size=coefficient_size)
        self._is_mixed = True
        self._assign_func = self._assign_mixed_2d if self._comp_size == 1 else self._assign_mixed_3d   # :195-198

    def _assign_mixed_2d(self, values: np.ndarray) -> None:
        coeff = self.ref_coefficient
        if values.ndim == 1:
            values = values.reshape(values.size // self._n_points_total, self._n_points_total)   # :298-300
        for info in self._mixed_subspace_info:
            block = values[:, info["offset"]: info["offset"] + info["n_pts"]]                    # :310
            coeff.x.array[info["flat_dofs"]] = block.reshape(-1)                                  # :311

    def _assign_mixed_3d(self, values: np.ndarray) -> None:
        coeff = self.ref_coefficient
        if values.ndim == 1:
            n_cells = values.size // (self._n_points_total * self._comp_size)                     # :320
            values = values.reshape(n_cells, self._n_points_total, self._comp_size)
        n_cells = values.shape[0]
        for info in self._mixed_subspace_info:
            chunk = values[:, info["offset"]: info["offset"] + info["n_pts"], :]                  # :330
            block = chunk[:, :, : info["val_size"]].reshape(n_cells, info["dofs_per_cell"])       # :333
            coeff.x.array[info["flat_dofs"]] = block.reshape(-1)                                  # :335


def _all_local_cells(op) -> np.ndarray:
    cells = getattr(op, "_full_cells", None)
    if cells is None:
        cells = np.arange(0, op.num_cells, dtype=np.int32)
        op._full_cells = cells
    return cells


def evaluate_operands(external_operators, entities: np.ndarray | None = None) -> dict:
    """Value-side restatement of `evaluate_operands` (:338-404).

    Each *unique* operand is evaluated once (result dict keyed by the operand, :374-403); an operand that
    is itself an operator recurses and stores the nested dict (:383-384); `entities` defaults to all
    local cells, cached on the first operator (:365-371); an empty list gives `{}` (:356-357).
    """
    if not external_operators:
        return {}
    if entities is None:
        entities = _all_local_cells(external_operators[0])
    table: dict = {}
    for op in external_operators:
        for operand in op.ufl_operands:
            if operand in table:
                continue
            if isinstance(operand, QuadratureExternalOperator):
                table[operand] = evaluate_operands([operand], entities)
            else:
                table[operand] = operand.eval(entities)
    return table


def _operand_values(op, evaluated_operands: dict) -> list:
    """Operand arrays of one operator, in operand order; nested operators are evaluated first and
    contribute their own result (:425-430)."""
    values = []
    for operand in op.ufl_operands:
        if isinstance(operand, QuadratureExternalOperator):
            values += evaluate_external_operators([operand], evaluated_operands[operand])
        else:
            values.append(evaluated_operands[operand])
    return values


def evaluate_external_operators(external_operators, evaluated_operands: dict) -> list:
    """Value-side restatement of `evaluate_external_operators` (:407-448).

    Per operator: call `external_function(derivatives)(*operand_arrays)` (:432); a tuple result assigns
    its FIRST entry to the coefficient (:435-438) through the operator's assigner (:441) — a wrong size
    raises ValueError exactly like `x.array[:] = values` does in the reference (:440-444);
    `scatter_forward()` follows (:445); the WHOLE result is appended so extra outputs (sigma, dp) reach
    the caller (:446).
    """
    results = []
    for op in external_operators:
        kernel = op.external_function(op.derivatives)
        outcome = kernel(*_operand_values(op, evaluated_operands))
        op._assign_func(outcome[0] if type(outcome) is tuple else outcome)
        op.ref_coefficient.x.scatter_forward()
        results.append(outcome)
    return results


__all__ = [
    "Coefficient", "Operand", "QuadratureExternalOperator", "MixedExternalOperator", "evaluate_operands",
    "evaluate_external_operators", "get_unrolled_dofmap",
]
