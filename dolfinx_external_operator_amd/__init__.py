"""dolfinx_external_operator_amd — MI355X (gfx950) kernels behind dolfinx-external-operator's
`FEMExternalOperator.external_function` callback.

Product path: `operators.make_*` -> ctypes (`_lib`) -> libdxo_hip.so (hand-written HIP, include/dxo.h).
There is no CPU fallback; `oracle/` is test infrastructure and is never imported from here.
"""
from ._lib import GATHER_COMPACT, GATHER_COMPACT_DIRECT, GATHER_COMPACT_PIPELINED, GATHER_FULL, GATHER_NONE, MEM_DEVICE, MEM_HOST, AssignDesc, AssignPlan, Context, MultiGpu, DxoError, IsiharaParams, McParams, VmParams, default_context, load_library
from .evaluation import (
    Coefficient,
    MixedExternalOperator,
    Operand,
    QuadratureExternalOperator,
    evaluate_external_operators,
    evaluate_operands,
    get_unrolled_dofmap,
)
from .device_assign import DeviceAssigner
from .operand_eval import DeviceMesh, DeviceOperand, LazyOperand
from .operators import make_conductivity, make_heat, make_icnn, make_isihara, make_mohr_coulomb, make_von_mises, von_mises_commit_state

__version__ = "0.1.0"

__all__ = [
    "AssignPlan", "DeviceAssigner",
    "Context", "DxoError", "VmParams", "MEM_HOST", "MEM_DEVICE", "default_context", "load_library",
    "make_von_mises", "make_heat", "make_conductivity", "make_mohr_coulomb", "make_icnn", "make_isihara", "McParams", "IsiharaParams", "von_mises_commit_state",
    "QuadratureExternalOperator", "MixedExternalOperator", "Operand", "Coefficient",
    "evaluate_operands", "evaluate_external_operators", "get_unrolled_dofmap", "DeviceMesh", "DeviceOperand", "LazyOperand", "AssignDesc",
    "MultiGpu", "GATHER_NONE", "GATHER_FULL", "GATHER_COMPACT", "GATHER_COMPACT_DIRECT", "GATHER_COMPACT_PIPELINED",
]
