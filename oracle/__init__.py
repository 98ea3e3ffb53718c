"""CPU parity oracle — TEST INFRASTRUCTURE ONLY.

May be imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package `dolfinx_external_operator_amd` never imports this module.
"""
from .loader import OracleLib, build_oracle, load_oracle  # noqa: F401
