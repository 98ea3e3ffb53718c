"""Helpers for the tests, benches and examples that are NOT part of the product path: DOLFINx-free element tables and
structured meshes (`tools.synthetic`)."""
