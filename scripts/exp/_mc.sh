for k in 0 1; do echo "prefetch $k"; DXO_HIP_LIBRARY=$PWD/dolfinx_external_operator_amd/build_exp/libdxo_mcpf$k.so python scripts/exp/mc_fraction.py --fractions 0,0.31,1 2>/dev/null; done
