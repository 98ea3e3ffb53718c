#!/bin/bash
# Variants of libdxo_hip.so for A/B timing: same ABI, some translation units rebuilt with other -D macros.
# usage: bash scripts/exp/build_variants.sh   -> dolfinx_external_operator_amd/build_exp/libdxo_<name>.so
# run with: python scripts/exp/vmfield_ab.py (fused strain + return map) or
#           DXO_HIP_LIBRARY=.../libdxo_<name>.so python bench.py --no-cpu --no-e2e
set -e
cd "$(dirname "$0")/../.."
P=dolfinx_external_operator_amd
python -c "import sys; sys.path.insert(0,'.'); from dolfinx_external_operator_amd._build import build_library; build_library()"
mkdir -p $P/build_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -I$P/csrc"
UNITS="vm_field von_mises operand mohr_coulomb"
build() {  # name, macros...
  name=$1; shift
  objs=$(ls $P/build/*.o | grep -v "/vm_field.o\|/von_mises.o\|/operand.o\|/mohr_coulomb.o")
  for u in $UNITS; do hipcc $FLAGS "$@" -c $P/csrc/$u.hip -o $P/build_exp/${u}_$name.o & done
  wait
  for u in $UNITS; do objs="$objs $P/build_exp/${u}_$name.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build_exp/libdxo_$name.so $objs -ldl -lpthread
  echo built $name
}
build base                               # same flags as the product (sanity: must time like libdxo_hip.so)
# (a variant that prefetched the next tile's inputs in persistent grids was tried this way and dropped: no difference)
# (tile-walk variants of the persistent grid — XCD-contiguous eighths, runs of 4 tiles per wave — were tried this way and lost)
# round 3, tried this way and dropped (scripts/exp/vmfield_ab.py on ONE shared output block; scripts/bench_mc.py):
#   -DDXO_VMF_PRELOAD=1 (sigma_n / p requested before the strain is formed)      0.844 vs 0.844 ms (2 waves/SIMD), 1.00 with 3 (spills)
#   -DDXO_VMF_WAVES=2 / 4                                                         0.847 / 1.30 ms (3: 0.844)
#   -DDXO_VMF_BLOCKS_PER_CU=8 / 32 / 64                                           0.819 / 0.827 / 0.867 ms (16: 0.816)
#   -DDXO_MC_CLASSIFY_MINW=4 / 5 (mc_classify at <= 128 / <= 96 registers)        1.28-1.31 / 1.46 ms (1: 1.18-1.26)
# round 3, single persistent Mohr-Coulomb kernel (mc_variant 2): tiles classified per visit
for k in 0 1; do build mcpf$k -DDXO_MC_PREFETCH=$k; done
# instrumented Mohr-Coulomb kernel for scripts/exp/archive/mc_phase_profile.py (cycle counters per phase; its dlambda output is overwritten)
build mcprof -DDXO_MC_PROF=1
