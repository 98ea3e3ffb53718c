#!/bin/bash
# Variants of the fused strain + return-map kernel (vm_field) for A/B timing: same ABI, other -D macros.
# usage: bash scripts/exp/build_variants.sh   -> dolfinx_external_operator_amd/build_exp/libdxo_<name>.so
# run one with: DXO_HIP_LIBRARY=.../libdxo_<name>.so python scripts/bench_operand.py --case 0 --operand-cell 0
set -e
cd "$(dirname "$0")/../.."
P=dolfinx_external_operator_amd
mkdir -p $P/build_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -I$P/csrc"
OTHERS=$(ls $P/build/*.o | grep -v "vm_field.o\|operand.o")
build() {  # name, macros...
  name=$1; shift
  hipcc $FLAGS "$@" -c $P/csrc/vm_field.hip -o $P/build_exp/vm_field_$name.o &
  hipcc $FLAGS "$@" -c $P/csrc/operand.hip -o $P/build_exp/operand_$name.o &
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build_exp/libdxo_$name.so $OTHERS $P/build_exp/vm_field_$name.o $P/build_exp/operand_$name.o -ldl -lpthread
  echo built $name
}
build rt3   -DDXO_OP_CT=0 -DDXO_OP_UNROLL=3
build ct3   -DDXO_OP_CT=1 -DDXO_OP_UNROLL=3
build ct9   -DDXO_OP_CT=1 -DDXO_OP_UNROLL=9
build ct27  -DDXO_OP_CT=1 -DDXO_OP_UNROLL=27
build ct9w2 -DDXO_OP_CT=1 -DDXO_OP_UNROLL=9 -DDXO_VMF_WAVES=2
build ct9w4 -DDXO_OP_CT=1 -DDXO_OP_UNROLL=9 -DDXO_VMF_WAVES=4
