#!/bin/bash
# A/B of compile-time choices of router_gate.hip on one box: builds libdvq_gv_<name>.so (product objects + a variant router_gate object)
# usage: tools/gate_variants.sh name "flags" [source]      (source defaults to router_gate.hip; run from anywhere)
set -e
cd "$(dirname "$0")/../dynamicvectorquantization_amd/csrc"
name=$1; flags=$2; src=${3:-router_gate.hip}
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off $flags -c $src -o /tmp/rg_$name.o
objs=$(ls *.o | grep -v '\.tune' | grep -v '^router_gate' | grep -v vq_assign_pipe | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=libdvq.map -o libdvq_gv_$name.so $objs /tmp/rg_$name.o
echo built libdvq_gv_$name.so
