#!/bin/bash
# A/B of compile-time choices on one box: libdvq_v_<name>.so = the product objects with the listed sources recompiled with extra flags
# usage: tools/lib_variant.sh name "flags" file1.hip [file2.hip ...]     then DVQ_LIBRARY=.../libdvq_v_<name>.so
set -e
cd "$(dirname "$0")/../dynamicvectorquantization_amd/csrc"
name=$1; flags=$2; shift 2
make -s all
repl=""
excl="\.tune|vq_assign_pipe"
for f in "$@"; do
  o=/tmp/v_${name}_${f%.hip}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off $flags -c $f -o $o
  repl="$repl $o"
  excl="$excl|^${f%.hip}\.o"
done
objs=$(ls *.o | grep -Ev "$excl" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=libdvq.map -o libdvq_v_$name.so $objs $repl
echo built libdvq_v_$name.so
