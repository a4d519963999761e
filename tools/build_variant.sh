#!/bin/bash
# build_variant.sh <name> <file.hip> [extra -D flags]: tools/tmpv/libdvq_<name>.so = libdvq.so with <file.hip> recompiled with the flags
# (A/B of compile-time choices on one GPU box; select with DVQ_LIBRARY=<path>).  The other objects are the product build's.
set -e
C=/root/repo/dynamicvectorquantization_amd/csrc
n=$1; f=$2; shift 2
mkdir -p /root/repo/tools/tmpv   # (git-ignored: *.so)
objs=""
for o in dvq_abi vq_fold vq_assign_exact vq_assign_filter vq_assign_routed route_select permute entropy_map ema_update router_gate exchange qconv vq_backward; do
  if [ "$o.hip" = "$f" ]; then
    SRC=${VARIANT_SRC:-$C/$f}; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I$C "$@" -c $SRC -o /tmp/var_${n}_$o.o
    objs="$objs /tmp/var_${n}_$o.o"
  else
    objs="$objs $C/$o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/libdvq.map -o /root/repo/tools/tmpv/libdvq_$n.so $objs
echo built /root/repo/tools/tmpv/libdvq_$n.so
