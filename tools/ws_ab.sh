#!/bin/bash
# A/B timings of gemm_ws.hip builds on ONE box: tools/ws_ab.sh "<-D flags A>" "<-D flags B>" ...
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed $flags -shared \
    -o /tmp/libcgnn_ab_$i.so connectome_gnn_amd/csrc/*.hip || exit 1
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    echo "== [$flags]"
    CGNN_LIB=/tmp/libcgnn_ab_$i.so python ${AB_SCRIPT:-tools/ws_times.py} 2>/dev/null
    i=$((i+1))
  done
done
