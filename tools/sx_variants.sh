#!/bin/bash
# build variants of the staged attention kernel (self_attn.hip with -D switches) next to the real library: build/sx/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
make -C transcar_amd/csrc -j8 >/dev/null
mkdir -p build/sx
OBJS=$(ls build/hip/*.o | grep -v self_attn.o)
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math $defs -c transcar_amd/csrc/self_attn.hip -o build/sx/self_attn_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/sx/lib_$name.so $OBJS build/sx/self_attn_$name.o
  echo built build/sx/lib_$name.so
done
