#!/bin/bash
# Build a variant of the library for an in-run A/B: tools/variant.sh <name> <file.hip> <extra hipcc flags...>  ->  variants/libssi_<name>.so
# (only <file.hip> is recompiled with the flags; the other objects are the in-tree ones).  Use with SSI_HIP_LIB=$PWD/variants/libssi_<name>.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../speech-integration_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -ffp-contract=fast "$@" -c $src -o /tmp/variant_$name.o
objs=""
for f in api elementwise embed_ce gemm_generic gemm_mfma attention_generic attention_mfma; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libssi_$name.so $objs
echo "variants/libssi_$name.so"
