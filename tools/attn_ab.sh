#!/bin/bash
# A/B of attention-kernel builds inside one run: tools/attn_ab.sh lib1.so lib2.so ...  (first entry "default" = the in-tree library)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then TAG=default python tools/attn_bench.py | tail -1; else SSI_HIP_LIB=$PWD/$lib TAG=$(basename $lib) python tools/attn_bench.py | tail -1; fi
  done
done
