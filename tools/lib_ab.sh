#!/bin/bash
# In-run A/B of library builds: tools/lib_ab.sh "<python command>" default variants/libssi_a.so ...   (two passes; "default" = in-tree library)
cmd=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (pass $rep)"
    if [ "$lib" = default ]; then $cmd; else SSI_HIP_LIB=$PWD/$lib $cmd; fi
  done
done
