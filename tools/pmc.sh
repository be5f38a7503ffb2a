#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <python script>
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d gpurun_out/$out -- python "$@" > gpurun_out/$out.log 2>&1
tail -1 gpurun_out/$out.log | cut -c1-200
