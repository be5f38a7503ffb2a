#!/bin/bash
# Register / scratch use of every kernel in one source file: tools/kernel_regs.sh gemm_mfma.hip [extra flags] (no GPU needed)
B=/opt/rocm/lib/llvm/bin; src=$1; shift
cd "$(dirname "$0")/../speech-integration_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast --cuda-device-only "$@" -c $src -o /tmp/kr.co
$B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr.co --output=/tmp/kr.elf
$B/llvm-readelf --notes /tmp/kr.elf | grep -E "^\s+\.name:|private_segment_fixed_size|\.vgpr_count|agpr_count|vgpr_spill|sgpr_spill" | paste - - - - - - | sed 's/ \+/ /g; s/_ZN12_GLOBAL__N_1//'
