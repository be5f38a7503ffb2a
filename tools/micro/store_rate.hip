// How fast does a CU get rid of an output tile?  Every workgroup (one per CU, 256 threads) writes a 256 x 256 bf16 tile = 128 KB with
// 16-B-per-lane stores, 32 per wave, then waits for them (s_waitcnt vmcnt(0)); cycles of wave 0 of workgroup 0 from the first store to the
// end of the wait, all 256 CUs storing at once (as the persistent GEMM's epilogues do).
//   MAP 0: the GEMM epilogue's map — a store covers 8 rows x 128 B (whole lines), rows ldc apart
//   MAP 1: one contiguous KiB per store (lane * 16)
//   MAP 2 / 3 / 4: a store covers 4 rows x 256 B / 2 rows x 512 B / 16 rows x 64 B of the tile
//   AUX  : cache-policy bits of the buffer store (0 default, 2 nt, 16 sc1, 17 sc0 sc1)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_rate.hip -o tools/micro/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MAP, int AUX, int GAP> __global__ __launch_bounds__(256, 1) void k(char* out, int64_t ldc_bytes, uint64_t* cyc, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000u);
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    uint64_t tot = 0;
    for (int r = 0; r < reps; ++r) {
        // tile (blockIdx.x, r): rows (blockIdx % 64) * 256 .., columns ((blockIdx / 64) * 4 + (r & 3)) * 512 B
        const int64_t tile = (int64_t)(blockIdx.x % 64) * 256 * ldc_bytes + ((blockIdx.x / 64) * 4 + (r & 3)) * 512;
        __syncthreads();
        const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            int off;
            if (MAP == 0) off = (int)(tile + ((wave >> 1) * 128 + (i >> 1) * 8 + (lane & 7)) * ldc_bytes + (wave & 1) * 256 + (i & 1) * 128 + (lane >> 3) * 16);
            else if (MAP == 1) off = (int)(tile + ((wave * 32 + i) * 2) * ldc_bytes + lane * 16);   // (two rows' worth, contiguous: a 1-KiB run)
            else if (MAP == 2) off = (int)(tile + (wave * 64 + (i >> 1) * 4 + (lane >> 4)) * ldc_bytes + (i & 1) * 256 + (lane & 15) * 16);
            else if (MAP == 3) off = (int)(tile + (wave * 64 + i * 2 + (lane >> 5)) * ldc_bytes + (lane & 31) * 16);
            else off = (int)(tile + ((wave >> 1) * 128 + (i >> 2) * 16 + (lane & 15)) * ldc_bytes + (wave & 1) * 256 + (i & 3) * 64 + (lane >> 4) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
            if (GAP) __builtin_amdgcn_s_sleep(GAP);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += __builtin_readcyclecounter() - t0;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = tot / reps;
}
template <typename K> void run(K kern, const char* name, int grid) {
    char* out; uint64_t* cyc;
    const int64_t ldc = 16384 * 2;  // bytes per output row (N = 16384 bf16)
    hipMalloc(&out, (size_t)16384 * ldc); hipMalloc(&cyc, 8);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, ldc, cyc, 8);
    hipDeviceSynchronize();
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s %7llu cycles per 128 KB tile  (%.1f B/clk per CU)\n", name, (unsigned long long)h, 131072.0 / h);
    hipFree(out); hipFree(cyc);
}
int main() {
    run(k<0, 0, 0>, "GEMM map, default policy, 256 CUs", 256);
    run(k<0, 0, 0>, "GEMM map, default policy, 32 CUs (one per 8)", 32);
    run(k<0, 0, 0>, "GEMM map, default policy, 1 CU", 1);
    run(k<1, 0, 0>, "contiguous KiB per store, default, 256 CUs", 256);
    run(k<2, 0, 0>, "4 rows x 256 B per store, default, 256 CUs", 256);
    run(k<3, 0, 0>, "2 rows x 512 B per store, default, 256 CUs", 256);
    run(k<4, 0, 0>, "16 rows x 64 B per store, default, 256 CUs", 256);
    run(k<0, 0, 0>, "GEMM map, default policy, 256 CUs (again)", 256);
    run(k<0, 2, 0>, "GEMM map, nt, 256 CUs", 256);
    run(k<0, 16, 0>, "GEMM map, sc1, 256 CUs", 256);
    run(k<0, 17, 0>, "GEMM map, sc0 sc1, 256 CUs", 256);
    run(k<0, 0, 8>, "GEMM map, default, s_sleep 8 between stores, 256 CUs", 256);
    return 0;
}
