// Issue cost (cycles per wave64 instruction) of the vector instructions that bound the attention softmax and the GEMM epilogues on gfx950,
// measured on ONE wave: N independent instructions of one kind between two s_memtime reads.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o tools/micro/valu_rates ; run on an MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(ASM)                                                                                   \
    float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    typedef float f2 __attribute__((ext_vector_type(2)));                                             \
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};                                   \
    (void)p0; (void)p1; (void)p2; (void)p3;                                                           \
    uint64_t t0 = __builtin_readcyclecounter();                                                       \
    for (int it = 0; it < 64; ++it) { REP16(ASM) }                                                    \
    uint64_t t1 = __builtin_readcyclecounter();                                                       \
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];          \
    if (threadIdx.x == 0) cyc[0] = t1 - t0;

__global__ void k_fma(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_pkfma(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
}
__global__ void k_exp(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_rcp(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_max3(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_max3_f32 %0, %0, %0, %0\n v_max3_f32 %1, %1, %1, %1\n v_max3_f32 %2, %2, %2, %2\n v_max3_f32 %3, %3, %3, %3\n v_max3_f32 %4, %4, %4, %4\n v_max3_f32 %5, %5, %5, %5\n v_max3_f32 %6, %6, %6, %6\n v_max3_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_cvt(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0\n v_cvt_pk_bf16_f32 %1, %1, %1\n v_cvt_pk_bf16_f32 %2, %2, %2\n v_cvt_pk_bf16_f32 %3, %3, %3\n v_cvt_pk_bf16_f32 %4, %4, %4\n v_cvt_pk_bf16_f32 %5, %5, %5\n v_cvt_pk_bf16_f32 %6, %6, %6\n v_cvt_pk_bf16_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_nop(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");)
}
__global__ void k_exp_fma(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}
__global__ void k_exp_3fma(const float* in, float* out, uint64_t* cyc) {
    BODY(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %1, %1\n v_fma_f32 %7, %7, %7, %7\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
}


// chip-wide throughput: `blocks` workgroups of `waves` waves run the same loops; wave-instructions per CU per microsecond from HIP events
template <typename K> double thr(K kern, int blocks, int waves, int per_rep) {
    float *in, *out; uint64_t* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, in, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(in); hipFree(out); hipFree(cyc);
    return 20.0 * blocks * waves * 64.0 * 16 * per_rep / (ms * 1e3);  // wave-instructions per microsecond, whole chip
}
int main() {
    printf("chip-wide issue rate in wave64-instructions per CU per ns (256 CUs): a 16-lane SIMD at 2.4 GHz would give 4 SIMDs x 2.4 / 4 = 2.4\n");
    for (int wps : {1, 2, 4}) {   // waves per SIMD: blocks of 256 threads, wps blocks per CU
        const int blocks = 256 * wps * 4;  // 4 rounds
        printf("%d wave(s) per SIMD: fma %.2f  pk_fma %.2f  exp %.2f  rcp %.2f  max3 %.2f  cvt_pk %.2f  exp+fma %.2f  exp+3fma %.2f\n", wps,
               thr(k_fma, blocks, 4, 8) / 256e3, thr(k_pkfma, blocks, 4, 4) / 256e3, thr(k_exp, blocks, 4, 8) / 256e3, thr(k_rcp, blocks, 4, 8) / 256e3,
               thr(k_max3, blocks, 4, 8) / 256e3, thr(k_cvt, blocks, 4, 8) / 256e3, thr(k_exp_fma, blocks, 4, 8) / 256e3, thr(k_exp_3fma, blocks, 4, 8) / 256e3);
    }
    return 0;
}
