// What does one MFMA "gap" cost on gfx950 when the vector ALU works beside the matrix pipe?  One wave per SIMD (256-thread workgroup per CU,
// __launch_bounds__(256, 1)), a loop of gaps { v_mfma_f32_32x32x16_bf16 ; v_mul ; v_exp ; v_mul ; v_cvt_pk } as in the pipelined attention
// backward, by the register class of the MFMA's operands.  Cycles per gap from s_memtime (wave 0 of workgroup 0), every CU busy.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_gap.hip -o tools/micro/mfma_gap ; run on an MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define VALU4 "v_mul_f32 %[x0], %[c], %[x0]\n v_exp_f32 %[x1], %[x1]\n v_mul_f32 %[x2], %[x2], %[x3]\n v_cvt_pk_bf16_f32 %[x4], %[x5], %[x6]\n"
// the same arithmetic on two elements: scalar { fma fma exp exp mul mul cvt_pk } against packed { pk_fma exp exp pk_mul cvt_pk }
#define VALU_S2 "v_fma_f32 %[p0], %[p0], %[c], %[q0]\n v_fma_f32 %[p1], %[p1], %[c], %[q0]\n v_exp_f32 %[q2], %[p0]\n v_exp_f32 %[q3], %[p1]\n v_mul_f32 %[q2], %[q2], %[p2]\n v_mul_f32 %[q3], %[q3], %[p3]\n v_cvt_pk_bf16_f32 %[x4], %[q2], %[q3]\n"
#define VALU_P2 "v_pk_fma_f32 %[pp], %[pp], %[cc], %[qq]\n v_exp_f32 %[q2], %[p0]\n v_exp_f32 %[q3], %[p1]\n v_pk_mul_f32 %[rr], %[rr], %[pq]\n v_cvt_pk_bf16_f32 %[x4], %[q2], %[q3]\n"
#define VOPS [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6)

// MODE 0: A = v, B = a, C/D = v (S / dP products)   1: A = v, B = v, C/D = a (dV / dK products)   2: A = a, B = v, C/D = a
// MODE 3: A = v, B = v, C/D = v                      VALU: vector instructions beside the MFMA or not
template <int MODE, int VALU> __global__ __launch_bounds__(256, 1) void k(const float* in, float* out, uint64_t* cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[threadIdx.x] + r + i;
    u32x4 av = {threadIdx.x, 1u, 2u, 3u}, bv = {5u, threadIdx.x, 7u, 9u};
    u32x4 aa = av, ba = bv;
    asm volatile("" : "=a"(aa) : "0"(aa));
    asm volatile("" : "=a"(ba) : "0"(ba));
    if (MODE == 1 || MODE == 2)
        for (int i = 0; i < 4; ++i) asm volatile("" : "=a"(acc[i]) : "0"(acc[i]));
    float x0 = in[threadIdx.x], x1 = x0 * 0.5f, x2 = x0 + 2, x3 = x0 + 3, x4 = 0, x5 = x0 + 5, x6 = x0 + 6;
    const float c = -1.44269504f;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 pp = {x0, x1}, rr = {x2, x3}, cc = {c, c}, qq = {x5, x6}, pp2 = {x1, x0}, rr2 = {x3, x2};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(av), "a"(ba));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "v"(av), "v"(bv));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g & 3]) : "a"(aa), "v"(bv));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(av), "v"(bv));
            // MODE 4: no MFMA at all (an epilogue: the matrix pipe idle)
            if (VALU == 1) asm volatile(VALU4 : VOPS : [c] "v"(c));
            if (VALU == 2) asm volatile(VALU_S2 : [p0] "+v"(x0), [p1] "+v"(x1), [q2] "+v"(x2), [q3] "+v"(x3), [x4] "+v"(x4) : [c] "v"(c), [q0] "v"(x5), [p2] "v"(x5), [p3] "v"(x6));
            if (VALU == 3) asm volatile(VALU_P2 : [pp] "+v"(pp), [rr] "+v"(rr), [q2] "+v"(x2), [q3] "+v"(x3), [x4] "+v"(x4) : [cc] "v"(cc), [qq] "v"(qq), [pq] "v"(qq), [p0] "v"(x0), [p1] "v"(x1));
            if (VALU == 4) asm volatile("v_pk_fma_f32 %[pp], %[pp], %[cc], %[qq]\n v_pk_mul_f32 %[rr], %[rr], %[qq]\n v_pk_fma_f32 %[p2], %[p2], %[cc], %[qq]\n v_pk_mul_f32 %[r2], %[r2], %[qq]\n" : [pp] "+v"(pp), [rr] "+v"(rr), [p2] "+v"(pp2), [r2] "+v"(rr2) : [cc] "v"(cc), [qq] "v"(qq));
            if (VALU == 5) asm volatile("v_fma_f32 %[p0], %[p0], %[c], %[q0]\n v_mul_f32 %[p1], %[p1], %[q0]\n v_fma_f32 %[q2], %[q2], %[c], %[q0]\n v_mul_f32 %[q3], %[q3], %[q0]\n" : [p0] "+v"(x0), [p1] "+v"(x1), [q2] "+v"(x2), [q3] "+v"(x3) : [c] "v"(c), [q0] "v"(x5));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n s_nop 15");
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + pp[0] + pp[1] + rr[0] + rr[1] + pp2[0] + pp2[1] + rr2[0] + rr2[1];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// The persistent GEMM's stream: v_mfma_f32_16x16x32_bf16 (16 cycles of matrix pipe each) on 16 independent accumulators in accumulation
// registers, NV vector instructions (mul / fma on 8 independent registers) behind every MFMA: how many fit before the stream slows down?
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NV, bool TRANS> __global__ __launch_bounds__(256, 1) void k16(const float* in, float* out, uint64_t* cyc, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = in[threadIdx.x] + r + i;
    for (int i = 0; i < 16; ++i) asm volatile("" : "=a"(acc[i]) : "0"(acc[i]));
    u32x4 av = {threadIdx.x, 1u, 2u, 3u}, bv = {5u, threadIdx.x, 7u, 9u};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x] + i;
    const float c = 1.0001f;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(av), "v"(bv));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int r = (g * NV + v) & 7;
                if (TRANS && v == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[r]) : "v"(c));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / 2;  // (run() divides by 8 gaps per iteration: 16 here)
}

template <typename K> double run(K kern, const char* name) {
    float *in, *out; uint64_t* cyc;
    const int iters = 2000;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 4096);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(in); hipFree(out); hipFree(cyc);
    const double per = (double)h / (iters * 8.0);
    printf("%-46s %.1f cycles per gap\n", name, per);
    return per;
}
int main() {
    run(k<0, 0>, "A=v B=a C/D=v, MFMA alone");
    run(k<1, 0>, "A=v B=v C/D=a, MFMA alone");
    run(k<0, 1>, "A=v B=a C/D=v, + mul exp mul cvt_pk");
    run(k<1, 1>, "A=v B=v C/D=a, + mul exp mul cvt_pk");
    run(k<2, 1>, "A=a B=v C/D=a, + mul exp mul cvt_pk");
    run(k<3, 1>, "A=v B=v C/D=v, + mul exp mul cvt_pk");
    run(k<0, 2>, "A=v B=a C/D=v, + fma fma exp exp mul mul cvt_pk");
    run(k<0, 3>, "A=v B=a C/D=v, + pk_fma exp exp pk_mul cvt_pk");
    run(k<0, 4>, "A=v B=a C/D=v, + pk_fma pk_mul pk_fma pk_mul");
    run(k<0, 5>, "A=v B=a C/D=v, + fma mul fma mul");
    run(k<4, 4>, "no MFMA: pk_fma pk_mul pk_fma pk_mul (8 elements)");
    run(k<4, 5>, "no MFMA: fma mul fma mul (4 elements)");
    run(k<4, 2>, "no MFMA: fma fma exp exp mul mul cvt_pk");
    run(k<4, 3>, "no MFMA: pk_fma exp exp pk_mul cvt_pk");
    run(k16<0, false>, "16x16x32 stream, MFMA alone");
    run(k16<1, false>, "16x16x32 stream + 1 fma per MFMA");
    run(k16<2, false>, "16x16x32 stream + 2 fma per MFMA");
    run(k16<3, false>, "16x16x32 stream + 3 fma per MFMA");
    run(k16<1, true>, "16x16x32 stream + 1 exp per MFMA");
    run(k16<2, true>, "16x16x32 stream + exp fma per MFMA");
    return 0;
}
