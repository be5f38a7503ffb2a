// Generic GEMM: any shape, fp32 or bf16 storage, fp32 accumulate on the vector ALUs.  This is the fp32 parity path
// (dtype=fp32 is a supported reference config, /root/reference/ssi/constants.py:25) and the fallback for shapes the
// MFMA kernel (gemm_mfma.hip) does not take.  64x64 output tile per 256-thread block, 4x4 outputs per thread, BK=16.
#include "common_hip.h"
#include <atomic>

int ssi_gemm_mfma_bf16(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                       int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                       int accumulate, void* stream);  // gemm_mfma.hip
bool ssi_gemm_mfma_supported(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                             int64_t ldb, const void* C, int64_t ldc, const void* R);

int ssi_gemm_mfma_bf16_splitk(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                              int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                              int accumulate, int splits, float* slabs, void* stream);

// process-wide switch (tests, A/B runs): atomic so that the forward thread and autograd's backward thread never race on it
static std::atomic<int> g_impl{SSI_IMPL_AUTO};
extern "C" int ssi_set_impl(int impl) {
    if (impl >= SSI_IMPL_AUTO && impl <= SSI_IMPL_MFMA_WG8) return g_impl.exchange(impl, std::memory_order_relaxed);
    return g_impl.load(std::memory_order_relaxed);
}
int ssi_get_impl() { return g_impl.load(std::memory_order_relaxed); }

void ssi_gemm_mfma_set_dynamic_tiles(int on);
extern "C" int ssi_set_gemm_tile_order(int mode) {
    if (mode != SSI_TILES_STATIC && mode != SSI_TILES_DYNAMIC) { ssi_set_error("ssi_set_gemm_tile_order: mode %d", mode); return SSI_ERR_ARG; }
    ssi_gemm_mfma_set_dynamic_tiles(mode == SSI_TILES_DYNAMIC);
    return SSI_OK;
}

// element (m,k) of op(A) at A[m*sam + k*sak]; element (k,n) of op(B) at B[k*sbk + n*sbn]
template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(int64_t M, int64_t N, int64_t K, const T* __restrict__ A,
                                                           int64_t sam, int64_t sak, const T* __restrict__ B, int64_t sbk,
                                                           int64_t sbn, T* __restrict__ C, int64_t ldc,
                                                           const T* __restrict__ R, float alpha,
                                                           const float* __restrict__ alpha_dev, int accumulate) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    // loader mapping: choose the fast-varying thread index along the contiguous memory dimension
    const bool a_k_contig = (sak == 1);
    const bool b_n_contig = (sbn == 1);
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + it * 256;  // 0..1023 over BM x BK
            int mm, kk;
            if (a_k_contig) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            const int64_t gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < K) ? to_f32<T>(A[gm * sam + gk * sak]) : 0.f;
            int nn, kb;
            if (b_n_contig) { nn = e & 63; kb = e >> 6; } else { kb = e & 15; nn = e >> 4; }
            const int64_t gn = n0 + nn, gkb = k0 + kb;
            Bs[kb][nn] = (gn < N && gkb < K) ? to_f32<T>(B[gkb * sbk + gn * sbn]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    const float al = alpha * (alpha_dev ? *alpha_dev : 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gn = n0 + tx * 4 + j;
            if (gn >= N) continue;
            float v = al * acc[i][j];
            if (accumulate) v += to_f32<T>(C[gm * ldc + gn]);
            if (R) v += to_f32<T>(R[gm * ldc + gn]);
            C[gm * ldc + gn] = from_f32<T>(v);
        }
    }
}

extern "C" int ssi_gemm(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                        int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                        int accumulate, int dtype, void* stream) {
    SSI_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0 && ldc >= N);
    SSI_CHECK_ARG(layout == SSI_GEMM_NT || layout == SSI_GEMM_NN || layout == SSI_GEMM_TN);
    if (M == 0 || N == 0) return SSI_OK;
    int64_t sam, sak, sbk, sbn;
    if (layout == SSI_GEMM_NT)      { SSI_CHECK_ARG(lda >= K && ldb >= K); sam = lda; sak = 1;   sbk = 1;   sbn = ldb; }
    else if (layout == SSI_GEMM_NN) { SSI_CHECK_ARG(lda >= K && ldb >= N); sam = lda; sak = 1;   sbk = ldb; sbn = 1; }
    else                            { SSI_CHECK_ARG(lda >= M && ldb >= N); sam = 1;   sak = lda; sbk = ldb; sbn = 1; }

    const bool mfma_ok = dtype == SSI_BF16 && ssi_gemm_mfma_supported(layout, M, N, K, A, lda, B, ldb, C, ldc, R);
    if ((g_impl == SSI_IMPL_MFMA || g_impl == SSI_IMPL_MFMA_WG8) && !mfma_ok) {
        ssi_set_error("ssi_gemm: MFMA path forced but shape/dtype unsupported (M=%lld N=%lld K=%lld dtype=%d)",
                      (long long)M, (long long)N, (long long)K, dtype);
        return SSI_ERR_UNSUPPORTED;
    }
    if (mfma_ok && g_impl != SSI_IMPL_GENERIC)
        return ssi_gemm_mfma_bf16(layout, M, N, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, stream);

    SSI_CHECK_ARG(ssi_cdiv(M, 64) <= 65535);
    dim3 grid((unsigned)ssi_cdiv(N, 64), (unsigned)ssi_cdiv(M, 64));
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(gemm_generic_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, M, N, K,
                                                 (const T*)A, sam, sak, (const T*)B, sbk, sbn, (T*)C, ldc, (const T*)R,
                                                 alpha, alpha_dev, accumulate));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// `batch` GEMMs of one shape in one launch: problem b works on A + b strideA, B + b strideB, C + b strideC (strides in elements).
// For contractions whose output grid cannot fill 256 CUs but which come in several independent copies: the weight gradients of the
// square projections (dW_o: 64 output tiles, dW_qkv: 96) of several LAYERS, deferred by the model until a group of layers has finished
// its backward, fill the chip as one launch at full K instead of a split-K launch + reduction per layer.  Shapes the persistent MFMA kernel
// does not take run as `batch` ssi_gemm calls.
bool ssi_gemm_mfma_bf16_batched(int layout, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t bsA, const void* B,
                                int64_t ldb, int64_t bsB, void* C, int64_t ldc, int64_t bsC, float alpha, const float* alpha_dev, int accumulate,
                                void* stream, int* rc);
extern "C" int ssi_gemm_batched(int layout, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t strideA,
                                const void* B, int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC, float alpha,
                                const float* alpha_dev, int accumulate, int dtype, void* stream) {
    SSI_CHECK_ARG(batch >= 0 && strideA >= 0 && strideB >= 0 && strideC >= 0);
    SSI_CHECK_ARG(dtype == SSI_BF16 || dtype == SSI_F32);
    if (batch == 0) return SSI_OK;
    SSI_CHECK_ARG(A && B && C);
    const int64_t es = dtype == SSI_BF16 ? 2 : 4;
    auto off = [&](const void* p, int64_t b, int64_t stride) { return (const void*)((const char*)p + b * stride * es); };
    bool mfma_ok = dtype == SSI_BF16 && batch > 1 && g_impl != SSI_IMPL_GENERIC && g_impl != SSI_IMPL_MFMA_WG8;
    for (int b = 0; b < batch && mfma_ok; ++b)
        mfma_ok = ssi_gemm_mfma_supported(layout, M, N, K, off(A, b, strideA), lda, off(B, b, strideB), ldb, off(C, b, strideC), ldc, nullptr);
    if (mfma_ok) {
        SSI_CHECK_ARG(layout == SSI_GEMM_NT || layout == SSI_GEMM_NN || layout == SSI_GEMM_TN);
        int rc = SSI_OK;
        if (ssi_gemm_mfma_bf16_batched(layout, batch, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, alpha, alpha_dev, accumulate,
                                       stream, &rc))
            return rc;
    }
    for (int b = 0; b < batch; ++b)
        if (int rc = ssi_gemm(layout, M, N, K, off(A, b, strideA), lda, off(B, b, strideB), ldb, (void*)off(C, b, strideC), ldc, nullptr, alpha,
                              alpha_dev, accumulate, dtype, stream))
            return rc;
    return SSI_OK;
}

// Split-K form of ssi_gemm for contractions whose output grid cannot fill 256 CUs (weight gradients of the square
// projections: K = tokens is long, M x N is small).  The K range is cut into `splits` slices computed by separate
// workgroups into fp32 slabs ([splits, M, N] in `workspace`), then one pass sums the slabs and applies the epilogue.
// MFMA path only (same shape rules as ssi_gemm); splits == 1 forwards to ssi_gemm.
extern "C" int64_t ssi_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits) {
    return splits <= 1 ? 0 : (int64_t)splits * M * N * (int64_t)sizeof(float);
}

extern "C" int ssi_gemm_splitk(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                               int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                               int accumulate, int dtype, int splits, void* workspace, int64_t workspace_bytes, void* stream) {
    const bool mfma_ok = dtype == SSI_BF16 && ssi_gemm_mfma_supported(layout, M, N, K, A, lda, B, ldb, C, ldc, R);
    if (splits <= 1 || !mfma_ok || g_impl == SSI_IMPL_GENERIC || K / 64 < splits)
        return ssi_gemm(layout, M, N, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, dtype, stream);
    SSI_CHECK_ARG(layout == SSI_GEMM_NT || layout == SSI_GEMM_NN || layout == SSI_GEMM_TN);
    SSI_CHECK_ARG(splits <= 64 && ((uintptr_t)workspace % 16) == 0);
    if (!workspace || workspace_bytes < ssi_gemm_splitk_workspace_bytes(M, N, splits)) { ssi_set_error("ssi_gemm_splitk: workspace too small"); return SSI_ERR_WORKSPACE; }
    return ssi_gemm_mfma_bf16_splitk(layout, M, N, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, splits,
                                     (float*)workspace, stream);
}

// ---- QKV projection + RoPE (K3 + K4): C = A B^T, then the interleaved rotation on the first rot_heads heads of every row --------
bool ssi_gemm_rope_mfma(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                        const float* rope, const int32_t* positions, int64_t seq, int64_t rot_cols, void* stream, int* rc);
extern "C" int ssi_rope_inplace(void* x, int64_t ld, int64_t rows, int64_t seq_len, int n_heads_rot, int head_dim, const float* table,
                                int64_t table_len, const int32_t* positions, int inverse, int dtype, void* stream);

extern "C" int ssi_gemm_rope(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                             int64_t seq_len, int n_heads_rot, int head_dim, const float* rope_table, int64_t table_len,
                             const int32_t* positions, int dtype, void* stream) {
    SSI_CHECK_ARG(A && B && C && rope_table && M >= 0 && N > 0 && K > 0 && seq_len > 0 && n_heads_rot > 0 && head_dim > 0);
    SSI_CHECK_ARG((int64_t)n_heads_rot * head_dim <= N && (positions != nullptr || table_len >= seq_len));
    if (M == 0) return SSI_OK;
    int rc = SSI_OK;
    if (dtype == SSI_BF16 && head_dim == 64 && g_impl != SSI_IMPL_GENERIC &&
        ssi_gemm_rope_mfma(M, N, K, A, lda, B, ldb, C, ldc, rope_table, positions, seq_len, (int64_t)n_heads_rot * head_dim, stream, &rc))
        return rc;
    if ((rc = ssi_gemm(SSI_GEMM_NT, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 1.f, nullptr, 0, dtype, stream))) return rc;
    return ssi_rope_inplace(C, ldc, M, seq_len, n_heads_rot, head_dim, rope_table, table_len, positions, 0, dtype, stream);
}

// ---- fused SwiGLU GEMMs (K7 of SURVEY.md §2.3: FeedForward w2(silu(w1 x) * w3 x) and its backward) -------------------------
bool ssi_gemm_swiglu_supported(int64_t M, int64_t inter, int64_t K, const void* p0, const void* p1, const void* p2, const void* p3,
                               int64_t ld0, int64_t ld1, int64_t ld2, int64_t ld3);
int ssi_gemm_swiglu_fwd_mfma(int64_t M, int64_t inter, int64_t K, const void* X, int64_t ldx, const void* W13, int64_t ldw,
                             void* GU, int64_t ldgu, void* ACT, int64_t ldact, void* stream);
int ssi_gemm_swiglu_bwd_mfma(int layout, int64_t M, int64_t inter, int64_t K, const void* DY, int64_t lddy, const void* W2, int64_t ldw,
                             const void* GU, int64_t ldgu, void* DGU, int64_t lddgu, void* stream);
bool ssi_gemm_swiglu_bwd_nn_supported(int64_t K, int64_t lddy, int64_t ldw);

extern "C" int ssi_gemm_swiglu_fwd(int64_t M, int64_t inter, int64_t K, const void* X, int64_t ldx, const void* W13, int64_t ldw,
                                   void* GU, int64_t ldgu, void* ACT, int64_t ldact, int dtype, void* stream) {
    SSI_CHECK_ARG(X && W13 && GU && ACT && M >= 0 && inter > 0 && K > 0 && ldgu >= 2 * inter && ldact >= inter);
    if (dtype == SSI_BF16 && g_impl != SSI_IMPL_GENERIC && ssi_gemm_swiglu_supported(M, inter, K, X, W13, GU, ACT, ldx, ldw, ldgu, ldact))
        return ssi_gemm_swiglu_fwd_mfma(M, inter, K, X, ldx, W13, ldw, GU, ldgu, ACT, ldact, stream);
    if (int rc = ssi_gemm(SSI_GEMM_NT, M, 2 * inter, K, X, ldx, W13, ldw, GU, ldgu, nullptr, 1.f, nullptr, 0, dtype, stream)) return rc;
    SSI_CHECK_ARG(ldgu == 2 * inter && ldact == inter);
    return ssi_swiglu_fwd(GU, ACT, M, inter, dtype, stream);
}

// d gu = swiglu_backward(DY * W2, GU).  W2 is given as [I, K] ("transposed copy", layout NT) or as [K, I] (layout NN).
// Unfused fallback needs `dact_ws` ([M, I] scratch, may be NULL only when the fused path is taken).
extern "C" int ssi_gemm_swiglu_bwd(int layout, int64_t M, int64_t inter, int64_t K, const void* DY, int64_t lddy, const void* W2,
                                   int64_t ldw, const void* GU, int64_t ldgu, void* DGU, int64_t lddgu, void* dact_ws, int dtype,
                                   void* stream) {
    SSI_CHECK_ARG(DY && W2 && GU && DGU && M >= 0 && inter > 0 && K > 0 && ldgu >= 2 * inter && lddgu >= 2 * inter);
    SSI_CHECK_ARG(layout == SSI_GEMM_NT || layout == SSI_GEMM_NN);
    if (dtype == SSI_BF16 && g_impl != SSI_IMPL_GENERIC && ssi_gemm_swiglu_supported(M, inter, K, DY, W2, GU, DGU, lddy, ldw, ldgu, lddgu) &&
        (layout == SSI_GEMM_NT || ssi_gemm_swiglu_bwd_nn_supported(K, lddy, ldw)))
        return ssi_gemm_swiglu_bwd_mfma(layout, M, inter, K, DY, lddy, W2, ldw, GU, ldgu, DGU, lddgu, stream);
    SSI_CHECK_ARG(dact_ws != nullptr && ldgu == 2 * inter && lddgu == 2 * inter);
    if (int rc = ssi_gemm(layout, M, inter, K, DY, lddy, W2, ldw, dact_ws, inter, nullptr, 1.f, nullptr, 0, dtype, stream)) return rc;
    return ssi_swiglu_bwd(dact_ws, GU, DGU, M, inter, dtype, stream);
}
