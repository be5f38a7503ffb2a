// Generic causal GQA attention (fp32 math on the vector ALUs, any storage dtype, head_dim in {16,32,64,128}).
// This is the fp32 parity path and the fallback for shapes attention_mfma.hip does not take.  One wave per query row
// (forward, dQ) or per key row (dK, dV); lanes stride over the other sequence axis with per-lane online softmax state
// that is merged across the wave at the end.  Semantics: F.scaled_dot_product_attention(q, k, v, is_causal=True,
// dropout_p=0) with k/v heads repeated n_heads/n_kv times (torchtune MultiHeadAttention).
// Packed rows: doc_start[b*S + s] / doc_end[b*S + s] = first position / one past the last position of
// the document that holds position s; a query then sees the keys doc_start <= j <= i only (torchtune's block-causal mask of
// `padded_collate_packed`), a key is seen by the queries j <= i < doc_end.
#include "common_hip.h"

int ssi_get_impl();
bool ssi_attn_mfma_supported(int64_t ld, int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype);
int ssi_attn_fwd_mfma(const void* qkv, int64_t ld, void* out, float* lse, const int32_t* doc_start, int64_t batch, int64_t seq,
                      int n_heads, int n_kv, void* stream);
int ssi_attn_bwd_mfma(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                      float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope, int64_t table_len,
                      const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv, void* workspace, int64_t workspace_bytes,
                      const int32_t* plan_dev, const int32_t* host_plan_header, void* stream);
void ssi_attn_note_dispatch(int v);
int64_t ssi_attn_mfma_bwd_workspace_bytes(int64_t batch, int64_t seq, int n_heads, int n_kv);
extern "C" int ssi_rope_inplace(void* x, int64_t ld, int64_t rows, int64_t seq_len, int n_heads_rot, int head_dim, const float* table,
                                int64_t table_len, const int32_t* positions, int inverse, int dtype, void* stream);

template <typename T, int HD>
__device__ __forceinline__ void load_row(const T* p, float (&r)[HD]) {
    constexpr int N = Vec16<T>::N;
#pragma unroll
    for (int v = 0; v < HD / N; ++v) {
        Vec16<T> a = load16(p + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) r[v * N + i] = a.get(i);
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_generic(const T* __restrict__ qkv, int64_t ld, T* __restrict__ out,
                                                        float* __restrict__ lse, const int32_t* __restrict__ doc_start,
                                                        int64_t batch, int64_t seq, int n_heads, int n_kv) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, h, i)
    if (w >= batch * n_heads * seq) return;
    const int64_t i = w % seq;
    const int h = (int)((w / seq) % n_heads);
    const int64_t b = w / (seq * n_heads);
    const int kvh = h / (n_heads / n_kv);
    const float scale = rsqrtf((float)HD);
    const T* qrow = qkv + (b * seq + i) * ld + (int64_t)h * HD;
    const int64_t koff = (int64_t)n_heads * HD + (int64_t)kvh * HD;
    const int64_t voff = (int64_t)(n_heads + n_kv) * HD + (int64_t)kvh * HD;
    float q[HD], o[HD];
    load_row<T, HD>(qrow, q);
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int64_t j0 = doc_start ? doc_start[b * seq + i] : 0;  // packed rows: keys of the query's own document only
    for (int64_t j = j0 + lane; j <= i; j += 64) {
        const T* krow = qkv + (b * seq + j) * ld + koff;
        const T* vrow = qkv + (b * seq + j) * ld + voff;
        float kr[HD];
        load_row<T, HD>(krow, kr);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(q[d], kr[d], s);
        s *= scale;
        const float mn = fmaxf(m, s);
        const float corr = expf(m - mn), p = expf(s - mn);
        l = l * corr + p;
        load_row<T, HD>(vrow, kr);
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = o[d] * corr + p * kr[d];
        m = mn;
    }
    const float M = wave_max(m);
    const float f = (m == -INFINITY) ? 0.f : expf(m - M);
    const float L = wave_sum(l * f);
    const float inv = 1.f / L;
    T* orow = out + (b * seq + i) * ((int64_t)n_heads * HD) + (int64_t)h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float t = wave_sum(o[d] * f) * inv;
        if (lane == (d & 63)) orow[d] = from_f32<T>(t);
    }
    if (lane == 0) lse[(b * n_heads + h) * seq + i] = M + logf(L);
}

// delta[b,h,i] = sum_d dout[i,h,d] * out[i,h,d]
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_delta_generic(const T* __restrict__ out, const T* __restrict__ dout,
                                                          float* __restrict__ delta, int64_t batch, int64_t seq,
                                                          int n_heads) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (b, h, i)
    if (g >= batch * n_heads * seq) return;
    const int64_t i = g % seq;
    const int h = (int)((g / seq) % n_heads);
    const int64_t b = g / (seq * n_heads);
    const int64_t off = (b * seq + i) * ((int64_t)n_heads * HD) + (int64_t)h * HD;
    float a[HD], c[HD];
    load_row<T, HD>(out + off, a);
    load_row<T, HD>(dout + off, c);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(a[d], c[d], s);
    delta[g] = s;
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_generic(const T* __restrict__ qkv, int64_t ld, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dqkv, const int32_t* __restrict__ doc_start,
                                                           int64_t batch, int64_t seq, int n_heads, int n_kv) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= batch * n_heads * seq) return;
    const int64_t i = w % seq;
    const int h = (int)((w / seq) % n_heads);
    const int64_t b = w / (seq * n_heads);
    const int kvh = h / (n_heads / n_kv);
    const float scale = rsqrtf((float)HD);
    const int64_t koff = (int64_t)n_heads * HD + (int64_t)kvh * HD;
    const int64_t voff = (int64_t)(n_heads + n_kv) * HD + (int64_t)kvh * HD;
    float q[HD], dO[HD], dq[HD];
    load_row<T, HD>(qkv + (b * seq + i) * ld + (int64_t)h * HD, q);
    load_row<T, HD>(dout + (b * seq + i) * ((int64_t)n_heads * HD) + (int64_t)h * HD, dO);
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    const float L = lse[(b * n_heads + h) * seq + i], dl = delta[(b * n_heads + h) * seq + i];
    const int64_t j0 = doc_start ? doc_start[b * seq + i] : 0;
    for (int64_t j = j0 + lane; j <= i; j += 64) {
        float r[HD];
        load_row<T, HD>(qkv + (b * seq + j) * ld + voff, r);
        float dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) dp = fmaf(dO[d], r[d], dp);
        load_row<T, HD>(qkv + (b * seq + j) * ld + koff, r);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(q[d], r[d], s);
        const float p = expf(s * scale - L);
        const float ds = p * (dp - dl) * scale;
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, r[d], dq[d]);
    }
    T* drow = dqkv + (b * seq + i) * ld + (int64_t)h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float t = wave_sum(dq[d]);
        if (lane == (d & 63)) drow[d] = from_f32<T>(t);
    }
}

// one wave per (b, kv head, key j); lanes stride over (q head of the group, query i >= j).  DK selects dK or dV.
template <typename T, int HD, bool DK>
__global__ __launch_bounds__(256) void attn_bwd_dkv_generic(const T* __restrict__ qkv, int64_t ld, const T* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            T* __restrict__ dqkv, const int32_t* __restrict__ doc_end,
                                                            int64_t batch, int64_t seq, int n_heads, int n_kv) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= batch * n_kv * seq) return;
    const int64_t j = w % seq;
    const int kvh = (int)((w / seq) % n_kv);
    const int64_t b = w / (seq * n_kv);
    const int rep = n_heads / n_kv;
    const float scale = rsqrtf((float)HD);
    const int64_t koff = (int64_t)n_heads * HD + (int64_t)kvh * HD;
    const int64_t voff = (int64_t)(n_heads + n_kv) * HD + (int64_t)kvh * HD;
    float kr[HD], acc[HD];
    load_row<T, HD>(qkv + (b * seq + j) * ld + koff, kr);
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    const int64_t nq = (doc_end ? doc_end[b * seq + j] : seq) - j;  // queries i = j .. end of the key's document - 1
    for (int64_t e = lane; e < nq * rep; e += 64) {
        const int64_t i = j + e / rep;
        const int h = kvh * rep + (int)(e % rep);
        float r[HD];
        load_row<T, HD>(qkv + (b * seq + i) * ld + (int64_t)h * HD, r);  // q_i
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(r[d], kr[d], s);
        const float p = expf(s * scale - lse[(b * n_heads + h) * seq + i]);
        if (DK) {
            float vr[HD];
            load_row<T, HD>(qkv + (b * seq + j) * ld + voff, vr);
            float dO[HD];
            load_row<T, HD>(dout + (b * seq + i) * ((int64_t)n_heads * HD) + (int64_t)h * HD, dO);
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dp = fmaf(dO[d], vr[d], dp);
            const float ds = p * (dp - delta[(b * n_heads + h) * seq + i]) * scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(ds, r[d], acc[d]);
        } else {
            float dO[HD];
            load_row<T, HD>(dout + (b * seq + i) * ((int64_t)n_heads * HD) + (int64_t)h * HD, dO);
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, dO[d], acc[d]);
        }
    }
    T* drow = dqkv + (b * seq + j) * ld + (DK ? koff : voff);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float t = wave_sum(acc[d]);
        if (lane == (d & 63)) drow[d] = from_f32<T>(t);
    }
}

#define ATTN_HD_SWITCH(hd, ...)                                                   \
    switch (hd) {                                                                 \
        case 16: { constexpr int HD = 16; __VA_ARGS__; } break;                   \
        case 32: { constexpr int HD = 32; __VA_ARGS__; } break;                   \
        case 64: { constexpr int HD = 64; __VA_ARGS__; } break;                   \
        case 128: { constexpr int HD = 128; __VA_ARGS__; } break;                 \
        default: ssi_set_error("attention: unsupported head_dim %d", (int)(hd)); return SSI_ERR_UNSUPPORTED; \
    }

static int attn_check(const void* qkv, int64_t ld, int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim) {
    SSI_CHECK_ARG(qkv && batch >= 0 && seq >= 0 && n_heads > 0 && n_kv > 0 && n_heads % n_kv == 0 && head_dim > 0);
    SSI_CHECK_ARG(ld >= (int64_t)(n_heads + 2 * n_kv) * head_dim && ld % 8 == 0);
    return SSI_OK;
}

extern "C" int ssi_attn_varlen_fwd(const void* qkv, int64_t ld, void* out, float* lse, const int32_t* doc_start, const int32_t* doc_end,
                                   int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype, void* stream) {
    if (int rc = attn_check(qkv, ld, batch, seq, n_heads, n_kv, head_dim)) return rc;
    SSI_CHECK_ARG(out && lse && ((doc_start == nullptr) == (doc_end == nullptr)));
    if (batch * seq == 0) return SSI_OK;
    const bool fast = ssi_attn_mfma_supported(ld, batch, seq, n_heads, n_kv, head_dim, dtype);
    if ((ssi_get_impl() == SSI_IMPL_MFMA || ssi_get_impl() == SSI_IMPL_MFMA_WG8) && !fast) { ssi_set_error("ssi_attn_fwd: MFMA path forced but unsupported shape"); return SSI_ERR_UNSUPPORTED; }
    if (fast && ssi_get_impl() != SSI_IMPL_GENERIC) return ssi_attn_fwd_mfma(qkv, ld, out, lse, doc_start, batch, seq, n_heads, n_kv, stream);
    const int64_t nw = batch * n_heads * seq;
    SSI_DISPATCH_DTYPE(dtype, ATTN_HD_SWITCH(head_dim, hipLaunchKernelGGL((attn_fwd_generic<T, HD>), dim3((unsigned)ssi_cdiv(nw, 4)),
                                                                         dim3(256), 0, (hipStream_t)stream, (const T*)qkv, ld,
                                                                         (T*)out, lse, doc_start, batch, seq, n_heads, n_kv)));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// rope_table != NULL: dqkv's q and k heads are returned in pre-RoPE space, i.e. followed by ssi_rope_inplace(inverse) — inside the
// MFMA kernels' epilogues, as a second launch on the generic path.
static int attn_varlen_bwd_impl(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                                float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                                int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                                int head_dim, int dtype, void* stream, void* workspace = nullptr, int64_t workspace_bytes = 0,
                                const int32_t* plan = nullptr, const int32_t* host_plan_header = nullptr) {
    if (int rc = attn_check(qkv, ld, batch, seq, n_heads, n_kv, head_dim)) return rc;
    SSI_CHECK_ARG(out && dout && lse && dqkv && delta && ((doc_start == nullptr) == (doc_end == nullptr)));
    if (batch * seq == 0) return SSI_OK;
    const bool fast = ssi_attn_mfma_supported(ld, batch, seq, n_heads, n_kv, head_dim, dtype);
    if ((ssi_get_impl() == SSI_IMPL_MFMA || ssi_get_impl() == SSI_IMPL_MFMA_WG8) && !fast) { ssi_set_error("ssi_attn_bwd: MFMA path forced but unsupported shape"); return SSI_ERR_UNSUPPORTED; }
    if (fast && ssi_get_impl() != SSI_IMPL_GENERIC)
        return ssi_attn_bwd_mfma(qkv, ld, out, dout, lse, dqkv, delta, doc_start, doc_end, rope_table, table_len, positions, batch, seq, n_heads,
                                 n_kv, workspace, workspace_bytes, plan, host_plan_header, stream);
    ssi_attn_note_dispatch(0);
    auto st = (hipStream_t)stream;
    const int64_t nq = batch * n_heads * seq, nk = batch * n_kv * seq;
    SSI_DISPATCH_DTYPE(dtype, ATTN_HD_SWITCH(head_dim, {
        hipLaunchKernelGGL((attn_delta_generic<T, HD>), dim3((unsigned)ssi_cdiv(nq, 256)), dim3(256), 0, st, (const T*)out,
                           (const T*)dout, delta, batch, seq, n_heads);
        hipLaunchKernelGGL((attn_bwd_dq_generic<T, HD>), dim3((unsigned)ssi_cdiv(nq, 4)), dim3(256), 0, st, (const T*)qkv, ld,
                           (const T*)dout, lse, delta, (T*)dqkv, doc_start, batch, seq, n_heads, n_kv);
        hipLaunchKernelGGL((attn_bwd_dkv_generic<T, HD, true>), dim3((unsigned)ssi_cdiv(nk, 4)), dim3(256), 0, st, (const T*)qkv,
                           ld, (const T*)dout, lse, delta, (T*)dqkv, doc_end, batch, seq, n_heads, n_kv);
        hipLaunchKernelGGL((attn_bwd_dkv_generic<T, HD, false>), dim3((unsigned)ssi_cdiv(nk, 4)), dim3(256), 0, st, (const T*)qkv,
                           ld, (const T*)dout, lse, delta, (T*)dqkv, doc_end, batch, seq, n_heads, n_kv);
    }));
    SSI_LAUNCH_CHECK();
    if (rope_table)
        return ssi_rope_inplace(dqkv, ld, batch * seq, seq, n_heads + n_kv, head_dim, rope_table, table_len, positions, 1, dtype, stream);
    return SSI_OK;
}

extern "C" int ssi_attn_varlen_bwd(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                                   float* delta, const int32_t* doc_start, const int32_t* doc_end, int64_t batch, int64_t seq,
                                   int n_heads, int n_kv, int head_dim, int dtype, void* stream) {
    return attn_varlen_bwd_impl(qkv, ld, out, dout, lse, dqkv, delta, doc_start, doc_end, nullptr, 0, nullptr, batch, seq, n_heads, n_kv,
                                head_dim, dtype, stream);
}

extern "C" int ssi_attn_varlen_bwd_rope(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                                        float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                                        int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                                        int head_dim, int dtype, void* stream) {
    SSI_CHECK_ARG(rope_table != nullptr && (positions != nullptr || table_len >= seq));
    return attn_varlen_bwd_impl(qkv, ld, out, dout, lse, dqkv, delta, doc_start, doc_end, rope_table, table_len, positions, batch, seq,
                                n_heads, n_kv, head_dim, dtype, stream);
}


/* ABI v6: the backward with a caller-owned workspace.  ssi_attn_bwd_workspace_bytes says how much this shape can use (0: none); with at least
 * that much, launches whose workgroups cannot fill the chip (small batches) run dK / dV as one workgroup per query head plus a reduction over the
 * heads.  rope_table may be NULL (then dqkv stays in post-RoPE space, as ssi_attn_varlen_bwd).  Without workspace: ssi_attn_varlen_bwd(_rope). */
extern "C" int64_t ssi_attn_bwd_workspace_bytes(int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype) {
    if (batch <= 0 || seq <= 0 || n_heads <= 0 || n_kv <= 0 || n_heads % n_kv) return 0;
    if (!ssi_attn_mfma_supported(8, batch, seq, n_heads, n_kv, head_dim, dtype) || ssi_get_impl() == SSI_IMPL_GENERIC) return 0;
    return ssi_attn_mfma_bwd_workspace_bytes(batch, seq, n_heads, n_kv);
}

extern "C" int ssi_attn_varlen_bwd_ws(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                                      float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                                      int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                                      int head_dim, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    SSI_CHECK_ARG(!rope_table || positions != nullptr || table_len >= seq);
    SSI_CHECK_ARG(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0));
    return attn_varlen_bwd_impl(qkv, ld, out, dout, lse, dqkv, delta, doc_start, doc_end, rope_table, table_len, positions, batch, seq,
                                n_heads, n_kv, head_dim, dtype, stream, workspace, workspace_bytes);
}

extern "C" int ssi_attn_varlen_bwd_plan(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                                        float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                                        int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                                        int head_dim, int dtype, void* workspace, int64_t workspace_bytes, const int32_t* plan,
                                        const int32_t* host_plan_header, void* stream) {
    SSI_CHECK_ARG(!rope_table || positions != nullptr || table_len >= seq);
    SSI_CHECK_ARG(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0));
    SSI_CHECK_ARG((plan == nullptr) == (host_plan_header == nullptr));
    SSI_CHECK_ARG(!plan || ((uintptr_t)plan & 15) == 0);
    return attn_varlen_bwd_impl(qkv, ld, out, dout, lse, dqkv, delta, doc_start, doc_end, rope_table, table_len, positions, batch, seq,
                                n_heads, n_kv, head_dim, dtype, stream, workspace, workspace_bytes, plan, host_plan_header);
}

extern "C" int ssi_attn_fwd(const void* qkv, int64_t ld, void* out, float* lse, int64_t batch, int64_t seq, int n_heads,
                            int n_kv, int head_dim, int dtype, void* stream) {
    return ssi_attn_varlen_fwd(qkv, ld, out, lse, nullptr, nullptr, batch, seq, n_heads, n_kv, head_dim, dtype, stream);
}

extern "C" int ssi_attn_bwd(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* delta, int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype,
                            void* stream) {
    return ssi_attn_varlen_bwd(qkv, ld, out, dout, lse, dqkv, delta, nullptr, nullptr, batch, seq, n_heads, n_kv, head_dim, dtype, stream);
}
