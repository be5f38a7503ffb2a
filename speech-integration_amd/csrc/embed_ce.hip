// K1 embedding gather / deterministic scatter-add, and K9 cross-entropy over the DSU-extended vocabulary.
#include "common.cuh"

// =====================================================================================================================
// K1 forward: out[t,:] = table[tokens[t],:]   (coalesced 16-B row copies; one block of 256 lanes per 4 KiB of row)
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tokens, const T* __restrict__ table,
                                                        T* __restrict__ out, int64_t n_tok, int dim, int64_t vocab) {
    constexpr int N = Vec16<T>::N;
    const int nvec = dim / N;
    for (int64_t t = blockIdx.x; t < n_tok; t += gridDim.x) {
        const int64_t tok = tokens[t];
        const bool ok = tok >= 0 && tok < vocab;
        for (int v = threadIdx.x; v < nvec; v += 256) {
            Vec16<T> a;
            if (ok) a = load16(table + tok * dim + v * N);
            else
#pragma unroll
                for (int i = 0; i < N; ++i) a.set(i, 0.f);
            store16(out + t * dim + v * N, a);
        }
    }
}

extern "C" int ssi_embed_fwd(const int64_t* tokens, const void* table, void* out, int64_t n_tok, int64_t dim,
                             int64_t vocab, int dtype, void* stream) {
    if (n_tok == 0) return SSI_OK;
    SSI_CHECK_ARG(tokens && table && out && n_tok > 0 && dim > 0 && dim % 8 == 0 && vocab > 0);
    const unsigned grid = (unsigned)(n_tok < 65536 ? n_tok : 65536);
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, tokens,
                                                 (const T*)table, (T*)out, n_tok, (int)dim, vocab));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K1 backward: dtable[v,:] += sum over positions t with tokens[t]==v of dout[t,:]
//   pass 1: first[v] = min t, count[v] = #occurrences            (integer atomics: order-independent results)
//   pass 2: one wave per position t; only the wave with t == first[token] works: it scans the token array forward in
//           64-wide chunks (ballot), adds the matching rows in increasing t (fixed order => bitwise reproducible, no float
//           atomics), stops after count[v] matches, and does ONE read-modify-write of the table row.
// =====================================================================================================================
__global__ __launch_bounds__(256) void embed_index_kernel(const int64_t* __restrict__ tokens, int64_t n_tok, int64_t vocab,
                                                          int* __restrict__ first, int* __restrict__ count) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const int64_t tok = tokens[t];
    if (tok < 0 || tok >= vocab) return;
    atomicMin(&first[tok], (int)t);
    atomicAdd(&count[tok], 1);
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tokens, const T* __restrict__ dout,
                                                        T* __restrict__ dtable, int64_t n_tok, int dim, int64_t vocab,
                                                        const int* __restrict__ first, const int* __restrict__ count) {
    constexpr int N = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int64_t t0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t0 >= n_tok) return;
    const int64_t tok = tokens[t0];
    if (tok < 0 || tok >= vocab) return;
    if (first[tok] != (int)t0) return;  // wave-uniform
    const int need = count[tok];
    const int nvec = dim / N;
    float acc[MAXV][N];
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) acc[k][i] = 0.f;
    // A frequent token (hundreds of occurrences spread over the sequence) makes this wave the one the launch waits for, and a
    // scan that fetched one 64-position chunk of ids, then the matching row, then the next chunk ... was a chain of dependent
    // memory round trips (0.5 ms at 16 384 tokens).  So: the ids of SCAN chunks are requested together, the positions that match
    // are listed in LDS in increasing order, and their rows are requested ROWS at a time and added in list order.
    constexpr int SCAN = 8, ROWS = 8;
    __shared__ int hits_lds[4][64 * SCAN];
    int* hits = hits_lds[threadIdx.x >> 6];
    int found = 0;
    // every load below is unconditional on a clamped address and its result is masked afterwards: behind a branch hipcc waits for
    // each guarded load before it issues the next, which is the chain all over again
    for (int64_t base0 = (t0 / 64) * 64; base0 < n_tok && found < need; base0 += 64 * SCAN) {
        int64_t tk[SCAN];
#pragma unroll
        for (int u = 0; u < SCAN; ++u) {
            const int64_t idx = base0 + 64 * u + lane;
            tk[u] = tokens[idx < n_tok ? idx : n_tok - 1];
        }
        int n = 0;
#pragma unroll
        for (int u = 0; u < SCAN; ++u) {
            const int64_t idx = base0 + 64 * u + lane;
            const bool hit = idx < n_tok && idx >= t0 && tk[u] == tok;
            const unsigned long long mask = __ballot(hit);
            if (hit) hits[n + __popcll(mask & ((1ull << lane) - 1ull))] = (int)(idx - base0);
            n += __popcll(mask);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the list is written before it is read back (same wave, no barrier needed)
        if (n == 1) {  // the common case (a token seen once in these 512 positions): one row, no padding requests
            const T* src = dout + (base0 + hits[0]) * dim;
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
                const int v = lane + k * 64;
                const Vec16<T> r = load16(src + (v < nvec ? v : nvec - 1) * N);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[k][i] += v < nvec ? r.get(i) : 0.f;
            }
        } else
        for (int h0 = 0; h0 < n; h0 += ROWS) {
            int pos[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) pos[u] = hits[h0 + u < n ? h0 + u : n - 1];
            Vec16<T> a[ROWS][MAXV];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const T* src = dout + (base0 + pos[u]) * dim;
#pragma unroll
                for (int k = 0; k < MAXV; ++k) {
                    const int v = lane + k * 64;
                    a[u][k] = load16(src + (v < nvec ? v : nvec - 1) * N);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // all ROWS x MAXV requests leave before the first is waited for (the scheduler sinks them otherwise)
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const bool row_ok = h0 + u < n;  // wave-uniform
#pragma unroll
                for (int k = 0; k < MAXV; ++k) {
                    const bool ok = row_ok && lane + k * 64 < nvec;
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[k][i] += ok ? a[u][k].get(i) : 0.f;
                }
            }
        }
        found += n;
    }
    T* dst = dtable + tok * dim;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int v = lane + k * 64;
        if (v < nvec) {
            Vec16<T> a = load16(dst + v * N);
#pragma unroll
            for (int i = 0; i < N; ++i) a.set(i, a.get(i) + acc[k][i]);
            store16(dst + v * N, a);
        }
    }
}

static inline int64_t embed_ws_vocab_slots(int64_t vocab) { return ssi_align_up(vocab, 64); }
// the workspace holds first[] and count[] over the vocabulary; sized by vocab, reported through n_tok-independent API
extern "C" int64_t ssi_embed_bwd_workspace_bytes(int64_t vocab) { return 2 * embed_ws_vocab_slots(vocab) * (int64_t)sizeof(int); }

template <typename T>
static int launch_embed_bwd(const int64_t* tokens, const T* dout, T* dtable, int64_t n_tok, int dim, int64_t vocab,
                            const int* first, const int* count, hipStream_t st) {
    const int64_t vec_per_lane = ssi_cdiv(dim / Vec16<T>::N, 64);
    const dim3 grid((unsigned)ssi_cdiv(n_tok, 4));
    if (vec_per_lane <= 1) hipLaunchKernelGGL((embed_bwd_kernel<T, 1>), grid, dim3(256), 0, st, tokens, dout, dtable, n_tok, dim, vocab, first, count);
    else if (vec_per_lane <= 2) hipLaunchKernelGGL((embed_bwd_kernel<T, 2>), grid, dim3(256), 0, st, tokens, dout, dtable, n_tok, dim, vocab, first, count);
    else if (vec_per_lane <= 4) hipLaunchKernelGGL((embed_bwd_kernel<T, 4>), grid, dim3(256), 0, st, tokens, dout, dtable, n_tok, dim, vocab, first, count);
    else if (vec_per_lane <= 8) hipLaunchKernelGGL((embed_bwd_kernel<T, 8>), grid, dim3(256), 0, st, tokens, dout, dtable, n_tok, dim, vocab, first, count);
    else { ssi_set_error("embed_bwd: dim %d too large", dim); return SSI_ERR_UNSUPPORTED; }
    return SSI_OK;
}

extern "C" int ssi_embed_bwd(const int64_t* tokens, const void* dout, void* dtable, int64_t n_tok, int64_t dim,
                             int64_t vocab, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_tok == 0) return SSI_OK;
    SSI_CHECK_ARG(tokens && dout && dtable && n_tok > 0 && n_tok < (1LL << 31) && dim > 0 && dim % 8 == 0 && vocab > 0);
    if (!workspace || workspace_bytes < ssi_embed_bwd_workspace_bytes(vocab)) { ssi_set_error("embed_bwd: workspace too small"); return SSI_ERR_WORKSPACE; }
    auto st = (hipStream_t)stream;
    int* first = (int*)workspace;
    int* count = first + embed_ws_vocab_slots(vocab);
    hipError_t e = hipMemsetAsync(first, 0x7f, embed_ws_vocab_slots(vocab) * sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, embed_ws_vocab_slots(vocab) * sizeof(int), st);
    if (e != hipSuccess) { ssi_set_error("embed_bwd: memset failed: %s", hipGetErrorString(e)); return SSI_ERR_HIP + (int)e; }
    hipLaunchKernelGGL(embed_index_kernel, dim3((unsigned)ssi_cdiv(n_tok, 256)), dim3(256), 0, st, tokens, n_tok, vocab, first, count);
    SSI_LAUNCH_CHECK();
    int rc = SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, rc = launch_embed_bwd<T>(tokens, (const T*)dout, (T*)dtable, n_tok, (int)dim, vocab, first, count, st));
    if (rc) return rc;
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K9 cross-entropy: one 512-thread block per row of [rows, ld] logits; fp32 online log-sum-exp; optional in-place
// gradient  softmax - onehot.  Pass 2 re-reads the row (266 KB at V=133 258) from L2/Infinity Cache.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(512) void ce_fwd_kernel(T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                     int64_t vocab, int64_t ignore_index, float* __restrict__ row_loss,
                                                     float* __restrict__ row_lse, int write_grad) {
    constexpr int N = Vec16<T>::N;
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    T* lr = logits + row * ld;
    const int64_t label = labels[row];
    const bool valid = label != ignore_index && label >= 0 && label < vocab;
    const int64_t nvec = ld / N;
    if (!valid) {  // block-uniform
        if (threadIdx.x == 0) { row_loss[row] = 0.f; if (row_lse) row_lse[row] = 0.f; }
        if (write_grad) {
            Vec16<T> z;
#pragma unroll
            for (int i = 0; i < N; ++i) z.set(i, 0.f);
            for (int64_t v = threadIdx.x; v < nvec; v += 512) store16(lr + v * N, z);
        }
        return;
    }
    float m = -INFINITY, s = 0.f;
    for (int64_t v = threadIdx.x; v < nvec; v += 512) {
        Vec16<T> a = load16(lr + v * N);
        float lm = -INFINITY;
#pragma unroll
        for (int i = 0; i < N; ++i) if (v * N + i < vocab) lm = fmaxf(lm, a.get(i));
        if (lm > m) { s *= expf(m - lm); m = lm; }
#pragma unroll
        for (int i = 0; i < N; ++i) if (v * N + i < vocab) s += expf(a.get(i) - m);
    }
    const float gm = block_max(m, red);
    s = (m == -INFINITY) ? 0.f : s * expf(m - gm);
    const float gs = block_sum(s, red);
    const float lse = gm + logf(gs);
    if (threadIdx.x == 0) {
        row_loss[row] = lse - to_f32<T>(lr[label]);
        if (row_lse) row_lse[row] = lse;
    }
    if (!write_grad) return;
    __syncthreads();  // lr[label] read above must precede the overwrite below
    for (int64_t v = threadIdx.x; v < nvec; v += 512) {
        Vec16<T> a = load16(lr + v * N), o;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int64_t c = v * N + i;
            float g = 0.f;
            if (c < vocab) g = expf(a.get(i) - lse) - (c == label ? 1.f : 0.f);
            o.set(i, g);
        }
        store16(lr + v * N, o);
    }
}

extern "C" int ssi_ce_fwd(void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab,
                          int64_t ignore_index, float* row_loss, float* row_lse, int write_grad, int dtype, void* stream) {
    SSI_CHECK_ARG(logits && labels && row_loss && rows >= 0 && vocab > 0 && ld >= vocab && ld % 8 == 0);
    if (rows == 0) return SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(ce_fwd_kernel<T>, dim3((unsigned)rows), dim3(512), 0, (hipStream_t)stream,
                                                 (T*)logits, ld, labels, vocab, ignore_index, row_loss, row_lse, write_grad));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                         int64_t rows, int64_t ignore_index, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 1024) {
        s += row_loss[r];
        c += (labels[r] != ignore_index) ? 1.f : 0.f;
    }
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) { out[0] = s / c; out[1] = s; out[2] = c; }
}

extern "C" int ssi_ce_reduce(const float* row_loss, const int64_t* labels, int64_t rows, int64_t ignore_index, float* out,
                             void* stream) {
    SSI_CHECK_ARG(row_loss && labels && out && rows >= 0 && rows < (1LL << 24));
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, labels, rows, ignore_index, out);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}
