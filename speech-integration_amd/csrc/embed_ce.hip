// K1 embedding gather / deterministic scatter-add, and K9 cross-entropy over the DSU-extended vocabulary.
#include "common_hip.h"

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
                                                     float* __restrict__ row_lse, int write_grad, const float* __restrict__ row_weight) {
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
    const float w = row_weight ? row_weight[row] : 1.f;  // weighted rows (ssi_ce_fwd_weighted): loss and gradient of the row times w
    if (threadIdx.x == 0) {
        row_loss[row] = w * (lse - to_f32<T>(lr[label]));
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
            if (c < vocab) g = w * (expf(a.get(i) - lse) - (c == label ? 1.f : 0.f));
            o.set(i, g);
        }
        store16(lr + v * N, o);
    }
}

// bf16 rows held in registers (the training step's form): one 1024-thread workgroup per CU walks rows; a row of NCH x 8192 logits
// (NCH = 17: 139 264 >= 133 376) lives in NCH 16-byte registers per thread, so the log-sum-exp and the gradient both come from ONE
// read of the row: 2 passes over the logits (read + write back = 8.7 GB at T = 16 384, V = 133 258) instead of the 3 of
// ce_fwd_kernel above (13.1 GB).  A register's next-row load is issued as soon as its gradient has been stored, so the next row
// streams in under the stores and the exps of the current one.  NCH = ceil(ld / 8192) exactly and ld - vocab < 8192: only the last
// two chunks can hold columns that are not vocabulary.  Straight-line body (the only branches are workgroup-uniform and read-only
// on the row registers): with per-chunk branches hipcc spills the row.
template <int NCH, bool write_grad>
__global__ __launch_bounds__(1024, 4) void ce_row_bf16_kernel(bf16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                              int64_t rows, int64_t vocab, int64_t ignore_index,
                                                              float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                              const float* __restrict__ row_weight) {
    __shared__ float red[16];
    constexpr float LOG2E = 1.44269504088896340736f;
    constexpr int CHUNK = 8192;                      // columns per chunk: 1024 threads x 8 bf16
    const int tid = threadIdx.x;
    const int voff = tid * 16;                       // the one per-lane byte offset; the chunk rides in the scalar offset
    const int col0 = tid * 8;                        // this lane's first column inside a chunk
    const int row_bytes = (int)(ld * 2);             // buffer bound: loads beyond the row return 0, stores beyond it are dropped
    const int vocab_i = (int)vocab;
    u32x4 x[NCH];
    auto is_valid = [&](int64_t label) { return label != ignore_index && label >= 0 && label < vocab; };
    auto rsrc_of = [&](int64_t row) { return __builtin_amdgcn_make_buffer_rsrc(logits + row * ld, 0, row_bytes, 0x00020000u); };
    auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
    auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
    // the three passes each re-derive the fp32 values from the packed registers: without this opaque touch the compiler keeps all
    // 8 x NCH converted floats alive across the passes and spills
    auto opaque = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) asm volatile("" : "+v"(x[c][d]));
    };
    int64_t row = blockIdx.x;
    if (row >= rows) return;
    {
        const __amdgpu_buffer_rsrc_t r0 = rsrc_of(row);
#pragma unroll
        for (int c = 0; c < NCH; ++c) x[c] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff, c * CHUNK * 2, 0);
    }
    for (; row < rows; row += gridDim.x) {
        const int64_t next = row + gridDim.x;
        const int64_t label = labels[row];
        const bool valid = is_valid(label);
        // a weighted row (ssi_ce_fwd_weighted): w * exp(x - lse) = exp2(x log2e - lse log2e + log2 w) — the weight rides in the exponent's
        // additive term at no cost per element; w = 1 (and no weights) adds an exact 0
        const float w = row_weight ? row_weight[row] : 1.f;
        const float log2w = row_weight ? __log2f(w) : 0.f;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(row);
        // columns that are not vocabulary (the pad columns [vocab, ld), and beyond the row where the loads returned 0) become -inf
        // once, in the registers: max, exp-sum and gradient (exp2(-inf) = 0) then need no column test at all
#pragma unroll
        for (int c = (NCH >= 2 ? NCH - 2 : 0); c < NCH; ++c) {
            const int left = vocab_i - c * CHUNK - col0;  // real columns from this lane's first one on
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (2 * d >= left) x[c][d] = (x[c][d] & 0xffff0000u) | 0x0000ff80u;
                if (2 * d + 1 >= left) x[c][d] = (x[c][d] & 0x0000ffffu) | 0xff800000u;
            }
        }
        float nl = -INFINITY;  // ignored / out-of-range label: every exp2 below gives 0 -> zero gradient row
        if (valid) {           // workgroup-uniform; reads the row registers only
            float xl = 0.f;
            if (tid == 0) xl = (float)logits[row * ld + label];  // before this row's gradient is written (program order of one thread)
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) m = fmaxf(m, fmaxf(lo(x[c][d]), hi(x[c][d])));
            // thread 0's label logit must have LANDED before any wave may overwrite that address with the gradient: the barrier inside
            // block_max only orders issue.  Free here: the row registers the max just consumed were the only other loads in flight.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float gm = block_max(m, red);
            opaque();
            const float nm = -gm * LOG2E;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    s += __builtin_amdgcn_exp2f(fmaf(lo(x[c][d]), LOG2E, nm)) + __builtin_amdgcn_exp2f(fmaf(hi(x[c][d]), LOG2E, nm));
            const float gs = block_sum(s, red);
            const float lse = gm + logf(gs);
            if (tid == 0) {
                row_loss[row] = w * (lse - xl);
                if (row_lse) row_lse[row] = lse;
            }
            nl = -lse * LOG2E + log2w;
        } else if (tid == 0) {
            row_loss[row] = 0.f;
            if (row_lse) row_lse[row] = 0.f;
        }
        opaque();
        // ---- gradient softmax - onehot, written over the logits; each register then takes the next row's chunk (past the last row:
        // this row again — a few wasted loads at the very end instead of a branch around every load)
        const int hot = valid ? (int)label - col0 : -(1 << 30);  // the label's column relative to this lane's first column of chunk 0
        const __amdgpu_buffer_rsrc_t rn = rsrc_of(next < rows ? next : row);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (write_grad) {
                float g[8];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    g[2 * d] = __builtin_amdgcn_exp2f(fmaf(lo(x[c][d]), LOG2E, nl));
                    g[2 * d + 1] = __builtin_amdgcn_exp2f(fmaf(hi(x[c][d]), LOG2E, nl));
                }
                const int h = hot - c * CHUNK;
                if ((unsigned)h < 8u) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e == h) g[e] -= w;
                }
                bf16x8 ob;
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)g[e];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), rs, voff, c * CHUNK * 2, 2 /* nt: streamed once */);
            }
            x[c] = __builtin_amdgcn_raw_buffer_load_b128(rn, voff, c * CHUNK * 2, 2 /* nt */);
            __builtin_amdgcn_sched_barrier(0);  // one chunk at a time: a hoisted next-row load would need a register of its own
        }
    }
}

static int ce_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();  // thread-safe one-time initialisation (C++11 magic static)
    return n;
}

extern "C" int ssi_ce_fwd_weighted(void* logits, int64_t ld, const int64_t* labels, const float* row_weight, int64_t rows, int64_t vocab,
                                   int64_t ignore_index, float* row_loss, float* row_lse, int write_grad, int dtype, void* stream) {
    SSI_CHECK_ARG(logits && labels && row_loss && rows >= 0 && vocab > 0 && ld >= vocab && ld % 8 == 0);
    if (rows == 0) return SSI_OK;
    const int64_t chunks = ssi_cdiv(ld, 8192);
    const bool row_form = dtype == SSI_BF16 && ((uintptr_t)logits & 15) == 0 && ld - vocab < 8192 && ld * 2 < (1LL << 31) &&
                          (chunks <= 4 || chunks == 8 || (chunks >= 16 && chunks <= 18));
    if (row_form) {
        const dim3 grid((unsigned)(rows < ce_num_cus() ? rows : ce_num_cus()));
#define SSI_CE_ROW(N)                                                                                                                       \
    case N:                                                                                                                                 \
        if (write_grad) hipLaunchKernelGGL((ce_row_bf16_kernel<N, true>), grid, dim3(1024), 0, (hipStream_t)stream, (bf16_t*)logits, ld, labels, \
                                           rows, vocab, ignore_index, row_loss, row_lse, row_weight);                                       \
        else hipLaunchKernelGGL((ce_row_bf16_kernel<N, false>), grid, dim3(1024), 0, (hipStream_t)stream, (bf16_t*)logits, ld, labels, rows,     \
                                vocab, ignore_index, row_loss, row_lse, row_weight);                                                        \
        break
        switch ((int)chunks) {
            SSI_CE_ROW(1); SSI_CE_ROW(2); SSI_CE_ROW(3); SSI_CE_ROW(4); SSI_CE_ROW(8); SSI_CE_ROW(16); SSI_CE_ROW(17); SSI_CE_ROW(18);
        }
#undef SSI_CE_ROW
        SSI_LAUNCH_CHECK();
        return SSI_OK;
    }
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(ce_fwd_kernel<T>, dim3((unsigned)rows), dim3(512), 0, (hipStream_t)stream,
                                                 (T*)logits, ld, labels, vocab, ignore_index, row_loss, row_lse, write_grad, row_weight));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

extern "C" int ssi_ce_fwd(void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab,
                          int64_t ignore_index, float* row_loss, float* row_lse, int write_grad, int dtype, void* stream) {
    return ssi_ce_fwd_weighted(logits, ld, labels, nullptr, rows, vocab, ignore_index, row_loss, row_lse, write_grad, dtype, stream);
}

// The valid-label count uses the predicate of the row kernels (not ignored AND inside [0, vocab)); labels that are neither ignored
// nor in range are counted in out[3] so that the caller can raise on its next host read-back (torch would device-assert).
__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                         int64_t rows, int64_t vocab, int64_t ignore_index, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f, bad = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 1024) {
        const int64_t l = labels[r];
        const bool in_range = l >= 0 && l < vocab;
        if (l != ignore_index && in_range) { s += row_loss[r]; c += 1.f; }
        else if (l != ignore_index) bad += 1.f;
    }
    s = block_sum(s, red);
    c = block_sum(c, red);
    bad = block_sum(bad, red);
    if (threadIdx.x == 0) { out[0] = s / c; out[1] = s; out[2] = c; out[3] = bad; }
}

extern "C" int ssi_ce_reduce(const float* row_loss, const int64_t* labels, int64_t rows, int64_t vocab, int64_t ignore_index,
                             float* out, void* stream) {
    SSI_CHECK_ARG(row_loss && labels && out && rows >= 0 && rows < (1LL << 24) && vocab > 0);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, labels, rows, vocab, ignore_index, out);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}
