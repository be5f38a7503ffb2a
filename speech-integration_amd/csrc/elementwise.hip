// Bandwidth-bound kernels of the training step: RMSNorm, RoPE, SwiGLU, token counting, grad scaling, sum of squares,
// AdamW.  All are HBM-roofline kernels: 16-byte per-lane accesses, one wave (64 lanes) per row for the row reductions,
// shuffles for the reductions, fp32 math with the reference's rounding points.
#include "common_hip.h"

// =====================================================================================================================
// K2 RMSNorm forward: one wave per row (torchtune.modules.RMSNorm.forward)
// =====================================================================================================================
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ scale,
                                                          T* __restrict__ y, float* __restrict__ rstd_out,
                                                          int64_t rows, int dim, float eps) {
    constexpr int N = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int nvec = dim / N;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), wave_stride = (int64_t)gridDim.x * 4;
    // each row is read once, into registers; the next row of the wave is in flight while this one is reduced and written
    Vec16<T> w[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int v = lane + k * 64;
        if (v < nvec) w[k] = load16(scale + v * N);
    }
    auto fetch = [&](int64_t row, Vec16<T>* a) {
        if (row >= rows) return;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) a[k] = load16(x + row * dim + v * N);
        }
    };
    auto finish = [&](int64_t row, const Vec16<T>* a) {
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) {
#pragma unroll
                for (int i = 0; i < N; ++i) { float f = a[k].get(i); ss += f * f; }
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)dim + eps);
        if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) {
                Vec16<T> o;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    // (x32 * rstd).type_as(x) * scale : round to storage type before the scale multiply
                    float xn = to_f32<T>(from_f32<T>(a[k].get(i) * rstd));
                    o.set(i, xn * w[k].get(i));
                }
                store16(y + row * dim + v * N, o);
            }
        }
    };
    Vec16<T> a0[MAXV], a1[MAXV];
    fetch(wave_global, a0);
    for (int64_t row = wave_global; row < rows; row += 2 * wave_stride) {
        fetch(row + wave_stride, a1);
        finish(row, a0);
        if (row + wave_stride >= rows) break;
        fetch(row + 2 * wave_stride, a0);
        finish(row + wave_stride, a1);
    }
}

// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ scale,
                                                          T* __restrict__ y, float* __restrict__ rstd_out,
                                                          int64_t rows, int dim, float eps) {
    constexpr int N = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * dim;
    T* yr = y + row * dim;
    const int nvec = dim / N;
    float ss = 0.f;
    for (int v = lane; v < nvec; v += 64) {
        Vec16<T> a = load16(xr + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) { float f = a.get(i); ss += f * f; }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)dim + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
    for (int v = lane; v < nvec; v += 64) {
        Vec16<T> a = load16(xr + v * N), w = load16(scale + v * N), o;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // (x32 * rstd).type_as(x) * scale : round to storage type before the scale multiply
            float xn = to_f32<T>(from_f32<T>(a.get(i) * rstd));
            o.set(i, xn * w.get(i));
        }
        store16(yr + v * N, o);
    }
}

// K2 RMSNorm backward.  Per wave: a strided set of rows; dscale partial sums kept per lane-column in registers and
// reduced over the block's 4 waves through LDS, then written as one partial row per block.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const T* __restrict__ scale, const float* __restrict__ rstd,
                                                          const T* __restrict__ dres, T* __restrict__ dx,
                                                          float* __restrict__ partials, int64_t rows, int dim) {
    constexpr int N = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][WS]
    constexpr int WS = MAXV * N * 64;  // floats per wave in the staging buffer (>= dim)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = dim / N;
    float dw[MAXV][N];
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) dw[k][i] = 0.f;

    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
    // A wave walks rows/waves rows one after the other (the dscale sums stay in its registers), so it alone would pay the whole
    // memory latency twice per row: each row is read ONCE, into registers, and the next row's loads are in flight while this
    // one is reduced and written (two register sets, alternating).
    struct RowRegs { Vec16<T> a[MAXV], g[MAXV], r[MAXV]; float rs; };
    Vec16<T> w[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int v = lane + k * 64;
        if (v < nvec) w[k] = load16(scale + v * N);
    }
    auto fetch = [&](int64_t row, RowRegs& q) {
        if (row >= rows) return;
        q.rs = rstd[row];
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) {
                q.a[k] = load16(x + row * dim + v * N);
                q.g[k] = load16(dy + row * dim + v * N);
                if (dres) q.r[k] = load16(dres + row * dim + v * N);
            }
        }
    };
    auto finish = [&](int64_t row, const RowRegs& q) {
        const float rs = q.rs;
        float c = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float xhat = q.a[k].get(i) * rs;
                    c += q.g[k].get(i) * w[k].get(i) * xhat;
                    dw[k][i] += q.g[k].get(i) * xhat;
                }
            }
        }
        c = wave_sum(c) / (float)dim;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = lane + k * 64;
            if (v < nvec) {
                Vec16<T> o;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float xhat = q.a[k].get(i) * rs;
                    float d = rs * (q.g[k].get(i) * w[k].get(i) - xhat * c);
                    if (dres) d += q.r[k].get(i);
                    o.set(i, d);
                }
                store16(dx + row * dim + v * N, o);
            }
        }
    };
    RowRegs q0, q1;
    fetch(wave_global, q0);
    for (int64_t row = wave_global; row < rows; row += 2 * wave_stride) {
        fetch(row + wave_stride, q1);
        finish(row, q0);
        if (row + wave_stride >= rows) break;
        fetch(row + 2 * wave_stride, q0);
        finish(row + wave_stride, q1);
    }
    // block reduce of dscale partials.  Staged [wave][k][i][lane]: consecutive lanes on consecutive banks.  The column-major form
    // [wave][column] (a lane's N consecutive floats = a 32-byte stride between lanes) was the one kernel of the step with LDS bank
    // conflicts (SQ_LDS_BANK_CONFLICT 14.4 % of its LDS cycles, profiles/r03_pmc_kernels.md); same sums in the same order.
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int v = lane + k * 64;
        if (v < nvec)
#pragma unroll
            for (int i = 0; i < N; ++i) lds[wave * WS + (k * N + i) * 64 + lane] = dw[k][i];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < WS; idx += 256) {
        const int l = idx & 63, ki = idx >> 6, v = l + (ki / N) * 64;
        if (v < nvec)
            partials[(int64_t)blockIdx.x * dim + v * N + ki % N] = lds[idx] + lds[WS + idx] + lds[2 * WS + idx] + lds[3 * WS + idx];
    }
}

// dscale[c] += sum_b partials[b][c]   (fixed order -> deterministic).  64 columns x 16 row groups per 1024-thread block: every
// thread has nblocks/16 independent loads in flight (with 4 row groups the 128-deep dependent chain made this launch cost as
// much as the 60x larger rmsnorm_bwd pass in front of it).
template <typename T>
__global__ __launch_bounds__(1024) void colsum_accum_kernel(const float* __restrict__ partials, T* __restrict__ dscale,
                                                            int nblocks, int dim, int accumulate) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (col < dim) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        int b = ry;
        for (; b + 48 < nblocks; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += partials[(int64_t)(b + 16 * u) * dim + col];
        }
        for (; b < nblocks; b += 16) part[0] += partials[(int64_t)b * dim + col];
        s = (part[0] + part[1]) + (part[2] + part[3]);
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < dim) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cx];
        dscale[col] = from_f32<T>(accumulate ? to_f32<T>(dscale[col]) + t : t);
    }
}

static inline int rmsnorm_bwd_blocks(int64_t rows) {
    // one 4-wave block per CU: with the row prefetch a wave hides its own latency, and fewer blocks mean fewer partial rows for the
    // column sum (measured at 16384 x 2048 bf16: 256 blocks 45.7 us, 512 50.3, 1024 60.3; before the prefetch 512 was best at 54.4)
    int64_t b = ssi_cdiv(rows, 4);
    return (int)(b < 256 ? b : 256);
}

extern "C" int64_t ssi_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
    return (int64_t)rmsnorm_bwd_blocks(rows) * dim * (int64_t)sizeof(float);
}

extern "C" int ssi_rmsnorm_fwd(const void* x, const void* scale, void* y, float* rstd, int64_t rows, int64_t dim,
                               float eps, int dtype, void* stream) {
    SSI_CHECK_ARG(x && scale && y && rows >= 0 && dim > 0 && dim % 8 == 0);
    if (rows == 0) return SSI_OK;
    const int64_t nb = ssi_cdiv(rows, 4) < 1024 ? ssi_cdiv(rows, 4) : 1024;  // 16384 x 2048 bf16: 1024 blocks 20.7 us, 256 23.7, 4096 21.2
    const int64_t vec_per_lane = ssi_cdiv(dim / (dtype == SSI_BF16 ? 8 : 4), 64);
    if (vec_per_lane > 8) { ssi_set_error("rmsnorm_fwd: dim %d too large", (int)dim); return SSI_ERR_UNSUPPORTED; }
#define SSI_RMS_FWD(MV)                                                                                                              \
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rmsnorm_fwd_kernel<T, MV>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, \
                                                 (const T*)x, (const T*)scale, (T*)y, rstd, rows, (int)dim, eps))
    if (vec_per_lane <= 1) { SSI_RMS_FWD(1); }
    else if (vec_per_lane <= 2) { SSI_RMS_FWD(2); }
    else if (vec_per_lane <= 4) { SSI_RMS_FWD(4); }
    else { SSI_RMS_FWD(8); }
#undef SSI_RMS_FWD
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

template <typename T>
static int launch_rmsnorm_bwd(const T* dy, const T* x, const T* scale, const float* rstd, const T* dres, T* dx, T* dscale,
                              int64_t rows, int dim, int nb, size_t lds_bytes, float* workspace, hipStream_t st, int accumulate) {
    const int64_t vec_per_lane = ssi_cdiv(dim / Vec16<T>::N, 64);
    const int maxv = vec_per_lane <= 1 ? 1 : vec_per_lane <= 2 ? 2 : vec_per_lane <= 4 ? 4 : 8;
    lds_bytes = (size_t)4 * maxv * Vec16<T>::N * 64 * sizeof(float);  // [4 waves][MAXV * N * 64]
    if (vec_per_lane <= 1)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 1>), dim3(nb), dim3(256), lds_bytes, st, dy, x, scale, rstd, dres, dx, workspace, rows, dim);
    else if (vec_per_lane <= 2)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 2>), dim3(nb), dim3(256), lds_bytes, st, dy, x, scale, rstd, dres, dx, workspace, rows, dim);
    else if (vec_per_lane <= 4)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 4>), dim3(nb), dim3(256), lds_bytes, st, dy, x, scale, rstd, dres, dx, workspace, rows, dim);
    else if (vec_per_lane <= 8)
        hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, 8>), dim3(nb), dim3(256), lds_bytes, st, dy, x, scale, rstd, dres, dx, workspace, rows, dim);
    else { ssi_set_error("rmsnorm_bwd: dim %d too large", dim); return SSI_ERR_UNSUPPORTED; }
    SSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_accum_kernel<T>, dim3((unsigned)ssi_cdiv(dim, 64)), dim3(1024), 0, st, (const float*)workspace, dscale, nb, dim, accumulate);
    return SSI_OK;
}

extern "C" int ssi_rmsnorm_bwd(const void* dy, const void* x, const void* scale, const float* rstd, const void* dres,
                               void* dx, void* dscale, int accumulate_dscale, int64_t rows, int64_t dim, int dtype, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    SSI_CHECK_ARG(dy && x && scale && rstd && dx && dscale && rows >= 0 && dim > 0 && dim % 8 == 0);
    if (rows == 0) return SSI_OK;
    const int nb = rmsnorm_bwd_blocks(rows);
    if (!workspace || workspace_bytes < ssi_rmsnorm_bwd_workspace_bytes(rows, dim)) {
        ssi_set_error("rmsnorm_bwd: workspace too small");
        return SSI_ERR_WORKSPACE;
    }
    const size_t lds_bytes = 4 * ssi_align_up(dim, 512) * sizeof(float);  // upper bound of what launch_rmsnorm_bwd asks for
    SSI_CHECK_ARG(dim <= 8 * 8 * 64 && 2 * lds_bytes <= 160 * 1024);
    int rc = SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, rc = launch_rmsnorm_bwd<T>((const T*)dy, (const T*)x, (const T*)scale, rstd, (const T*)dres,
                                                         (T*)dx, (T*)dscale, rows, (int)dim, nb, lds_bytes,
                                                         (float*)workspace, (hipStream_t)stream, accumulate_dscale));
    if (rc) return rc;
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K4 RoPE (Llama3ScaledRoPE.forward): adjacent pairs, fp32 math, in place.  One 16-byte vector per thread.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ x, int64_t ld, int64_t rows, int64_t seq_len,
                                                   int rot_width, int head_dim, const float* __restrict__ table,
                                                   const int32_t* __restrict__ positions, float sign) {
    constexpr int N = Vec16<T>::N;
    const int vec_per_row = rot_width / N;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows * vec_per_row) return;
    const int64_t row = gid / vec_per_row;
    const int col = (int)(gid % vec_per_row) * N;
    const int64_t pos = positions ? (int64_t)positions[row] : (row % seq_len);
    const int pair0 = (col % head_dim) >> 1;
    const float* tb = table + (pos * (head_dim >> 1) + pair0) * 2;
    T* p = x + row * ld + col;
    Vec16<T> a = load16(p), o;
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const float c = tb[i], s = tb[i + 1] * sign;
        const float x0 = a.get(i), x1 = a.get(i + 1);
        float o0, o1;
        ssi_rope_pair(x0, x1, c, s, o0, o1);
        o.set(i, o0);
        o.set(i + 1, o1);
    }
    store16(p, o);
}

extern "C" int ssi_rope_inplace(void* x, int64_t ld, int64_t rows, int64_t seq_len, int n_heads_rot, int head_dim,
                                const float* table, int64_t table_len, const int32_t* positions, int inverse, int dtype,
                                void* stream) {
    SSI_CHECK_ARG(x && table && rows >= 0 && seq_len > 0 && n_heads_rot > 0 && head_dim > 0 && head_dim % 8 == 0);
    SSI_CHECK_ARG(ld % 8 == 0 && (int64_t)n_heads_rot * head_dim <= ld);
    SSI_CHECK_ARG(positions != nullptr || table_len >= seq_len);
    if (rows == 0) return SSI_OK;
    const int rot_width = n_heads_rot * head_dim;
    SSI_DISPATCH_DTYPE(dtype, {
        const int64_t total = rows * (rot_width / Vec16<T>::N);
        hipLaunchKernelGGL(rope_kernel<T>, dim3((unsigned)ssi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (T*)x,
                           ld, rows, seq_len, rot_width, head_dim, table, positions, inverse ? -1.f : 1.f);
    });
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K7 SwiGLU elementwise (torchtune FeedForward middle): act = silu(gate) * up ; gu = [gate | up]
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ gu, T* __restrict__ act, int64_t rows,
                                                         int64_t inter) {
    constexpr int N = Vec16<T>::N;
    const int64_t vpr = inter / N;
    for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < rows * vpr; gid += (int64_t)gridDim.x * 256) {
        const int64_t row = gid / vpr, col = (gid % vpr) * N;
        Vec16<T> g = load16(gu + row * 2 * inter + col), u = load16(gu + row * 2 * inter + inter + col), o;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float gf = g.get(i);
            const float s = to_f32<T>(from_f32<T>(ssi_silu<T>(gf)));  // F.silu result rounded to storage type
            o.set(i, s * u.get(i));
        }
        store16(act + row * inter + col, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ dact, const T* __restrict__ gu,
                                                         T* __restrict__ dgu, int64_t rows, int64_t inter) {
    constexpr int N = Vec16<T>::N;
    const int64_t vpr = inter / N;
    for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < rows * vpr; gid += (int64_t)gridDim.x * 256) {
        const int64_t row = gid / vpr, col = (gid % vpr) * N;
        Vec16<T> g = load16(gu + row * 2 * inter + col), u = load16(gu + row * 2 * inter + inter + col);
        Vec16<T> d = load16(dact + row * inter + col), og, ou;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float dg, du;
            ssi_swiglu_bwd_elem<T>(g.get(i), u.get(i), d.get(i), dg, du);
            ou.set(i, du);
            og.set(i, dg);
        }
        store16(dgu + row * 2 * inter + col, og);
        store16(dgu + row * 2 * inter + inter + col, ou);
    }
}

static inline unsigned stream_grid(int64_t work_items, int64_t cap = 8192) {
    int64_t b = ssi_cdiv(work_items, 256);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int ssi_swiglu_fwd(const void* gu, void* act, int64_t rows, int64_t inter, int dtype, void* stream) {
    SSI_CHECK_ARG(gu && act && rows >= 0 && inter > 0 && inter % 8 == 0);
    if (rows == 0) return SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(swiglu_fwd_kernel<T>, dim3(stream_grid(rows * inter / Vec16<T>::N)),
                                                 dim3(256), 0, (hipStream_t)stream, (const T*)gu, (T*)act, rows, inter));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

extern "C" int ssi_swiglu_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int64_t inter, int dtype,
                              void* stream) {
    SSI_CHECK_ARG(dact && gu && dgu && rows >= 0 && inter > 0 && inter % 8 == 0);
    if (rows == 0) return SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(swiglu_bwd_kernel<T>, dim3(stream_grid(rows * inter / Vec16<T>::N)),
                                                 dim3(256), 0, (hipStream_t)stream, (const T*)dact, (const T*)gu,
                                                 (T*)dgu, rows, inter));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K14 token-type counts + valid-label count in ONE launch (replaces 6-7 .sum().item() syncs, trainer.py:388,391)
// =====================================================================================================================
#define SSI_MAX_RANGES 8
__global__ __launch_bounds__(256) void count_tokens_kernel(const int64_t* __restrict__ tokens,
                                                           const int64_t* __restrict__ labels, int64_t n,
                                                           const int64_t* __restrict__ ranges, int n_ranges,
                                                           int64_t pad_id, int64_t ignore_index,
                                                           unsigned long long* __restrict__ out) {
    __shared__ unsigned int blk[SSI_MAX_RANGES + 2];
    if (threadIdx.x < SSI_MAX_RANGES + 2) blk[threadIdx.x] = 0;
    __syncthreads();
    int64_t lo[SSI_MAX_RANGES], hi[SSI_MAX_RANGES];
    unsigned int cnt[SSI_MAX_RANGES + 2];
#pragma unroll
    for (int r = 0; r < SSI_MAX_RANGES; ++r) {
        lo[r] = r < n_ranges ? ranges[2 * r] : 1;
        hi[r] = r < n_ranges ? ranges[2 * r + 1] : 0;
    }
#pragma unroll
    for (int r = 0; r < SSI_MAX_RANGES + 2; ++r) cnt[r] = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = tokens[i];
#pragma unroll
        for (int r = 0; r < SSI_MAX_RANGES; ++r) cnt[r] += (t >= lo[r] && t <= hi[r]) ? 1u : 0u;
        cnt[SSI_MAX_RANGES] += (t != pad_id) ? 1u : 0u;
        if (labels) cnt[SSI_MAX_RANGES + 1] += (labels[i] != ignore_index) ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < SSI_MAX_RANGES + 2; ++r) {
        unsigned int v = cnt[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&blk[r], v);
    }
    __syncthreads();
    if (threadIdx.x < SSI_MAX_RANGES + 2) {
        const int r = threadIdx.x;
        const int dst = r < SSI_MAX_RANGES ? r : n_ranges + (r - SSI_MAX_RANGES);
        if ((r >= SSI_MAX_RANGES || r < n_ranges) && blk[r]) atomicAdd(&out[dst], (unsigned long long)blk[r]);
    }
}

extern "C" int ssi_count_tokens(const int64_t* tokens, const int64_t* labels, int64_t n, const int64_t* ranges,
                                int n_ranges, int64_t pad_id, int64_t ignore_index, int64_t* out, void* stream) {
    SSI_CHECK_ARG((tokens || n == 0) && out && n >= 0 && n_ranges >= 0 && n_ranges <= SSI_MAX_RANGES && (ranges || n_ranges == 0));
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int64_t) * (n_ranges + 2), (hipStream_t)stream);
    if (e != hipSuccess) { ssi_set_error("count_tokens: memset failed: %s", hipGetErrorString(e)); return SSI_ERR_HIP + (int)e; }
    if (n == 0) return SSI_OK;
    int64_t b = ssi_cdiv(n, 256);
    if (b > 256) b = 256;
    hipLaunchKernelGGL(count_tokens_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, tokens, labels, n, ranges,
                       n_ranges, pad_id, ignore_index, (unsigned long long*)out);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// Packed rows: positions + document bounds from input_pos in ONE launch (the model did this with ~10 torch ops per step).
// One 256-thread workgroup per row; a document starts where input_pos == 0 (and at position 0 of the row).  Each thread owns a
// contiguous span: last start at or before it (forward) and first start after it (backward) come from two LDS scans over the
// per-thread summaries, then the span is filled.  Integer work: results are exact.
// =====================================================================================================================
__global__ __launch_bounds__(256) void doc_ranges_kernel(const int64_t* __restrict__ input_pos, int64_t seq, int max_pos,
                                                         int32_t* __restrict__ pos, int32_t* __restrict__ doc_start,
                                                         int32_t* __restrict__ doc_end, int32_t* __restrict__ n_clamped) {
    __shared__ int last_start[256], first_start[256];
    const int64_t row = blockIdx.x;
    const int64_t* ip = input_pos + row * seq;
    const int t = threadIdx.x;
    const int span = (int)((seq + 255) / 256);
    const int lo = t * span, hi = (int)(lo + span < seq ? lo + span : seq);
    int ls = -1, fs = (int)seq;  // last start inside my span, first start inside my span
    for (int s = lo; s < hi; ++s) {
        const bool st = s == 0 || ip[s] == 0;
        if (st) { ls = s; if (fs == (int)seq) fs = s; }
    }
    last_start[t] = ls;
    first_start[t] = fs;
    __syncthreads();
    int before = -1;             // last start in the spans before mine
    for (int u = t - 1; u >= 0 && before < 0; --u) before = last_start[u];
    int after = (int)seq;        // first start in the spans after mine
    for (int u = t + 1; u < 256 && after == (int)seq; ++u) after = first_start[u];
    // forward: document start of every position of my span
    int cur = before, clamped = 0;
    for (int s = lo; s < hi; ++s) {
        if (s == 0 || ip[s] == 0) cur = s;
        doc_start[row * seq + s] = cur;
        const int64_t p = ip[s];
        clamped += (p < 0 || p > max_pos) ? 1 : 0;
        pos[row * seq + s] = (int32_t)(p < 0 ? 0 : (p > max_pos ? max_pos : p));
    }
    if (n_clamped && clamped) atomicAdd(n_clamped, clamped);  // integer: exact whatever the order
    // backward: one past the last position of the document = the first start strictly after s
    int nxt = after;
    for (int s = hi - 1; s >= lo; --s) {
        doc_end[row * seq + s] = nxt;
        if (s == 0 || ip[s] == 0) nxt = s;
    }
}

extern "C" int ssi_doc_ranges(const int64_t* input_pos, int64_t batch, int64_t seq, int64_t max_pos, int32_t* positions,
                              int32_t* doc_start, int32_t* doc_end, int32_t* n_clamped, void* stream) {
    SSI_CHECK_ARG(input_pos && positions && doc_start && doc_end && batch >= 0 && seq >= 0 && seq < (1LL << 31) && max_pos >= 0 && max_pos < (1LL << 31));
    if (batch == 0 || seq == 0) return SSI_OK;
    hipLaunchKernelGGL(doc_ranges_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, input_pos, seq, (int)max_pos, positions,
                       doc_start, doc_end, n_clamped);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// K11 scale_grads, K12 sum of squares (for clip_grad_norm_), K13 AdamW — flat-buffer streaming kernels
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, int64_t n, float scale, const float* scale_dev) {
    constexpr int N = Vec16<T>::N;
    const float s = scale * (scale_dev ? *scale_dev : 1.f);
    const int64_t nvec = n / N;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        Vec16<T> a = load16(x + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) a.set(i, a.get(i) * s);
        store16(x + v * N, a);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - nvec * N)) {
        const int64_t i = nvec * N + threadIdx.x;
        x[i] = from_f32<T>(to_f32<T>(x[i]) * s);
    }
}

extern "C" int ssi_scale_inplace(void* x, int64_t n, float scale, const float* scale_dev, int dtype, void* stream) {
    SSI_CHECK_ARG(x && n >= 0 && ((uintptr_t)x % 16) == 0);
    if (n == 0) return SSI_OK;
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(scale_kernel<T>, dim3(stream_grid(n / Vec16<T>::N + 1)), dim3(256), 0,
                                                 (hipStream_t)stream, (T*)x, n, scale, scale_dev));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

#define SSI_SUMSQ_BLOCKS 1024
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ x, int64_t n, float* __restrict__ part) {
    constexpr int N = Vec16<T>::N;
    __shared__ float red[16];
    const int64_t nvec = n / N;
    float s = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        Vec16<T> a = load16(x + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) { const float f = a.get(i); s += f * f; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - nvec * N)) {
        const float f = to_f32<T>(x[nvec * N + threadIdx.x]);
        s += f * f;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ float red[16];
    float s = (int)threadIdx.x < nb ? part[threadIdx.x] : 0.f;
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
extern "C" int64_t ssi_sumsq_workspace_bytes(int64_t) { return SSI_SUMSQ_BLOCKS * sizeof(float); }
extern "C" int ssi_sumsq(const void* x, int64_t n, int dtype, float* out, void* workspace, int64_t workspace_bytes,
                         void* stream) {
    SSI_CHECK_ARG(x && out && n >= 0 && ((uintptr_t)x % 16) == 0);
    if (!workspace || workspace_bytes < ssi_sumsq_workspace_bytes(n)) { ssi_set_error("sumsq: workspace too small"); return SSI_ERR_WORKSPACE; }
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3(SSI_SUMSQ_BLOCKS), dim3(256), 0,
                                                 (hipStream_t)stream, (const T*)x, n, (float*)workspace));
    SSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace,
                       SSI_SUMSQ_BLOCKS, out);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ p, T* __restrict__ g, T* __restrict__ m,
                                                    T* __restrict__ v, int64_t n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float step_size, float inv_bc2_sqrt,
                                                    const float* __restrict__ grad_scale_dev, int zero_grad) {
    constexpr int N = Vec16<T>::N;
    const float gs = grad_scale_dev ? *grad_scale_dev : 1.f;
    if ((zero_grad & 2) && !(fabsf(gs) < INFINITY)) return;  // bit 1: a non-finite scale (an accumulation window without a label) changes nothing
    zero_grad &= 1;
    const int64_t nvec = n / N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        Vec16<T> pp = load16_nt(p + i * N), gg = load16_nt(g + i * N), mm = load16_nt(m + i * N), vv = load16_nt(v + i * N);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float gr = gg.get(k) * gs;
            float pf = pp.get(k);
            pf -= lr * wd * pf;
            const float mf = mm.get(k) + (1.f - beta1) * (gr - mm.get(k));
            const float vf = beta2 * vv.get(k) + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(vf) * inv_bc2_sqrt + eps;
            pf -= step_size * mf / denom;
            pp.set(k, pf); mm.set(k, mf); vv.set(k, vf);
            if (zero_grad) gg.set(k, 0.f);
        }
        store16_nt(p + i * N, pp); store16_nt(m + i * N, mm); store16_nt(v + i * N, vv);
        if (zero_grad) store16_nt(g + i * N, gg);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - nvec * N)) {
        const int64_t i = nvec * N + threadIdx.x;
        const float gr = to_f32<T>(g[i]) * gs;
        float pf = to_f32<T>(p[i]);
        pf -= lr * wd * pf;
        const float mf = to_f32<T>(m[i]) + (1.f - beta1) * (gr - to_f32<T>(m[i]));
        const float vf = beta2 * to_f32<T>(v[i]) + (1.f - beta2) * gr * gr;
        pf -= step_size * mf / (sqrtf(vf) * inv_bc2_sqrt + eps);
        p[i] = from_f32<T>(pf); m[i] = from_f32<T>(mf); v[i] = from_f32<T>(vf);
        if (zero_grad) g[i] = from_f32<T>(0.f);
    }
}

extern "C" int ssi_adamw_step(void* param, void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale_dev,
                              int zero_grad, int dtype, void* stream) {
    SSI_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1);
    SSI_CHECK_ARG(((uintptr_t)param % 16) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                  ((uintptr_t)exp_avg_sq % 16) == 0);
    if (n == 0) return SSI_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    // 17 GB touched once: non-temporal accesses and as many short blocks as there are 16-byte vectors (no grid-stride loop at the model's
    // size: 608 k blocks).  tools/adamw_bench.py, 1.246 G bf16 parameters: 8192 long-lived blocks 5.5 TB/s, 32768 6.2, 131072 6.3, one vector
    // per thread 6.47 TB/s (2.70 ms); two vectors per thread with all eight loads in flight first: 6.1
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(adamw_kernel<T>, dim3(stream_grid(n / Vec16<T>::N + 1, 1 << 20)), dim3(256), 0,
                                                 (hipStream_t)stream, (T*)param, (T*)grad, (T*)exp_avg, (T*)exp_avg_sq, n,
                                                 lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
                                                 grad_scale_dev, zero_grad));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// =====================================================================================================================
// 2-D transpose dst[c][r] = src[r][c] (64 x 64 tiles through LDS; 16-B global accesses on both sides).  Used once per
// optimizer step to refresh the [in, out] copies of the projection weights, so that the data-gradient GEMMs
// dX = dY W run in the k-contiguous (NT) operand form instead of the transposed-read form.
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst, int64_t ldd,
                                                        int64_t rows, int64_t cols) {
    constexpr int N = Vec16<T>::N;          // elements per 16 B
    constexpr int TS = 64;                  // tile edge
    __shared__ T tile[TS][TS + 2 * N / N + 2];  // +pad breaks the power-of-two row stride
    const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
    constexpr int VPR = TS / N;             // vectors per tile row
    for (int v = threadIdx.x; v < TS * VPR; v += 256) {
        const int r = v / VPR, cv = (v % VPR) * N;
        if (r0 + r < rows && c0 + cv < cols) {
            Vec16<T> a = load16(src + (r0 + r) * lds_ + c0 + cv);
#pragma unroll
            for (int i = 0; i < N; ++i) tile[cv + i][r] = from_f32<T>(a.get(i));
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < TS * VPR; v += 256) {
        const int c = v / VPR, rv = (v % VPR) * N;
        if (c0 + c < cols && r0 + rv < rows) {
            Vec16<T> o;
#pragma unroll
            for (int i = 0; i < N; ++i) o.set(i, to_f32<T>(tile[c][rv + i]));
            store16(dst + (c0 + c) * ldd + r0 + rv, o);
        }
    }
}

extern "C" int ssi_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols, int dtype,
                             void* stream) {
    SSI_CHECK_ARG(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows);
    SSI_CHECK_ARG(rows % 8 == 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0);
    if (rows == 0 || cols == 0) return SSI_OK;
    SSI_CHECK_ARG(ssi_cdiv(rows, 64) <= 65535);
    dim3 grid((unsigned)ssi_cdiv(cols, 64), (unsigned)ssi_cdiv(rows, 64));
    SSI_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)src, ld_src,
                                                 (T*)dst, ld_dst, rows, cols));
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}
