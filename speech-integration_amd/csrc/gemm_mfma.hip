// bf16 MFMA GEMM for gfx950 (MI355X): the dense contractions of the training step — QKV / output / gate-up / down
// projections, the tied LM head, and their data- and weight-gradients.
//
// Tile: 256 x 256 x 64 per 512-thread workgroup (8 waves as 2(M) x 4(N), 128 x 64 outputs per wave,
//       32 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 128 accumulator VGPRs per lane).
// LDS : 2 buffers x (A 32 KiB + B 32 KiB) = 128 KiB for the K pipeline, reused (144 KiB total) by the epilogue.
// HBM -> LDS: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), tile t+2 issued as soon as tile t has been read,
//       one barrier per K-tile; inside a tile every block of 16 MFMAs overlaps the ds_reads of the next block.  LDS images are lane-linear; the bank swizzle is applied on the per-lane SOURCE address
//       and again on the read address (same involution both sides).
// Operand forms (all three layouts of ssi_gemm use the same main loop):
//   ROW  tile [rows][64 k]  (k contiguous in memory): fragments by ds_read_b128, 16-B chunk c of row r stored at
//        chunk c ^ ((r >> 1) & 7)  -> conflict-free for the 16-lane ds_read_b128 groups.
//   COL  tile [64 k][cols]  (k strided in memory: NN's B, TN's A and B): fragments by 2 x ds_read_b64_tr_b16 (the
//        hardware transpose read), 16-B chunk c of k-row r stored at chunk c ^ (g(r) << 1),
//        g(r) = (r & 3) | (((r >> 3) & 1) << 2)  -> the 8 k-rows one half-wave touches land on 8 distinct 32-B slots.
// Epilogue: accumulators are produced with swapped MFMA operands (D' = B.A^T), so each lane owns 4 consecutive output
//       columns; they are scaled, rounded to bf16, staged through LDS and written as full 128-B row segments, adding
//       the residual / previous C in the same pass (same rounding points as F.linear followed by `+`).
// Workgroup -> tile map: XCD-aware (consecutive tiles of a group share A rows / B columns inside one XCD's L2).
#include "common.cuh"

int ssi_get_impl();

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int WAVES_N = 4;                                   // 2 x 4 wave grid
constexpr int WM = 128, WN = 64;                             // per-wave output
constexpr int MT = WM / 16, NT = WN / 16;                    // 8 x 4 accumulator tiles
constexpr int TILE_BYTES = BM * BK * 2;                      // 32 KiB per operand tile
constexpr int PIPE_BYTES = 2 * 2 * TILE_BYTES;               // 128 KiB
constexpr int EPI_ROW_BYTES = WN * 2 + 16;                   // 144 B padded row of a wave's bf16 output tile
constexpr int EPI_WAVE_BYTES = WM * EPI_ROW_BYTES;           // 18 KiB
constexpr int LDS_BYTES = 8 * EPI_WAVE_BYTES > PIPE_BYTES ? 8 * EPI_WAVE_BYTES : PIPE_BYTES;  // 144 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ int col_swz(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

// ---- HBM -> LDS staging of one operand tile ------------------------------------------------------------------------
// ROW: rows r0..r0+255 of a [*, ld] matrix, k columns k0..k0+63.  COL: k-rows k0..k0+63, columns c0..c0+255.
// k-strided (COL) operand tiles of the weight-gradient form go global -> VGPR -> ds_write_b128: measured on this chip,
// LDS-DMA writes in flight slow ds_read_b64_tr_b16 down (TN 1.05 -> 1.17 PFLOP/s with register staging), while plain
// ds_read_b64/b128 are unaffected.
__device__ __forceinline__ void stage_col_load(const bf16_t* __restrict__ g, int64_t ld, int64_t r0, int64_t k0, int tid,
                                               u32x4 (&regs)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int krow = p * 16 + (tid >> 5);
        const int chunk = (tid & 31) ^ col_swz(krow);
        regs[p] = *reinterpret_cast<const u32x4*>(g + (k0 + krow) * ld + r0 + chunk * 8);
    }
}
__device__ __forceinline__ void stage_col_write(char* lds_tile, int tid, const u32x4 (&regs)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(lds_tile + p * 8192 + tid * 16) = regs[p];
}

template <bool COL>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t ld, int64_t r0, int64_t k0, char* lds_tile,
                                           int tid, int64_t split_rows = 0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const bf16_t* src;
        if (!COL) {
            const int row = p * 64 + (tid >> 3);
            const int chunk = (tid & 7) ^ ((row >> 1) & 7);
            // split_rows = I: tile rows 0..127 come from gate rows r0/2.., rows 128..255 from the matching up rows I + r0/2..
            const int64_t grow = split_rows ? (row < 128 ? (r0 >> 1) + row : split_rows + (r0 >> 1) + row - 128) : r0 + row;
            src = g + grow * ld + k0 + chunk * 8;
        } else {
            const int krow = p * 16 + (tid >> 5);
            const int chunk = (tid & 31) ^ col_swz(krow);
            src = g + (k0 + krow) * ld + r0 + chunk * 8;
        }
        // wave-uniform LDS base; the hardware adds lane * 16
        const int wave_base = __builtin_amdgcn_readfirstlane(p * 8192 + (tid >> 6) * 1024);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_tile + wave_base), 16, 0, 0);
    }
}

// ---- LDS -> MFMA fragment --------------------------------------------------------------------------------------------
// fragment of 16 "outer" indices (rows of A / columns of B) x 32 k: lane l holds outer = base + (l & 15), k = kh*32 + 8*(l>>4) + j
template <bool COL>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int outer_base, int kh, int lane) {
    if (!COL) {
        const int row = outer_base + (lane & 15);
        const int chunk = (kh * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + chunk * 16);
    } else {
        // two transposed reads of 4 k-rows x 16 columns each; in a 16-lane group lane 4q+p addresses k-row q, cols 4p..4p+3
        const int i = lane & 15, q = i >> 2, p = i & 3;
        const int kbase = kh * 32 + 8 * (lane >> 4);
        const int chunk = (outer_base >> 3) + (p >> 1);
        const int k0 = kbase + q, k1 = kbase + 4 + q;
        const char* a0 = lds_tile + k0 * 512 + ((chunk ^ col_swz(k0)) * 16) + 8 * (p & 1);
        const char* a1 = lds_tile + k1 * 512 + ((chunk ^ col_swz(k1)) * 16) + 8 * (p & 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    }
}

__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    // XCD-aware remap (bijective for any grid size): blocks b and b+8 share an XCD, give each XCD a contiguous span
    const int nwg = tiles_m * tiles_n;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // grouped order: GM m-tiles x all n-tiles per group
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int group = t / per_group, in_group = t % per_group;
    const int gm = (tiles_m - group * GM) < GM ? (tiles_m - group * GM) : GM;
    tm = group * GM + in_group % gm;
    tn = in_group / gm;
}

// SPLITK: blockIdx.y selects a contiguous range of K-tiles; the fp32 partial tile goes to slab[blockIdx.y] (an [M, N] fp32
// matrix in the workspace) and splitk_reduce_kernel applies alpha / accumulate / residual and the bf16 rounding.
// EPI: 0 plain epilogue; 1 SwiGLU forward (the 256 output columns of a tile are 128 gate + the matching 128 up columns of
// W13; writes GU = [gate | up] and ACT = silu(gate) * up); 2 SwiGLU backward (C tile = d act, never stored: reads gate/up
// from GU and writes d gate / d up into DGU).  Same rounding points as the separate swiglu kernels (bf16 GEMM result first).
enum { EPI_PLAIN = 0, EPI_SWIGLU_FWD = 1, EPI_SWIGLU_BWD = 2 };
struct EpiArgs {
    bf16_t* out2;        // FWD: ACT [M, I]      BWD: DGU [M, 2I]
    int64_t ld_out2;
    const bf16_t* in2;   // BWD: GU [M, 2I]
    int64_t ld_in2;
    int64_t inter;       // I
};

template <bool A_COL, bool B_COL, bool SPLITK, int EPI = EPI_PLAIN>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_mfma_kernel(int tiles_m, int tiles_n, int64_t K,
                                                               const bf16_t* __restrict__ A, int64_t lda,
                                                               const bf16_t* __restrict__ B, int64_t ldb,
                                                               bf16_t* __restrict__ C, int64_t ldc,
                                                               const bf16_t* __restrict__ R, float alpha,
                                                               const float* __restrict__ alpha_dev, int accumulate,
                                                               float* __restrict__ slabs, EpiArgs ea) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_total = (int)(K / BK);
    const int kt_begin = SPLITK ? (int)((int64_t)blockIdx.y * nk_total / gridDim.y) : 0;
    const int kt_end = SPLITK ? (int)((int64_t)(blockIdx.y + 1) * nk_total / gridDim.y) : nk_total;
    const int nk = kt_end - kt_begin;
    const int64_t kofs = (int64_t)kt_begin * BK;
    auto tileA = [&](int buf) { return smem + buf * 2 * TILE_BYTES; };
    auto tileB = [&](int buf) { return smem + buf * 2 * TILE_BYTES + TILE_BYTES; };
    auto stage = [&](int kt, int buf) {
        stage_tile<A_COL>(A, lda, m0, kofs + (int64_t)kt * BK, tileA(buf), tid);
        stage_tile<B_COL>(B, ldb, n0, kofs + (int64_t)kt * BK, tileB(buf), tid, EPI == EPI_SWIGLU_FWD ? ea.inter : 0);
    };
    if constexpr (A_COL && B_COL) {
        // ---- weight-gradient form: both operands register-staged, one tile ahead; fragments of a whole k-half at once ----
        u32x4 ra[4], rb[4];
        stage_col_load(A, lda, m0, kofs, tid, ra);
        stage_col_load(B, ldb, n0, kofs, tid, rb);
        stage_col_write(tileA(0), tid, ra);
        stage_col_write(tileB(0), tid, rb);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            __syncthreads();  // tile kt visible; every wave is done reading buffer cur^1
            if (kt + 1 < nk) {
                stage_col_load(A, lda, m0, kofs + (int64_t)(kt + 1) * BK, tid, ra);
                stage_col_load(B, ldb, n0, kofs + (int64_t)(kt + 1) * BK, tid, rb);
            }
            const char* la = tileA(cur);
            const char* lb = tileB(cur);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 bfr[NT], afr[MT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bfr[j] = read_frag<true>(lb, wn * WN + j * 16, kh, lane);
#pragma unroll
                for (int i = 0; i < MT; ++i) afr[i] = read_frag<true>(la, wm * WM + i * 16, kh, lane);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], afr[i], acc[j][i], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            if (kt + 1 < nk) {
                stage_col_write(tileA(cur ^ 1), tid, ra);
                stage_col_write(tileB(cur ^ 1), tid, rb);
            }
        }
    } else {
    // Fragment registers: the wave's 128 x 64 tile is walked as 2 row halves (4 m-tiles each) x 2 k-halves per K-tile.
    // a0/a1 alternate between the row halves, b0/b1 between the k-halves; the ds_reads of the NEXT block are issued one
    // per MFMA of the current block (sched_group_barrier), so LDS reads stream continuously under the matrix pipe
    // (measured +6 % on the k-contiguous form; the transposed-read forms are bound by ds_read_b64_tr_b16 itself:
    // replacing them by plain ds_read_b64 of the same addresses recovers the k-contiguous rate — DESIGN.md).
    bf16x8 a0[4], a1[4], b0[4], b1[4];
    auto readA = [&](bf16x8 (&dst)[4], const char* la, int half, int kh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = read_frag<A_COL>(la, wm * WM + (half * 4 + i) * 16, kh, lane);
    };
    auto readB = [&](bf16x8 (&dst)[4], const char* lb, int kh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = read_frag<B_COL>(lb, wn * WN + j * 16, kh, lane);
    };
    auto mma = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[4], int half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j][half * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][half * 4 + i], 0, 0, 0);
    };
    // ask the scheduler for: 1 MFMA, then up to RPM ds_reads, 16 times (reads of the NEXT block ride between the MFMAs)
#define INTERLEAVE(RPM)                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);                   \
    }

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // both tiles landed
    readA(a0, tileA(0), 0, 0);
    readB(b0, tileB(0), 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* la = tileA(cur);
        const char* lb = tileB(cur);
        readA(a1, la, 1, 0);
        mma(a0, b0, 0);
        INTERLEAVE(1);
        __builtin_amdgcn_sched_barrier(0);
        readA(a0, la, 0, 1);
        readB(b1, lb, 1);
        mma(a1, b0, 1);
        INTERLEAVE(1);
        __builtin_amdgcn_sched_barrier(0);
        readA(a1, la, 1, 1);
        mma(a0, b1, 0);
        INTERLEAVE(1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            // hipcc does not reliably order LDS-DMA against later ds_reads of another buffer: wait explicitly
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // every wave has read all of tile kt; tile kt+1 has landed
            if (kt + 2 < nk) stage(kt + 2, cur);
            readA(a0, tileA(cur ^ 1), 0, 0);
            readB(b0, tileB(cur ^ 1), 0);
        }
        mma(a1, b1, 1);
        INTERLEAVE(1);
        __builtin_amdgcn_sched_barrier(0);
    }

    }

    // ---- epilogue: acc[j][i] holds C[m = wm*128 + i*16 + (lane&15)][n = wn*64 + j*16 + (lane>>4)*4 + r] ----------------
    if (SPLITK) {  // fp32 partials, 4 consecutive columns per lane (64-B row segments per 16-lane group)
        float* slab = slabs + (int64_t)blockIdx.y * ((int64_t)tiles_m * BM) * ((int64_t)tiles_n * BN);
        const int64_t ldn = (int64_t)tiles_n * BN;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t row = m0 + wm * WM + i * 16 + (lane & 15), col = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
                *reinterpret_cast<f32x4*>(slab + row * ldn + col) = acc[j][i];
            }
        return;
    }
    __syncthreads();
    const float al = alpha * (alpha_dev ? *alpha_dev : 1.f);
    char* ep = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[j][i][r] * al);
            const int row = i * 16 + (lane & 15), col = j * 16 + (lane >> 4) * 4;
            *reinterpret_cast<bf16x4*>(ep + row * EPI_ROW_BYTES + col * 2) = v;
        }
    if (EPI == EPI_SWIGLU_FWD) {
        // tile columns 0..127 = gate (waves wn 0,1), 128..255 = the matching up columns (waves wn 2,3).  Each wave finishes 64
        // rows of a (gate, up) pair of 64-column blocks: GU = [gate | up], ACT = silu(gate).to(bf16) * up
        __syncthreads();
        const int cb = wn & 1, half = wn >> 1;
        const char* epg = smem + (wm * WAVES_N + cb) * EPI_WAVE_BYTES;
        const char* epu = smem + (wm * WAVES_N + cb + 2) * EPI_WAVE_BYTES;
        const int64_t g0 = (n0 >> 1) + cb * 64;  // gate column of this 64-column block
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = half * 64 + it * 8 + (lane >> 3), chunk = lane & 7;
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(epg + row * EPI_ROW_BYTES + chunk * 16);
            const bf16x8 uv = *reinterpret_cast<const bf16x8*>(epu + row * EPI_ROW_BYTES + chunk * 16);
            bf16x8 av;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gf = (float)gv[e];
                const float sl = (float)(bf16_t)(gf / (1.f + expf(-gf)));
                av[e] = (bf16_t)(sl * (float)uv[e]);
            }
            const int64_t grow = m0 + wm * WM + row;
            *reinterpret_cast<bf16x8*>(C + grow * ldc + g0 + chunk * 8) = gv;
            *reinterpret_cast<bf16x8*>(C + grow * ldc + ea.inter + g0 + chunk * 8) = uv;
            *reinterpret_cast<bf16x8*>(ea.out2 + grow * ea.ld_out2 + g0 + chunk * 8) = av;
        }
        return;
    }
    if (EPI == EPI_SWIGLU_BWD) {
        // the tile is d act [256 rows, 256 columns of I]: d gate = d act * up * silu'(gate), d up = d act * silu(gate)
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            bf16x8 gv[4], uv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (blk * 4 + q) * 8 + (lane >> 3), chunk = lane & 7;
                const bf16_t* gp = ea.in2 + (m0 + wm * WM + row) * ea.ld_in2 + n0 + wn * WN + chunk * 8;
                gv[q] = *reinterpret_cast<const bf16x8*>(gp);
                uv[q] = *reinterpret_cast<const bf16x8*>(gp + ea.inter);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (blk * 4 + q) * 8 + (lane >> 3), chunk = lane & 7;
                const bf16x8 dv = *reinterpret_cast<const bf16x8*>(ep + row * EPI_ROW_BYTES + chunk * 16);
                bf16x8 og, ou;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gf = (float)gv[q][e], df = (float)dv[e];
                    const float sig = 1.f / (1.f + expf(-gf));
                    ou[e] = (bf16_t)(df * (gf * sig));
                    og[e] = (bf16_t)(df * (float)uv[q][e] * (sig * (1.f + gf * (1.f - sig))));
                }
                bf16_t* op = ea.out2 + (m0 + wm * WM + row) * ea.ld_out2 + n0 + wn * WN + chunk * 8;
                *reinterpret_cast<bf16x8*>(op) = og;
                *reinterpret_cast<bf16x8*>(op + ea.inter) = ou;
            }
        }
        return;
    }
    // each wave reads back only what it wrote: no workgroup barrier needed, the compiler's lgkmcnt wait orders it
#pragma unroll
    for (int it = 0; it < WM / 8; ++it) {
        const int row = it * 8 + (lane >> 3), chunk = lane & 7;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(ep + row * EPI_ROW_BYTES + chunk * 16);
        const int64_t off = (m0 + wm * WM + row) * ldc + n0 + wn * WN + chunk * 8;
        if (accumulate || R) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
            if (accumulate) {
                bf16x8 c = *reinterpret_cast<const bf16x8*>(C + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)c[e];
            }
            if (R) {
                bf16x8 c = *reinterpret_cast<const bf16x8*>(R + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)c[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
        }
        *reinterpret_cast<bf16x8*>(C + off) = v;
    }
}

// C = (accumulate ? C : 0) + bf16(alpha * sum_s slab[s]) (+ R): one 16-B store per thread, same rounding points as the
// direct epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int64_t M, int64_t N,
                                                            bf16_t* __restrict__ C, int64_t ldc, const bf16_t* __restrict__ R,
                                                            float alpha, const float* __restrict__ alpha_dev, int accumulate) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nvec = M * N / 8;
    if (v >= nvec) return;
    const float al = alpha * (alpha_dev ? *alpha_dev : 1.f);
    const int64_t row = (v * 8) / N, col = (v * 8) % N;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float* p = slabs + (int64_t)s * M * N + row * N + col;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] += a[e]; f[4 + e] += b[e]; }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(f[e] * al);
    const int64_t off = row * ldc + col;
    if (accumulate || R) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)o[e];
        if (accumulate) {
            const bf16x8 c = *reinterpret_cast<const bf16x8*>(C + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)c[e];
        }
        if (R) {
            const bf16x8 c = *reinterpret_cast<const bf16x8*>(R + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)c[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)f[e];
    }
    *reinterpret_cast<bf16x8*>(C + off) = o;
}

template <bool A_COL, bool B_COL, bool SPLITK, int EPI = EPI_PLAIN>
int launch(int tiles_m, int tiles_n, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
           int64_t ldc, const void* R, float alpha, const float* alpha_dev, int accumulate, hipStream_t st,
           int splits = 1, float* slabs = nullptr, EpiArgs ea = EpiArgs{nullptr, 0, nullptr, 0, 0}) {
    auto kern = gemm_mfma_kernel<A_COL, B_COL, SPLITK, EPI>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { ssi_set_error("gemm_mfma: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return SSI_ERR_HIP + (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * tiles_n), (unsigned)splits), dim3(NTHREADS), LDS_BYTES, st, tiles_m,
                       tiles_n, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, (const bf16_t*)R, alpha,
                       alpha_dev, accumulate, slabs, ea);
    SSI_LAUNCH_CHECK();
    if (SPLITK) {
        const int64_t M = (int64_t)tiles_m * BM, N = (int64_t)tiles_n * BN;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ssi_cdiv(M * N / 8, 256)), dim3(256), 0, st, slabs, splits, M, N,
                           (bf16_t*)C, ldc, (const bf16_t*)R, alpha, alpha_dev, accumulate);
        SSI_LAUNCH_CHECK();
    }
    return SSI_OK;
}


}  // namespace

bool ssi_gemm_mfma_supported(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                             int64_t ldb, const void* C, int64_t ldc, const void* R) {
    if (M <= 0 || N <= 0 || K <= 0) return false;
    if (M % BM || N % BN || K % BK) return false;
    if (lda % 8 || ldb % 8 || ldc % 8) return false;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)R) & 15) return false;
    if ((M / BM) * (N / BN) > (1LL << 30)) return false;
    (void)layout;
    return true;
}

int ssi_gemm_mfma_bf16(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                       int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                       int accumulate, void* stream) {
    const int tm = (int)(M / BM), tn = (int)(N / BN);
    auto st = (hipStream_t)stream;
#define GO(AC, BC) return launch<AC, BC, false>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, st)
    switch (layout) {
        case SSI_GEMM_NT: GO(false, false);
        case SSI_GEMM_NN: GO(false, true);
        case SSI_GEMM_TN: GO(true, true);
    }
#undef GO
    return SSI_ERR_ARG;
}

int ssi_gemm_mfma_bf16_splitk(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                              int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                              int accumulate, int splits, float* slabs, void* stream) {
    const int tm = (int)(M / BM), tn = (int)(N / BN);
    auto st = (hipStream_t)stream;
#define GO(AC, BC) return launch<AC, BC, true>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, st, splits, slabs)
    switch (layout) {
        case SSI_GEMM_NT: GO(false, false);
        case SSI_GEMM_NN: GO(false, true);
        case SSI_GEMM_TN: GO(true, true);
    }
#undef GO
    return SSI_ERR_ARG;
}

// ---- fused SwiGLU entries (MFMA path only; callers fall back to ssi_gemm + ssi_swiglu_* when this returns UNSUPPORTED) ----
bool ssi_gemm_swiglu_supported(int64_t M, int64_t inter, int64_t K, const void* p0, const void* p1, const void* p2, const void* p3,
                               int64_t ld0, int64_t ld1, int64_t ld2, int64_t ld3) {
    if (M <= 0 || M % BM || inter % BN || K % BK) return false;
    if ((ld0 | ld1 | ld2 | ld3) % 8) return false;
    if (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3) & 15) return false;
    return true;
}

int ssi_gemm_swiglu_fwd_mfma(int64_t M, int64_t inter, int64_t K, const void* X, int64_t ldx, const void* W13, int64_t ldw,
                             void* GU, int64_t ldgu, void* ACT, int64_t ldact, void* stream) {
    EpiArgs ea{(bf16_t*)ACT, ldact, nullptr, 0, inter};
    // output tiles: 256 rows x (128 gate + 128 up) columns -> tiles_n = 2I / 256
    return launch<false, false, false, EPI_SWIGLU_FWD>((int)(M / BM), (int)(2 * inter / BN), K, X, ldx, W13, ldw, GU, ldgu, nullptr, 1.f,
                                                       nullptr, 0, (hipStream_t)stream, 1, nullptr, ea);
}

int ssi_gemm_swiglu_bwd_mfma(int64_t M, int64_t inter, int64_t K, const void* DY, int64_t lddy, const void* W2T, int64_t ldw,
                             const void* GU, int64_t ldgu, void* DGU, int64_t lddgu, void* stream) {
    EpiArgs ea{(bf16_t*)DGU, lddgu, (const bf16_t*)GU, ldgu, inter};
    // d act [M, I] = DY [M, K] * W2T[I, K]^T; the tile never reaches memory
    return launch<false, false, false, EPI_SWIGLU_BWD>((int)(M / BM), (int)(inter / BN), K, DY, lddy, W2T, ldw, DGU, lddgu, nullptr, 1.f,
                                                       nullptr, 0, (hipStream_t)stream, 1, nullptr, ea);
}
