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
#include "common_hip.h"
#include <atomic>
#include <type_traits>

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

// grouped order of the tile sequence: GM m-tiles x all n-tiles per group (second half of tile_coords)
__device__ __forceinline__ void tile_from_t(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int group = t / per_group, in_group = t % per_group;
    const int gm = (tiles_m - group * GM) < GM ? (tiles_m - group * GM) : GM;
    tm = group * GM + in_group % gm;
    tn = in_group / gm;
}

// Tile scheduler state of the persistent kernel: per launch slot, 8 per-XCD "next tile of my span" counters + a count of
// finished workgroups (the last one to finish clears the slot for its next use).  Static device memory, no allocation.
__device__ int g_nt4_sched[16][16];

// SPLITK: blockIdx.y selects a contiguous range of K-tiles; the fp32 partial tile goes to slab[blockIdx.y] (an [M, N] fp32
// matrix in the workspace) and splitk_reduce_kernel applies alpha / accumulate / residual and the bf16 rounding.
// EPI: 0 plain epilogue; 1 SwiGLU forward (the 256 output columns of a tile are 128 gate + the matching 128 up columns of
// W13; writes GU = [gate | up] and ACT = silu(gate) * up); 2 SwiGLU backward (C tile = d act, never stored: reads gate/up
// from GU and writes d gate / d up into DGU).  Same rounding points as the separate swiglu kernels (bf16 GEMM result first).
enum { EPI_PLAIN = 0, EPI_SWIGLU_FWD = 1, EPI_SWIGLU_BWD = 2, EPI_ROPE = 3 };
struct EpiArgs {
    bf16_t* out2;        // FWD: ACT [M, I]      BWD: DGU [M, 2I]
    int64_t ld_out2;
    const bf16_t* in2;   // BWD: GU [M, 2I]
    int64_t ld_in2;
    int64_t inter;       // I
    // EPI_ROPE (QKV projection): interleaved RoPE on output columns [0, rot_cols) (the q and k heads, 64 wide), applied to the
    // bf16-rounded GEMM result and rounded again = ssi_rope_inplace on the stored tensor
    const float* rope = nullptr;    // [table_len, 32, 2] (cos, sin)
    const int32_t* pos = nullptr;   // [M] positions, or NULL: position = row % seq
    int64_t seq = 1;
    int64_t rot_cols = 0;
    // batched form (EPI_PLAIN, no split-K): `batch` problems of one shape in one launch; tile t belongs to problem t / (tiles_m * tiles_n),
    // whose operands start bsA / bsB / bsC elements after those of the problem before (C's stride also applies to R)
    int batch = 1;
    int64_t bsA = 0, bsB = 0, bsC = 0;
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
    // nothing is in flight towards LDS any more (the last K-step's wait above saw to it); said once more where no branch can skip it, so
    // that the listing shows it on every path to the end of the kernel (tools/kernel_lint.py, rule R3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
                const float sl = (float)(bf16_t)ssi_silu<bf16_t>(gf);
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
                    float dg, du;
                    ssi_swiglu_bwd_elem<bf16_t>((float)gv[q][e], (float)uv[q][e], (float)dv[e], dg, du);
                    ou[e] = (bf16_t)du;
                    og[e] = (bf16_t)dg;
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
    // once per instantiation, whichever host thread gets here first (forward and autograd's backward thread both launch GEMMs)
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr_rc != hipSuccess) { ssi_set_error("gemm_mfma: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(attr_rc)); return SSI_ERR_HIP + (int)attr_rc; }
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

// =====================================================================================================================
// NT4: the k-contiguous (NT) form as a persistent kernel with one wave per SIMD — the shape the chip's own library picks
// for these sizes (256 x 256 x 64 tile, 4 waves), written for this step's epilogues.
//   * 256 threads = 4 waves as 2 x 2, 128 x 128 outputs per wave: all 256 AGPRs hold accumulators, LDS fragment traffic is
//     2/3 of the 8-wave kernel's.  The MFMAs are inline asm with the accumulator tied in place: with the whole AGPR file in
//     use the builtin form makes the allocator rotate accumulators through copies (hundreds of v_accvgpr moves per K-tile).
//   * gridDim.x (= CUs) workgroups draw output tiles from a scheduler (per-XCD spans, see below) until none is left; the
//     order in which tiles are handed out is the XCD-aware grouped order.  The global -> VGPR -> LDS operand stream runs three
//     K-steps ahead of the MFMAs through two alternating register sets (every load has two full K-steps to land) and does
//     not stop at tile boundaries: a tile's first operands arrive while the previous tile is still being multiplied.
//   * One K-step = 8 blocks of 16 MFMAs.  Four fragment slots (4 x 16 rows x 32 k each) rotate so that each block fetches
//     just the one set the next block is missing; the ds_write / buffer_load of the staging stream are spread one per
//     ~4 MFMAs (a slot carrying a ds_write_b128 costs ~35 cycles instead of 16, so they must not bunch up); one raw
//     s_barrier per K-step.  Measured: 2390 cycles per K-step against 2048 of pure MFMA issue.
//   * Epilogue straight from registers: two v_permlane16_swap per tile pair give each lane 8 consecutive bf16 columns
//     (16-B stores, 64 B per row per instruction); no LDS, so the operand stream keeps running underneath it.
// Requires K % 128 == 0 and not both `accumulate` and R; other NT calls use the 8-wave kernel above.
// =====================================================================================================================
// cache-policy bits of the epilogue stores (2 = nt, 16 = sc1): measured neutral on the step's shapes, left at the default
#ifndef NT4_ST_AUX
#define NT4_ST_AUX 0
#endif
#ifndef NT4_ST_AUX_BIG  // outputs far larger than the 256 MiB Infinity Cache (gate/up/act of the fused SwiGLU forward, 0.8 GB)
#define NT4_ST_AUX_BIG 2
#endif
// schedule of the LDS-DMA loop (slots = MFMA indices inside a 64-MFMA phase): F1 reads every NT4_RS MFMAs from slot 0, barrier at NT4_BAR,
// NT4_NA pieces in phase A at NT4_A0 + i * NT4_SA, the other pieces in phase B at 3 + 5 i
#ifndef NT4_RS
#define NT4_RS 2
#define NT4_BAR 40
#define NT4_A0 44
#define NT4_SA 6
#define NT4_NA 4
#endif
constexpr int NT4_THREADS = 256;
constexpr int NT4_LDS_BYTES = PIPE_BYTES + 16;  // operand pipeline + the scheduler's broadcast word
std::atomic<int> g_nt4_dynamic{0};
inline unsigned nt4_next_slot() {
    static std::atomic<unsigned> n{0};
    return n.fetch_add(1, std::memory_order_relaxed);
}
constexpr int NT4_WM = 128, NT4_WN = 128;
constexpr unsigned BUF_RSRC_DW3 = 0x00020000u;  // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, BUF_RSRC_DW3);
}

// A_COL / B_COL: false = the operand is k-contiguous in memory, true = k-strided (its tile is a [64 k][256 columns] image read
//      back with ds_read_b64_tr_b16, as in the 8-wave kernel).  NT = (false, false), NN = (false, true: data gradients against
//      the untransposed weights), TN = (true, true: weight gradients)
// PREV: 0 = C is overwritten, 1 = C += result (accumulate), 2 = C = R + result (residual)
// SPLITK: a unit of work is (output tile, K-slice); the fp32 partial tile goes to the workspace in the accumulator's own layout
//      (unit, wave, 16x16 tile, lane: every store is one contiguous KiB) and nt4_splitk_reduce_kernel finishes the tile
// DMA (k-contiguous operands only): the operand stream goes HBM -> LDS by LDS-DMA (buffer_load ... lds, no staging registers, no
//      ds_write) two K-steps ahead, and the registers hold a whole K-step of fragments instead (see the DMA main loop below)
template <bool A_COL, bool B_COL, int EPI, int PREV, bool SPLITK, bool DMA, bool BATCHED = false>
__device__ __forceinline__ void nt4_body(char* smem, int tiles_m, int tiles_n, int64_t K, const bf16_t* __restrict__ A,
                                         int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                         bf16_t* __restrict__ C, int64_t ldc, const bf16_t* __restrict__ R,
                                         float alpha, const float* __restrict__ alpha_dev, EpiArgs ea, int slot,  // slot < 0: static tile order
                                         int splits, float* __restrict__ slabs) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int per_prob = tiles_m * tiles_n;        // output tiles of one problem
    const int ntiles_out = BATCHED ? per_prob * ea.batch : per_prob;  // output tiles of the launch (BATCHED: ssi_gemm_batched)
    const int ntiles = SPLITK ? ntiles_out * splits : ntiles_out;  // units of work
    // unit -> (problem, tile coordinates inside it)
    auto locate = [&](int t, int& tm, int& tn) {
        int tt = SPLITK ? t % ntiles_out : t, pb = 0;
        if constexpr (BATCHED) { pb = tt / per_prob; tt -= pb * per_prob; }
        tile_from_t(tt, tiles_m, tiles_n, tm, tn);
        return pb;
    };
    const int G = (int)gridDim.x;
    const int nk_total = (int)(K / BK);  // even, >= 4 (per K-slice when SPLITK)
    // K-steps [k_lo, k_hi) of unit t (slice boundaries rounded to even step counts)
    auto k_lo = [&](int t) { return SPLITK ? (int)(((int64_t)nk_total * (t / ntiles_out) / splits) & ~1LL) : 0; };
    auto k_hi = [&](int t) { return SPLITK ? ((t / ntiles_out) + 1 == splits ? nk_total : (int)(((int64_t)nk_total * (t / ntiles_out + 1) / splits) & ~1LL)) : nk_total; };

    // ---- tile scheduler: workgroups draw tiles from the contiguous span of their XCD (operand panels stay in that XCD's L2) and,
    // once it is empty, from the other spans.  Dynamic rather than "tile b, b+G, ...": a workgroup that starts late, or not at all,
    // because another kernel (RCCL during the gradient exchange) holds its CU, then costs its share of tiles, not a whole round.
    // The first tile of a workgroup is fixed (index b/8 of its XCD's span), later ones come from the span's atomic counter; the
    // request for the tile after next is issued at the start of a tile and read at its end, so its round trip is never waited for.
    int* sched = g_nt4_sched[slot < 0 ? 0 : slot];
    volatile int* lds_next = reinterpret_cast<volatile int*>(smem + PIPE_BYTES);
    const int xcd = (int)blockIdx.x & 7, span_q = ntiles >> 3, span_r = ntiles & 7;
    auto span_len = [&](int x) { return span_q + (x < span_r ? 1 : 0); };
    auto span_lo = [&](int x) { return x < span_r ? x * (span_q + 1) : span_r * (span_q + 1) + (x - span_r) * span_q; };
    // the first two tiles of every workgroup are fixed (no atomic in the prologue; shapes of up to two rounds never use the counters)
    auto span_wgs = [&](int x) { return (G - x + 7) >> 3; };  // workgroups whose blockIdx maps to XCD x
    auto span_fixed = [&](int x) { const int n = 2 * span_wgs(x); return n < span_len(x) ? n : span_len(x); };
    // slot < 0 = static order (tile index += G/8 within the span): optimal when every workgroup has its CU from the start, since
    // equal tiles then need no balancing and a greedy scheduler can only hurt (a workgroup that is a little early takes a third
    // tile where everybody should do two: measured -3..-12 % on the 2-3-round shapes).  The trainer switches to the dynamic
    // order when the gradient exchange shares the GPU with the backward GEMMs (world size > 1).
    const bool dynamic = slot >= 0;
    int my_idx = (int)blockIdx.x >> 3;  // static order: position inside the span
    int pending = 0;
    auto request_tile = [&]() {
        if (dynamic && tid == 0) pending = atomicAdd(&sched[xcd], 1);
    };
    auto receive_tile = [&]() -> int {
        if (!dynamic) {
            my_idx += G >> 3;
            return ((G & 7) == 0 && my_idx < span_len(xcd)) ? span_lo(xcd) + my_idx : -1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everybody has read the previous answer
        if (tid == 0) {
            int t = -1;
            const int i = pending + span_fixed(xcd);
            if (i < span_len(xcd)) t = span_lo(xcd) + i;
            if (t < 0) {  // own span empty: take from the others (only at the very end of a launch); one look at all the counters
                int seen[8];  // first, so that a finished launch costs one load instead of seven atomic round trips
#pragma unroll
                for (int x = 0; x < 8; ++x) seen[x] = __hip_atomic_load(&sched[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int k = 1; k < 8 && t < 0; ++k) {
                    const int x = (xcd + k) & 7;
                    if (seen[x] + span_fixed(x) >= span_len(x)) continue;
                    const int j = atomicAdd(&sched[x], 1) + span_fixed(x);
                    if (j < span_len(x)) t = span_lo(x) + j;
                }
            }
            *lds_next = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return __builtin_amdgcn_readfirstlane(*lds_next);
    };
    int cur = ((int)blockIdx.x >> 3) < span_len(xcd) ? span_lo(xcd) + ((int)blockIdx.x >> 3) : -1;
    int nxt = -1;
    if (dynamic) {
        const int second = ((int)blockIdx.x >> 3) + span_wgs(xcd);
        nxt = (cur >= 0 && second < span_len(xcd)) ? span_lo(xcd) + second : -1;
    }

    f32x4 acc[8][8];  // [j (n-tile)][i (m-tile)], defined by the zero-C MFMAs of each tile's first K-step
    // LDS: [A buf 0 | A buf 1 | B buf 0 | B buf 1]; with the buffer index a compile-time constant every fragment address is one
    // per-lane base register plus a 16-bit immediate
    auto tileA = [&](int buf) { return smem + buf * TILE_BYTES; };
    auto tileB = [&](int buf) { return smem + 2 * TILE_BYTES + buf * TILE_BYTES; };

    // ---- load side: (lv, lkt) = output tile and K-tile of the next fetch ------------------------------------------------
    int lkt = 0, lnk = nk_total;  // K-step of the next fetch inside its unit, K-steps of that unit
    const bf16_t* baseA = A;
    const bf16_t* baseB = B;
    const int64_t kstepA_ = A_COL ? BK * lda : BK, kstepB_ = B_COL ? BK * ldb : BK;
    auto set_load_tile = [&](int t) {
        int tm, tn;
        const int pb = locate(t, tm, tn);
        const bf16_t* Ap = BATCHED ? A + pb * ea.bsA : A;
        const bf16_t* Bp = BATCHED ? B + pb * ea.bsB : B;
        baseA = (A_COL ? Ap + (int64_t)tm * BM : Ap + (int64_t)tm * BM * lda) + k_lo(t) * kstepA_;
        // SwiGLU forward: the 256 tile columns are [gate 0..63 | up 0..63 | gate 64..127 | up 64..127] of 128 W13 column pairs
        baseB = (B_COL ? Bp + (int64_t)tn * BN : Bp + (int64_t)tn * (EPI == EPI_SWIGLU_FWD ? BN / 2 : BN) * ldb) + k_lo(t) * kstepB_;
        lnk = k_hi(t) - k_lo(t);
    };
    auto advance = [&]() {
        if (++lkt == lnk) {  // the fetches run at most 3 K-steps ahead of the MFMAs: this is always the switch to tile `nxt`
            lkt = 0;
            if (nxt >= 0) set_load_tile(nxt);  // after the last tile: keep re-fetching valid memory, never consumed
        }
    };
    // staging, ROW tiles: piece p = rows p*32 + (tid >> 3), 16-B chunk (tid & 7), swizzled on the LDS side.  COL tiles: piece p =
    // k-rows p*8 + (tid >> 5), chunk (tid & 31) swizzled on the SOURCE side (LDS image lane-linear); the swizzle term
    // col_swz(k-row) only depends on p through its parity.  One VGPR byte offset per operand (two for COL); the (piece,
    // K-tile) part rides in the scalar offset of the buffer load.
    auto lane_off = [&](bool col, int64_t ld, int odd) {
        if (col) return (int)(((tid >> 5) * ld + (((tid & 31) ^ col_swz(odd * 8 + (tid >> 5))) * 8)) * 2);
        return (int)(((tid >> 3) * ld + (tid & 7) * 8) * 2);
    };
    const int offA0 = lane_off(A_COL, lda, 0), offA1 = A_COL ? lane_off(true, lda, 1) : offA0;
    const int offB0 = lane_off(B_COL, ldb, 0), offB1 = B_COL ? lane_off(true, ldb, 1) : offB0;
    const int64_t kstepA = kstepA_, kstepB = kstepB_;  // elements per K-step: added to the (64-bit) buffer base
    auto pieceA = [&](int p) { return A_COL ? (int)(p * 8 * lda * 2) : (int)(p * 32 * lda * 2); };
    auto pieceB = [&](int p) {
        if (B_COL) return (int)(p * 8 * ldb * 2);
        if (EPI == EPI_SWIGLU_FWD) {
            const int blk64 = p >> 1;  // 64-row block of the tile: 0 gate lo, 1 up lo, 2 gate hi, 3 up hi
            return (int)((((blk64 & 1) ? ea.inter : 0) + (blk64 >> 1) * 64 + (p & 1) * 32) * ldb * 2);
        }
        return (int)(p * 32 * ldb * 2);
    };
    const int st_row = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) * 16);
    const int st_ofsA = A_COL ? tid * 16 : st_row, st_ofsB = B_COL ? tid * 16 : st_row;
    auto gloadA = [&](u32x4& dst, int p) { dst = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(baseA + lkt * kstepA), (p & 1) ? offA1 : offA0, pieceA(p), 0); };
    auto gloadB = [&](u32x4& dst, int p) { dst = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(baseB + lkt * kstepB), (p & 1) ? offB1 : offB0, pieceB(p), 0); };
    auto lwriteA = [&](char* tile, int p, const u32x4& v) { *reinterpret_cast<u32x4*>(tile + st_ofsA + p * 4096) = v; };
    auto lwriteB = [&](char* tile, int p, const u32x4& v) { *reinterpret_cast<u32x4*>(tile + st_ofsB + p * 4096) = v; };

    bf16x8 S0[4], S1[4], S2[4], S3[4];  // fragment slots; roles rotate through one K-step (see body)
    // COL fragments: the address of 16-column group t (0..7) of a wave's operand is (group-0 address) XOR (t << 5) — the
    // swizzle only touches the three chunk bits that t occupies — so one per-lane base per operand serves all 8 groups; the
    // XOR is redone per read (kept from being hoisted into 16 live registers by the asm in body).
    int trA = 0, trB = 0;
    if (A_COL || B_COL) {
        const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, k0l = 8 * (lane >> 4) + q;
        const int common = k0l * 512 + (pp >> 1) * 16 + 8 * (pp & 1) + (col_swz(k0l) << 4);
        trA = common + wm * 256;
        trB = common + wn * 256;
    }
    auto rd_tr = [&](const char* tile, int base, int t8, int kh) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const char* a0 = tile + (base ^ (t8 << 5)) + kh * 16384;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 2048));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    };
    auto rdA1 = [&](bf16x8& dst, const char* la, int half, int kh, int i) {
        if (A_COL) dst = rd_tr(la, trA, half * 4 + i, kh);
        else dst = read_frag<false>(la, wm * NT4_WM + (half * 4 + i) * 16, kh, lane);
    };
    auto rdB1 = [&](bf16x8& dst, const char* lb, int half, int kh, int j) {
        if (B_COL) dst = rd_tr(lb, trB, half * 4 + j, kh);
        else dst = read_frag<false>(lb, wn * NT4_WN + (half * 4 + j) * 16, kh, lane);
    };
    // one block = 16 MFMAs (4 m-tiles of slot a x 4 n-tiles of slot b); extra(q) is issued right after MFMA q and pinned there.
    // zero_c: the block is the first to touch its 16 accumulator tiles in this output tile (C operand = 0, no zero fill).
    auto blk = [&](const bf16x8 (&a)[4], int ah, const bf16x8 (&b)[4], int bh, auto zero_c, auto extra) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = q >> 2, j = q & 3;
            if (decltype(zero_c)::value)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[bh * 4 + j][ah * 4 + i]) : "v"(b[j]), "v"(a[i]));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[bh * 4 + j][ah * 4 + i]) : "v"(b[j]), "v"(a[i]));
            extra(q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using F_ = std::false_type;
    // entry: S0 = A rows-lo k-lo, S1 = B cols-lo k-lo of this K-step (LDS buffer cur = K-step parity); (xa, xb) hold the next K-step
    auto body = [&](auto cur_c, u32x4 (&xa)[8], u32x4 (&xb)[8], auto first) {
        constexpr int cur = decltype(cur_c)::value;
        if (A_COL) asm volatile("" : "+v"(trA));
        if (B_COL) asm volatile("" : "+v"(trB));
        const char* la = tileA(cur);
        const char* lb = tileB(cur);
        char* na = tileA(cur ^ 1);
        char* nb = tileB(cur ^ 1);
        // staging piece pc (0..7 = A, 8..15 = B) of the next K-step: registers -> idle LDS buffer in block pc/3, slot
        // 4 + 4*(pc%3); its register is refilled (K-step +3) two slots later
        auto stg = [&](int b, int q) {
            if (q < 4 || (q & 1)) return;
            const int pc = b * 3 + ((q - 4) >> 2);
            if (pc >= 16) return;
            if ((q & 3) == 0) {
                if (pc < 8) lwriteA(na, pc, xa[pc]); else lwriteB(nb, pc - 8, xb[pc - 8]);
            } else {
                if (pc < 8) gloadA(xa[pc], pc); else gloadB(xb[pc - 8], pc - 8);
            }
        };
        blk(S0, 0, S1, 0, first, [&](int q) { if (q < 4) rdB1(S2[q], lb, 1, 0, q); else stg(0, q); });
        blk(S0, 0, S2, 1, first, [&](int q) { if (q < 4) rdA1(S3[q], la, 1, 0, q); else stg(1, q); });
        blk(S3, 1, S2, 1, first, [&](int q) { if (q < 4) rdA1(S0[q], la, 0, 1, q); else stg(2, q); });
        blk(S3, 1, S1, 0, first, [&](int q) { if (q < 4) rdB1(S2[q], lb, 0, 1, q); else stg(3, q); });
        blk(S0, 0, S2, 0, F_{}, [&](int q) { if (q < 4) rdB1(S1[q], lb, 1, 1, q); else stg(4, q); });
        blk(S0, 0, S1, 1, F_{}, [&](int q) { if (q < 4) rdA1(S3[q], la, 1, 1, q); else stg(5, q); });
        advance();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // next K-step visible; every wave has read all of this one
        blk(S3, 1, S1, 1, F_{}, [&](int q) { if (q < 4) rdA1(S0[q], na, 0, 0, q); });
        blk(S3, 1, S2, 0, F_{}, [&](int q) { if (q < 4) rdB1(S1[q], nb, 0, 0, q); });
    };
    u32x4 ra0[8], rb0[8], ra1[8], rb1[8];
    // ---- DMA main loop (NT form) ---------------------------------------------------------------------------------------------
    // LDS: the same two [A | B] buffers and the same swizzled row image as the register-staged path, but filled by LDS-DMA: a
    // piece = one wave-instruction = 8 rows x 128 B, lane-linear in LDS, the 16-byte-chunk swizzle applied to the per-lane SOURCE
    // address.  Per K-step and wave: 128 MFMAs, 32 ds_read_b128, 16 DMA issues, two barriers; nothing else touches a VGPR.
    //   registers : F0 = the 8 A + 8 B fragments of the k-half 0 of this K-step, F1 = those of k-half 1 (128 VGPRs)
    //   phase A   : 64 MFMAs on F0.  Under them F1 is read from this K-step's LDS buffer; once every wave has done so (barrier) the
    //               buffer is free and the DMA of K-step +2 starts into it
    //   phase B   : 64 MFMAs on F1.  The DMA of K-step +1 (issued one K-step ago) is waited for (counted vmcnt: this K-step's 16
    //               pieces stay in flight) + barrier, then F0 of the next K-step is read from the other buffer
    bf16x8 F0A[8], F0B[8], F1A[8], F1B[8];
    const int dofA = (int)(((tid >> 3) * lda + (((tid & 7) ^ ((tid >> 4) & 7)) * 8)) * 2);
    const int dofB = (int)(((tid >> 3) * ldb + (((tid & 7) ^ ((tid >> 4) & 7)) * 8)) * 2);
    // One piece = buffer_load_dwordx4 ... lds: LDS destination = M0 + lane * 16.  M0 is written one MFMA AHEAD of the load (dma_m0 then
    // dma_go): written right in front of it, as the builtin form does, every piece stalls the wave's issue for ~50 cycles (measured:
    // SQ_WAIT_INST_ANY +25 %, the loop 18 % longer); with an MFMA in between the wait disappears under the matrix pipe.
    typedef __attribute__((address_space(3))) char lds_c;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + (unsigned)wave * 1024u);  // one cast, then integers
    auto rsrc_words = [&](const bf16_t* base) {  // raw buffer resource: base, stride 0, 2 GiB window, 32-bit data format
        const uint64_t a = (uint64_t)(uintptr_t)base;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
        r[2] = 0x7fffffffu;
        r[3] = BUF_RSRC_DW3;
        return r;
    };
    auto dma_m0 = [&](int buf, int d) {  // piece d: 0..7 = A rows d*32 .. +31, 8..15 = B
        const unsigned dst = lds_wave + (unsigned)((d < 8 ? buf * TILE_BYTES : 2 * TILE_BYTES + buf * TILE_BYTES) + (d & 7) * 4096);  // tileA / tileB
        asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
    };
    u32x4 rsA, rsB;  // buffer resources of the K-step being fetched (set once per K-step by dma_src)
    auto dma_src = [&]() { rsA = rsrc_words(baseA + lkt * kstepA); rsB = rsrc_words(baseB + lkt * kstepB); };
    // k-contiguous operand: 8 rows x 128 B per piece, swizzled source chunk (dofA / dofB); k-strided operand: 2 k-rows x 512 B per piece with
    // the register path's own source offsets (its LDS image is lane-linear already)
    auto dma_go = [&](int d) {
        if (d < 8) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(A_COL ? ((d & 1) ? offA1 : offA0) : dofA), "s"(rsA), "s"(pieceA(d)) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(B_COL ? ((d & 1) ? offB1 : offB0) : dofB), "s"(rsB), "s"(pieceB(d - 8)) : "memory");
    };
    auto dma = [&](int buf, int d) {  // prologue form: no MFMA to hide behind
        dma_m0(buf, d);
        asm volatile("s_nop 1" ::: "memory");
        dma_go(d);
    };
    auto fragA = [&](const char* t, int i, int kh) {
        if constexpr (A_COL) return rd_tr(t, trA, i, kh);
        else return read_frag<false>(t, wm * NT4_WM + i * 16, kh, lane);
    };
    auto fragB = [&](const char* t, int j, int kh) {
        if constexpr (B_COL) return rd_tr(t, trB, j, kh);
        else return read_frag<false>(t, wn * NT4_WN + j * 16, kh, lane);
    };
    // 64 MFMAs: every accumulator tile once, A fragment outer; extra(m) is issued right after MFMA m and pinned there
    auto phase = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[8], auto zero_c, auto extra) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int i = m >> 3, j = m & 7;
            if (decltype(zero_c)::value)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[j][i]) : "v"(fb[j]), "v"(fa[i]));
            else
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(fb[j]), "v"(fa[i]));
            extra(m);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto dma_body = [&](auto cur_c, auto first) {
        constexpr int cur = decltype(cur_c)::value;
        if (A_COL) asm volatile("" : "+v"(trA));  // keep the per-read XOR of the transposed-read addresses from being hoisted into 16 registers
        if (B_COL) asm volatile("" : "+v"(trB));
        const char* la = tileA(cur);
        const char* lb = tileB(cur);
        const char* na = tileA(cur ^ 1);
        const char* nb = tileB(cur ^ 1);
        // DMA pieces are spread over the rest of the K-step (4 waves x 1 KiB every ~5 MFMAs keeps the CU's load path about half busy; bunched
        // right behind the barrier the pieces queue up and every issue stalls the wave): pieces 0..NA-1 in phase A, the rest in phase B
        constexpr int RS = NT4_RS, BAR = NT4_BAR, A0 = NT4_A0, SA = NT4_SA, NA = NT4_NA;
        static_assert(15 * RS + 4 <= BAR && BAR + 2 <= A0 - 1 && A0 + (NA - 1) * SA <= 63 && 3 + 5 * (15 - NA) <= 58, "DMA schedule");
        phase(F0A, F0B, first, [&](int m) {
            if (m < 16 * RS && m % RS == 0) {
                const int r = m / RS;
                if (r < 8) F1B[r] = fragB(lb, r, 1); else F1A[r - 8] = fragA(la, r - 8, 1);
            }
            if (m == BAR) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave holds its F1: this buffer may be overwritten
                dma_src();
            }
            if (m >= A0 - 1 && m < A0 - 1 + NA * SA && (m - (A0 - 1)) % SA == 0) dma_m0(cur, (m - (A0 - 1)) / SA);
            if (m >= A0 && m < A0 + NA * SA && (m - A0) % SA == 0) dma_go((m - A0) / SA);
        });
        // the fetch cursor moves on BETWEEN the phases (this K-step's buffer resources were taken at the barrier above).  Not inside the
        // unrolled MFMA loops: with the split-K unit arithmetic (integer divisions) inlined there, the 64-iteration loop passes hipcc's
        // size limit for `#pragma unroll`, stays a runtime loop, and the accumulator array it then indexes dynamically lands in scratch
        advance();
        phase(F1A, F1B, F_{}, [&](int m) {
            if (m >= 2 && m < 2 + 5 * (16 - NA) && (m - 2) % 5 == 0) dma_m0(cur, NA + (m - 2) / 5);  // 2, 7, ...
            if (m >= 3 && m < 3 + 5 * (16 - NA) && (m - 3) % 5 == 0) dma_go(NA + (m - 3) / 5);       // 3, 8, ...
            if (m == 11) {
                // K-step +1 (16 pieces issued during the last K-step) has landed for this wave — the NA + 2 pieces of this K-step issued so
                // far may still be in flight — and, with the barrier, for every wave
                if constexpr (NA == 4) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                else if constexpr (NA == 6) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                else if constexpr (NA == 7) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // next K-step's F0 from the other buffer: 16 reads on the even slots 12..50 that carry no DMA issue (18, 28, 38, 48 do)
            if (m >= 12 && m <= 50 && !(m & 1) && m % 10 != 8) {
                const int r = (m - 12) / 2 - (m - 8) / 10;
                if (r < 8) F0B[r] = fragB(nb, r, 0); else F0A[r - 8] = fragA(na, r - 8, 0);
            }
        });
    };
    if constexpr (DMA) {
        set_load_tile(cur >= 0 ? cur : 0);
        if (nk_total == 0) return;
        dma_src();
#pragma unroll
        for (int d = 0; d < 16; ++d) dma(0, d);
        advance();
        dma_src();
#pragma unroll
        for (int d = 0; d < 16; ++d) dma(1, d);
        advance();
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");  // K-step 0 is in LDS
#pragma unroll
        for (int r = 0; r < 8; ++r) F0B[r] = fragB(tileB(0), r, 0);
#pragma unroll
        for (int r = 0; r < 8; ++r) F0A[r] = fragA(tileA(0), r, 0);
        if (cur >= 0 && !dynamic) nxt = receive_tile();
    } else {
    auto fetch_step = [&](u32x4 (&xa)[8], u32x4 (&xb)[8]) {
#pragma unroll
        for (int p = 0; p < 8; ++p) gloadA(xa[p], p);
#pragma unroll
        for (int p = 0; p < 8; ++p) gloadB(xb[p], p);
        advance();
    };
    set_load_tile(cur >= 0 ? cur : 0);
    if (nk_total == 0) return;  // (never: keeps the scheduler state below out of a degenerate launch)
    fetch_step(ra0, rb0);
#pragma unroll
    for (int p = 0; p < 8; ++p) lwriteA(tileA(0), p, ra0[p]);
#pragma unroll
    for (int p = 0; p < 8; ++p) lwriteB(tileB(0), p, rb0[p]);
    fetch_step(ra1, rb1);
    fetch_step(ra0, rb0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) { rdA1(S0[q], tileA(0), 0, 0, q); rdB1(S1[q], tileB(0), 0, 0, q); }
    if (cur >= 0 && !dynamic) nxt = receive_tile();

    }
    const float al = alpha * (alpha_dev ? *alpha_dev : 1.f);
    const int g = lane >> 4;
    while (cur >= 0) {
        int tm, tn;
        const int pb = locate(cur, tm, tn);
        const int nk = k_hi(cur) - k_lo(cur);
        using B0_ = std::integral_constant<int, 0>;
        using B1_ = std::integral_constant<int, 1>;
        // the tile after next is requested now and read after the epilogue: the answer is the oldest vector-memory operation in
        // flight, so the counted waits of the K-steps below retire it (asking later would mean waiting for the epilogue's stores)
        if (nxt >= 0) request_tile();
        if constexpr (DMA) {
            dma_body(B0_{}, std::true_type{});
            dma_body(B1_{}, F_{});
            for (int kt = 2; kt < nk; kt += 2) {
                dma_body(B0_{}, F_{});
                dma_body(B1_{}, F_{});
            }
        } else {
            body(B0_{}, ra1, rb1, std::true_type{});
            body(B1_{}, ra0, rb0, F_{});
            for (int kt = 2; kt < nk; kt += 2) {
                body(B0_{}, ra1, rb1, F_{});
                body(B1_{}, ra0, rb0, F_{});
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the asm MFMAs are opaque to the hazard recogniser: let the last results land
        // ---- epilogue: acc[j][i] holds C[m = wm*128 + i*16 + (lane&15)][n = wn*128 + j*16 + (lane>>4)*4 + r] -------------
        // pair(i, ja, jb): tiles ja, jb of m-tile i -> after the swaps lane (r, g) owns 8 consecutive columns of ONE of them:
        // even 16-lane groups tile ja, odd groups tile jb, columns (g >> 1) * 8 .. + 7 of that tile
        auto pair = [&](int i, int ja, int jb, auto scale_c) {
            bf16x4 x, y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float fx, fy;  // explicit reads, in program order: left to the allocator these become long-range AGPR shuffles
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(fx) : "a"(acc[ja][i][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(fy) : "a"(acc[jb][i][r]));
                x[r] = (bf16_t)(decltype(scale_c)::value ? fx * al : fx);
                y[r] = (bf16_t)(decltype(scale_c)::value ? fy * al : fy);
            }
            const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
            const auto s0 = __builtin_amdgcn_permlane16_swap(xu[0], yu[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(xu[1], yu[1], false, false);
            u32x4 o;
            o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
            return o;
        };
        // duo(i, j0): the four tiles j0 .. j0+3 (64 columns = one 128-B line per row) of m-tile i as two 16-B vectors per lane
        // arranged for FULL-LINE stores: a store whose 16-lane groups each cover 64 B of 16 different rows costs ~7x a store of
        // whole 128-B lines (measured), so lanes 8..15 of every group trade rows with lanes 0..7 (two DPP row shifts):
        //   lo: rows 0..7  — lanes r < 8 keep (row r, columns 0..31),  lanes r >= 8 take (row r-8, columns 32..63)
        //   hi: rows 8..15 — lanes r < 8 take (row r+8, columns 0..31), lanes r >= 8 keep (row r, columns 32..63)
        auto duo = [&](int i, int j0, auto scale_c, u32x4& lo, u32x4& hi) {
            const u32x4 oa = pair(i, j0, j0 + 1, scale_c), ob = pair(i, j0 + 2, j0 + 3, scale_c);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                lo[d] = (unsigned)__builtin_amdgcn_update_dpp((int)oa[d], (int)ob[d], 0x118 /* row_shr:8 */, 0xF, 0xC, false);
                hi[d] = (unsigned)__builtin_amdgcn_update_dpp((int)ob[d], (int)oa[d], 0x108 /* row_shl:8 */, 0xF, 0x3, false);
            }
        };
        if constexpr (SPLITK) {
            // fp32 partial tile, accumulator layout: [unit][wave][i*8 + j][lane] x f32x4, one contiguous KiB per store
            const __amdgpu_buffer_rsrc_t rsS = make_rsrc(slabs + ((int64_t)cur * 4 + wave) * (64 * 64 * 4));
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float f;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f) : "a"(acc[j][i][r]));
                        v[r] = __builtin_bit_cast(unsigned, f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsS, lane * 16, (i * 8 + j) * 1024, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
        const int64_t row0 = (int64_t)tm * BM + wm * NT4_WM;
        // lane -> (row within an 8-row group, column within a 64-column duo)
        const int lane_row = lane & 7, lane_col = ((lane >> 3) & 1) * 32 + (g & 1) * 16 + (g >> 1) * 8;
        auto epilogue = [&](auto scale_c) {
        if constexpr (EPI == EPI_PLAIN) {
            const int64_t tile_off = (BATCHED ? pb * ea.bsC : 0) + row0 * ldc + (int64_t)tn * BN + wn * NT4_WN;
            const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C + tile_off);
            const __amdgpu_buffer_rsrc_t rsP = make_rsrc(PREV == 2 ? R + tile_off : C + tile_off);
            const int voff = (int)((lane_row * ldc + lane_col) * 2);
            auto soff = [&](int i, int h, int jd) { return (int)(((i * 16 + h * 8) * ldc + jd * 64) * 2); };
            u32x4 pv[2][4];  // [m-tile parity][(duo, half)]
            auto ldprev = [&](int i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[i & 1][q] = __builtin_amdgcn_raw_buffer_load_b128(rsP, voff, soff(i, q & 1, q >> 1), 0);
            };
            auto add_prev = [&](u32x4& o, const u32x4& p) {
                bf16x8 ob = __builtin_bit_cast(bf16x8, o);
                const bf16x8 c = __builtin_bit_cast(bf16x8, p);
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (bf16_t)((float)ob[e] + (float)c[e]);
                o = __builtin_bit_cast(u32x4, ob);
            };
            if (PREV) ldprev(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PREV && i + 1 < 8) ldprev(i + 1);
#pragma unroll
                for (int jd = 0; jd < 2; ++jd) {
                    u32x4 lo, hi;
                    duo(i, jd * 4, scale_c, lo, hi);
                    if (PREV) { add_prev(lo, pv[i & 1][jd * 2]); add_prev(hi, pv[i & 1][jd * 2 + 1]); }
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rsC, voff, soff(i, 0, jd), NT4_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rsC, voff, soff(i, 1, jd), NT4_ST_AUX);
                    __builtin_amdgcn_sched_barrier(0);  // one duo at a time: bounded register pressure
                }
            }
        } else if constexpr (EPI == EPI_ROPE) {
            const int64_t tile_off = row0 * ldc + (int64_t)tn * BN + wn * NT4_WN;
            const __amdgpu_buffer_rsrc_t rsC = make_rsrc(C + tile_off);
            const int voff = (int)((lane_row * ldc + lane_col) * 2);
            auto soff = [&](int i, int hh, int jd) { return (int)(((i * 16 + hh * 8) * ldc + jd * 64) * 2); };
            const bool rot = (int64_t)tn * BN < ea.rot_cols;  // tile-uniform: the v heads are not rotated (rot_cols is a multiple of 256)
            // a lane's 8 columns start at a multiple of 8 inside a 64-wide head, the same one for every vector it stores: the
            // (cos, sin) it needs depend on the row only.  Positions of its 16 rows first, then the table rows one m-tile ahead.
            int prow[8][2];
            f32x4 cs[2][2][2];  // [m-tile parity][row half][pairs 0-1 | pairs 2-3]
            if (rot) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int64_t r = row0 + i * 16 + hh * 8 + lane_row;
                        prow[i][hh] = ea.pos ? ea.pos[r] : (int)(r % ea.seq);
                    }
            }
            auto ldcs = [&](int i) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float* tb = ea.rope + (int64_t)prow[i][hh] * 64 + (lane_col & 63);
                    cs[i & 1][hh][0] = *reinterpret_cast<const f32x4*>(tb);
                    cs[i & 1][hh][1] = *reinterpret_cast<const f32x4*>(tb + 4);
                }
            };
            auto rotate = [&](u32x4& o, const f32x4& c01, const f32x4& c23) {
                bf16x8 v = __builtin_bit_cast(bf16x8, o);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4& c = p < 2 ? c01 : c23;
                    float o0, o1;
                    ssi_rope_pair((float)v[2 * p], (float)v[2 * p + 1], c[(p & 1) * 2], c[(p & 1) * 2 + 1], o0, o1);
                    v[2 * p] = (bf16_t)o0;
                    v[2 * p + 1] = (bf16_t)o1;
                }
                o = __builtin_bit_cast(u32x4, v);
            };
            if (rot) ldcs(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (rot && i + 1 < 8) ldcs(i + 1);
#pragma unroll
                for (int jd = 0; jd < 2; ++jd) {
                    u32x4 lo, hi;
                    duo(i, jd * 4, scale_c, lo, hi);
                    if (rot) { rotate(lo, cs[i & 1][0][0], cs[i & 1][0][1]); rotate(hi, cs[i & 1][1][0], cs[i & 1][1][1]); }
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rsC, voff, soff(i, 0, jd), NT4_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rsC, voff, soff(i, 1, jd), NT4_ST_AUX);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU_FWD) {
            // n-tiles 0..3 = gate columns c0 .. c0+63, n-tiles 4..7 = the matching up columns, c0 = tn*128 + wn*64:
            // GU = [gate | up], ACT = silu(gate).to(bf16) * up   (same rounding points as ssi_swiglu_fwd)
            const int64_t c0 = (int64_t)tn * (BN / 2) + wn * 64;
            const __amdgpu_buffer_rsrc_t rsGU = make_rsrc(C + row0 * ldc + c0);
            const __amdgpu_buffer_rsrc_t rsACT = make_rsrc(ea.out2 + row0 * ea.ld_out2 + c0);
            const int voff = (int)((lane_row * ldc + lane_col) * 2), voff2 = (int)((lane_row * ea.ld_out2 + lane_col) * 2);
            const int up_off = (int)(ea.inter * 2);
            auto act_of = [&](const u32x4& gq, const u32x4& uq) {
                const bf16x8 gv = __builtin_bit_cast(bf16x8, gq), uv = __builtin_bit_cast(bf16x8, uq);
                bf16x8 av;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sl = (float)(bf16_t)ssi_silu<bf16_t>((float)gv[e]);
                    av[e] = (bf16_t)(sl * (float)uv[e]);
                }
                return __builtin_bit_cast(u32x4, av);
            };
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u32x4 glo, ghi, ulo, uhi;
                duo(i, 0, scale_c, glo, ghi);
                duo(i, 4, scale_c, ulo, uhi);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4& gq = h ? ghi : glo;
                    const u32x4& uq = h ? uhi : ulo;
                    const int so = (int)(((i * 16 + h * 8) * ldc) * 2), so2 = (int)(((i * 16 + h * 8) * ea.ld_out2) * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(gq, rsGU, voff, so, NT4_ST_AUX_BIG);
                    __builtin_amdgcn_raw_buffer_store_b128(uq, rsGU, voff, so + up_off, NT4_ST_AUX_BIG);
                    __builtin_amdgcn_raw_buffer_store_b128(act_of(gq, uq), rsACT, voff2, so2, NT4_ST_AUX_BIG);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // SwiGLU backward: the tile is d act [256 rows, 256 columns of I], never stored: d gate = d act * up * silu'(gate),
            // d up = d act * silu(gate), gate/up read from GU, written to DGU (same formulas as ssi_swiglu_bwd)
            const int64_t c0 = (int64_t)tn * BN + wn * NT4_WN;
            const __amdgpu_buffer_rsrc_t rsGU = make_rsrc(ea.in2 + row0 * ea.ld_in2 + c0);
            const __amdgpu_buffer_rsrc_t rsDGU = make_rsrc(ea.out2 + row0 * ea.ld_out2 + c0);
            const int voff = (int)((lane_row * ea.ld_in2 + lane_col) * 2), voff2 = (int)((lane_row * ea.ld_out2 + lane_col) * 2);
            const int up_off = (int)(ea.inter * 2);
            u32x4 pg[2][2], pu[2][2];  // [unit parity][half]; unit = (m-tile i, duo jd), fetched one unit ahead
            auto ldprev = [&](int u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int so = (int)((((u >> 1) * 16 + h * 8) * ea.ld_in2 + (u & 1) * 64) * 2);
                    pg[u & 1][h] = __builtin_amdgcn_raw_buffer_load_b128(rsGU, voff, so, 0);
                    pu[u & 1][h] = __builtin_amdgcn_raw_buffer_load_b128(rsGU, voff, so + up_off, 0);
                }
            };
            auto grads = [&](const u32x4& dq, const u32x4& gq, const u32x4& uq, u32x4& og_out, u32x4& ou_out) {
                const bf16x8 dv = __builtin_bit_cast(bf16x8, dq), gv = __builtin_bit_cast(bf16x8, gq), uv = __builtin_bit_cast(bf16x8, uq);
                bf16x8 og, ou;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float dg, du;
                    ssi_swiglu_bwd_elem<bf16_t>((float)gv[e], (float)uv[e], (float)dv[e], dg, du);
                    ou[e] = (bf16_t)du;
                    og[e] = (bf16_t)dg;
                }
                og_out = __builtin_bit_cast(u32x4, og);
                ou_out = __builtin_bit_cast(u32x4, ou);
            };
            ldprev(0);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u + 1 < 16) ldprev(u + 1);
                const int i = u >> 1, jd = u & 1;
                u32x4 lo, hi;
                duo(i, jd * 4, scale_c, lo, hi);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 og, ou;
                    grads(h ? hi : lo, pg[u & 1][h], pu[u & 1][h], og, ou);
                    const int so2 = (int)(((i * 16 + h * 8) * ea.ld_out2 + jd * 64) * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(og, rsDGU, voff2, so2, NT4_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(ou, rsDGU, voff2, so2 + up_off, NT4_ST_AUX);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        };
        if (al == 1.f) epilogue(std::false_type{});
        else epilogue(std::true_type{});
        }
        cur = nxt;
        if (cur >= 0) nxt = receive_tile();
        // the first fragments of the next tile (its K-step 0 sits in LDS buffer 0) are read again here rather than kept live
        // across the epilogue (the DMA path keeps its F0 live: it has no staging registers to make room for)
        if constexpr (!DMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { rdA1(S0[q], tileA(0), 0, 0, q); rdB1(S1[q], tileB(0), 0, 0, q); }
        }
    }
    // Behind its last tile the fetch cursor kept requesting (valid memory, never consumed): those LDS-DMA requests must have landed before
    // the workgroup ends — its LDS goes to whichever workgroup the CU runs next.  (Also waits for the last epilogue's stores: the kernel's
    // end waits for them anyway.)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the last workgroup to get here clears the scheduler slot for the launch that uses it next
    if (dynamic && tid == 0 && atomicAdd(&sched[8], 1) == G - 1) {  // plain stores: nobody reads the slot before this kernel has ended
        volatile int* vs = sched;
#pragma unroll
        for (int i = 0; i < 9; ++i) vs[i] = 0;
    }
}

template <bool A_COL, bool B_COL, int EPI, int PREV, bool SPLITK = false>
__global__ __launch_bounds__(NT4_THREADS, 1) void gemm_nt4_kernel(int tiles_m, int tiles_n, int64_t K, const bf16_t* __restrict__ A,
                                                                  int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                  bf16_t* __restrict__ C, int64_t ldc, const bf16_t* __restrict__ R,
                                                                  float alpha, const float* __restrict__ alpha_dev, EpiArgs ea, int slot,
                                                                  int splits, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nt4_body<A_COL, B_COL, EPI, PREV, SPLITK, false>(smem, tiles_m, tiles_n, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, ea, slot, splits, slabs);
}

// the LDS-DMA main loop (same tile walk, scheduler and epilogues); BATCHED: several problems of one shape in one launch (EpiArgs::batch)
template <bool A_COL, bool B_COL, int EPI, int PREV, bool SPLITK = false, bool BATCHED = false>
__global__ __launch_bounds__(NT4_THREADS, 1) void gemm_nt4dma_kernel(int tiles_m, int tiles_n, int64_t K, const bf16_t* __restrict__ A,
                                                                     int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                     bf16_t* __restrict__ C, int64_t ldc, const bf16_t* __restrict__ R,
                                                                     float alpha, const float* __restrict__ alpha_dev, EpiArgs ea, int slot,
                                                                     int splits, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nt4_body<A_COL, B_COL, EPI, PREV, SPLITK, true, BATCHED>(smem, tiles_m, tiles_n, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, ea, slot, splits, slabs);
}

// Finishes split-K tiles: one wave per (output tile, wave of the GEMM workgroup, m-tile i, 64-column duo).  Sums the K-slices'
// partials (contiguous KiB reads), then the same rounding points and the same full-line stores as the direct epilogue:
// C = (accumulate ? C : 0) + bf16(alpha * sum).
__global__ __launch_bounds__(256) void nt4_splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int tiles_m, int tiles_n,
                                                                bf16_t* __restrict__ C, int64_t ldc, const bf16_t* __restrict__ R, float alpha,
                                                                const float* __restrict__ alpha_dev, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (tile, wave, i, jd)
    const int ntiles = tiles_m * tiles_n;
    if (u >= (int64_t)ntiles * 64) return;
    const int jd = (int)(u & 1), i = (int)((u >> 1) & 7), wave = (int)((u >> 4) & 3), t = (int)(u >> 6);
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4;
    int tm, tn;
    tile_from_t(t, tiles_m, tiles_n, tm, tn);
    const float al = alpha * (alpha_dev ? *alpha_dev : 1.f);
    f32x4 sum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const float* p = slabs + (((int64_t)s * ntiles + t) * 4 + wave) * (64 * 64 * 4) + (int64_t)(i * 8 + jd * 4) * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + j * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[j][r] += v[r];
        }
    }
    auto pair = [&](const f32x4& a, const f32x4& b) {
        bf16x4 x, y;
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[r] = (bf16_t)(a[r] * al); y[r] = (bf16_t)(b[r] * al); }
        const u32x2 xu = __builtin_bit_cast(u32x2, x), yu = __builtin_bit_cast(u32x2, y);
        const auto s0 = __builtin_amdgcn_permlane16_swap(xu[0], yu[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(xu[1], yu[1], false, false);
        u32x4 o;
        o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
        return o;
    };
    const u32x4 oa = pair(sum[0], sum[1]), ob = pair(sum[2], sum[3]);
    u32x4 lo, hi;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        lo[d] = (unsigned)__builtin_amdgcn_update_dpp((int)oa[d], (int)ob[d], 0x118 /* row_shr:8 */, 0xF, 0xC, false);
        hi[d] = (unsigned)__builtin_amdgcn_update_dpp((int)ob[d], (int)oa[d], 0x108 /* row_shl:8 */, 0xF, 0x3, false);
    }
    const int lane_row = lane & 7, lane_col = ((lane >> 3) & 1) * 32 + (g & 1) * 16 + (g >> 1) * 8;
    bf16_t* base = C + ((int64_t)tm * BM + wm * NT4_WM + i * 16 + lane_row) * ldc + (int64_t)tn * BN + wn * NT4_WN + jd * 64 + lane_col;
    // previous values (accumulate) or the residual (same rows, same leading dimension as C) are added to the ROUNDED product, as the
    // unsplit epilogue does (F.linear then `+`: two roundings)
    const int64_t r_off = R ? (R - C) : 0;
    auto finish = [&](u32x4 o, bf16_t* dst) {
        if (accumulate || R) {
            bf16x8 ob8 = __builtin_bit_cast(bf16x8, o);
            const bf16x8 c = *reinterpret_cast<const bf16x8*>(dst + r_off);
#pragma unroll
            for (int e = 0; e < 8; ++e) ob8[e] = (bf16_t)((float)ob8[e] + (float)c[e]);
            o = __builtin_bit_cast(u32x4, ob8);
        }
        *reinterpret_cast<u32x4*>(dst) = o;
    };
    finish(lo, base);
    finish(hi, base + 8 * ldc);
}

template <bool A_COL, bool B_COL, int EPI, int PREV, bool SPLITK = false, bool DMA = false, bool BATCHED = false>
int launch_nt4(int tiles_m, int tiles_n, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               const void* R, float alpha, const float* alpha_dev, hipStream_t st, EpiArgs ea = EpiArgs{nullptr, 0, nullptr, 0, 0},
               int splits = 1, float* slabs = nullptr) {
    void (*kern)(int, int, int64_t, const bf16_t*, int64_t, const bf16_t*, int64_t, bf16_t*, int64_t, const bf16_t*, float, const float*, EpiArgs, int, int,
                 float*);
    static_assert(!BATCHED || (DMA && !SPLITK && EPI == EPI_PLAIN), "batched form: LDS-DMA loop, plain epilogue, no split-K");
    if constexpr (DMA) kern = gemm_nt4dma_kernel<A_COL, B_COL, EPI, PREV, SPLITK, BATCHED>;
    else kern = gemm_nt4_kernel<A_COL, B_COL, EPI, PREV, SPLITK>;
    // one-time set-up per instantiation as C++11 thread-safe statics: the forward thread and autograd's backward thread may both be the
    // first to launch a given form
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NT4_LDS_BYTES);
    if (attr_rc != hipSuccess) { ssi_set_error("gemm_nt4: cannot reserve %d B of LDS: %s", NT4_LDS_BYTES, hipGetErrorString(attr_rc)); return SSI_ERR_HIP + (int)attr_rc; }
    static const int num_cu = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    const int ntiles = tiles_m * tiles_n * (SPLITK ? splits : 1) * (BATCHED ? ea.batch : 1);
    int grid = ntiles < num_cu ? ntiles : num_cu;
    const int slot = g_nt4_dynamic.load(std::memory_order_relaxed) ? (int)(nt4_next_slot() & 15) : -1;  // consecutive launches: different slots
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT4_THREADS), NT4_LDS_BYTES, st, tiles_m, tiles_n, K, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, (bf16_t*)C, ldc, (const bf16_t*)R, alpha, alpha_dev, ea, slot, splits, slabs);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// Main loop of the persistent kernel per operand form: 1 = LDS-DMA (default: +7..15 % on every form of the step, profiles/r02_*), 0 = the
// register-staged loop (kept for A/B builds: -DSSI_NT_DMA=0 -DSSI_NN_DMA=0 -DSSI_TN_DMA=0)
#ifndef SSI_NT_DMA
#define SSI_NT_DMA 1
#endif
constexpr bool NT_DMA = SSI_NT_DMA != 0;
#ifndef SSI_NN_DMA
#define SSI_NN_DMA 1
#endif
#ifndef SSI_TN_DMA
#define SSI_TN_DMA 1
#endif
constexpr bool NN_DMA = SSI_NN_DMA != 0, TN_DMA = SSI_TN_DMA != 0;
bool nt4_ok(int64_t K) { return K % (2 * BK) == 0 && K >= 4 * BK && ssi_get_impl() != SSI_IMPL_MFMA_WG8; }
// buffer-load offsets are 32-bit: a tile's rows (k-contiguous) or one K-step's k-rows (k-strided) must stay within 2 GiB
bool nt4_ld_ok(int64_t lda, int64_t ldb) { return 256 * lda * 2 < (1LL << 31) && 256 * ldb * 2 < (1LL << 31); }

}  // namespace

void ssi_gemm_mfma_set_dynamic_tiles(int on) { g_nt4_dynamic.store(on ? 1 : 0, std::memory_order_relaxed); }

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
        case SSI_GEMM_NT:
            if (nt4_ok(K) && nt4_ld_ok(lda, ldb) && !(accumulate && R)) {
                if (accumulate) return launch_nt4<false, false, EPI_PLAIN, 1, false, NT_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                if (R) return launch_nt4<false, false, EPI_PLAIN, 2, false, NT_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                return launch_nt4<false, false, EPI_PLAIN, 0, false, NT_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
            }
            GO(false, false);
        case SSI_GEMM_NN:
            if (nt4_ok(K) && nt4_ld_ok(lda, ldb) && !(accumulate && R)) {
                if (accumulate) return launch_nt4<false, true, EPI_PLAIN, 1, false, NN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                if (R) return launch_nt4<false, true, EPI_PLAIN, 2, false, NN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                return launch_nt4<false, true, EPI_PLAIN, 0, false, NN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
            }
            GO(false, true);
        case SSI_GEMM_TN:
            if (nt4_ok(K) && nt4_ld_ok(lda, ldb) && !(accumulate && R)) {
                if (accumulate) return launch_nt4<true, true, EPI_PLAIN, 1, false, TN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                if (R) return launch_nt4<true, true, EPI_PLAIN, 2, false, TN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
                return launch_nt4<true, true, EPI_PLAIN, 0, false, TN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, st);
            }
            GO(true, true);
    }
#undef GO
    return SSI_ERR_ARG;
}

// `batch` problems of one shape in one launch of the persistent kernel: the weight-gradient form (TN) on the LDS-DMA loop, no residual;
// false = not taken, the caller loops over ssi_gemm
bool ssi_gemm_mfma_bf16_batched(int layout, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t bsA, const void* B,
                                int64_t ldb, int64_t bsB, void* C, int64_t ldc, int64_t bsC, float alpha, const float* alpha_dev, int accumulate,
                                void* stream, int* rc) {
    if (layout != SSI_GEMM_TN || !TN_DMA || !nt4_ok(K) || !nt4_ld_ok(lda, ldb) || (bsA | bsB | bsC) % 8 || (M / BM) * (N / BN) * batch > (1LL << 30))
        return false;
    const int tm = (int)(M / BM), tn = (int)(N / BN);
    auto st = (hipStream_t)stream;
    EpiArgs ea{nullptr, 0, nullptr, 0, 0};
    ea.batch = batch; ea.bsA = bsA; ea.bsB = bsB; ea.bsC = bsC;
    *rc = accumulate ? launch_nt4<true, true, EPI_PLAIN, 1, false, true, true>(tm, tn, K, A, lda, B, ldb, C, ldc, nullptr, alpha, alpha_dev, st, ea)
                     : launch_nt4<true, true, EPI_PLAIN, 0, false, true, true>(tm, tn, K, A, lda, B, ldb, C, ldc, nullptr, alpha, alpha_dev, st, ea);
    return true;
}

int ssi_gemm_mfma_bf16_splitk(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                              int64_t ldb, void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev,
                              int accumulate, int splits, float* slabs, void* stream) {
    const int tm = (int)(M / BM), tn = (int)(N / BN);
    auto st = (hipStream_t)stream;
    // on the persistent kernel: units = tile x K-slice, every slice at least 6 K-steps.  Round 1-3: the weight-gradient form only; round 4: the
    // k-contiguous (NT) and data-gradient (NN) forms too, for output grids that leave CUs idle at small batches (T = 4096: W_o, W2 forward and the
    // data gradients of the N = 2048 projections have 128 tiles) or fill the last round badly (ragged T).  A residual is added by the reduction
    // pass (not together with accumulate, like the unsplit form).
    if (nt4_ok(K) && nt4_ld_ok(lda, ldb) && !(R && accumulate) && K / BK / splits >= 6 && ldc % 8 == 0) {
        const EpiArgs none{nullptr, 0, nullptr, 0, 0};
        int rc = SSI_ERR_ARG;
        if (layout == SSI_GEMM_TN) rc = launch_nt4<true, true, EPI_PLAIN, 0, true, TN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, nullptr, 1.f, nullptr, st, none, splits, slabs);
        else if (layout == SSI_GEMM_NT) rc = launch_nt4<false, false, EPI_PLAIN, 0, true, NT_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, nullptr, 1.f, nullptr, st, none, splits, slabs);
        else if (layout == SSI_GEMM_NN) rc = launch_nt4<false, true, EPI_PLAIN, 0, true, NN_DMA>(tm, tn, K, A, lda, B, ldb, C, ldc, nullptr, 1.f, nullptr, st, none, splits, slabs);
        if (rc) return rc;
        hipLaunchKernelGGL(nt4_splitk_reduce_kernel, dim3((unsigned)ssi_cdiv((int64_t)tm * tn * 64, 4)), dim3(256), 0, st, slabs, splits, tm, tn,
                           (bf16_t*)C, ldc, (const bf16_t*)R, alpha, alpha_dev, accumulate);
        SSI_LAUNCH_CHECK();
        return SSI_OK;
    }
#define GO(AC, BC) return launch<AC, BC, true>(tm, tn, K, A, lda, B, ldb, C, ldc, R, alpha, alpha_dev, accumulate, st, splits, slabs)
    switch (layout) {
        case SSI_GEMM_NT: GO(false, false);
        case SSI_GEMM_NN: GO(false, true);
        case SSI_GEMM_TN: GO(true, true);
    }
#undef GO
    return SSI_ERR_ARG;
}

// QKV projection with the RoPE rotation of the q / k heads in the epilogue (head_dim 64, rot_cols % 256 == 0); false = not taken
bool ssi_gemm_rope_mfma(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                        const float* rope, const int32_t* positions, int64_t seq, int64_t rot_cols, void* stream, int* rc) {
    if (!nt4_ok(K) || !nt4_ld_ok(lda, ldb) || M % BM || N % BN || rot_cols % BN || rot_cols > N || lda % 8 || ldb % 8 || ldc % 8) return false;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return false;
    EpiArgs ea{nullptr, 0, nullptr, 0, 0};
    ea.rope = rope; ea.pos = positions; ea.seq = seq; ea.rot_cols = rot_cols;
    *rc = launch_nt4<false, false, EPI_ROPE, 0, false, NT_DMA>((int)(M / BM), (int)(N / BN), K, A, lda, B, ldb, C, ldc, nullptr, 1.f, nullptr,
                                                (hipStream_t)stream, ea);
    return true;
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
    if (nt4_ok(K))
        return launch_nt4<false, false, EPI_SWIGLU_FWD, 0, false, NT_DMA>((int)(M / BM), (int)(2 * inter / BN), K, X, ldx, W13, ldw, GU, ldgu, nullptr, 1.f, nullptr,
                                             (hipStream_t)stream, ea);
    // output tiles: 256 rows x (128 gate + 128 up) columns -> tiles_n = 2I / 256
    return launch<false, false, false, EPI_SWIGLU_FWD>((int)(M / BM), (int)(2 * inter / BN), K, X, ldx, W13, ldw, GU, ldgu, nullptr, 1.f,
                                                       nullptr, 0, (hipStream_t)stream, 1, nullptr, ea);
}

int ssi_gemm_swiglu_bwd_mfma(int layout, int64_t M, int64_t inter, int64_t K, const void* DY, int64_t lddy, const void* W2, int64_t ldw,
                             const void* GU, int64_t ldgu, void* DGU, int64_t lddgu, void* stream) {
    EpiArgs ea{(bf16_t*)DGU, lddgu, (const bf16_t*)GU, ldgu, inter};
    // d act [M, I] = DY [M, K] * W2 (NN: W2 [K, I]) or * W2T^T (NT: transposed copy [I, K]); the tile never reaches memory
    const int tm = (int)(M / BM), tn = (int)(inter / BN);
    if (layout == SSI_GEMM_NN) {
        if (nt4_ok(K) && nt4_ld_ok(lddy, ldw))
            return launch_nt4<false, true, EPI_SWIGLU_BWD, 0, false, NN_DMA>(tm, tn, K, DY, lddy, W2, ldw, DGU, lddgu, nullptr, 1.f, nullptr, (hipStream_t)stream, ea);
        return SSI_ERR_UNSUPPORTED;
    }
    if (nt4_ok(K) && nt4_ld_ok(lddy, ldw))
        return launch_nt4<false, false, EPI_SWIGLU_BWD, 0, false, NT_DMA>(tm, tn, K, DY, lddy, W2, ldw, DGU, lddgu, nullptr, 1.f, nullptr, (hipStream_t)stream, ea);
    return launch<false, false, false, EPI_SWIGLU_BWD>(tm, tn, K, DY, lddy, W2, ldw, DGU, lddgu, nullptr, 1.f, nullptr, 0, (hipStream_t)stream, 1,
                                                       nullptr, ea);
}

// NN form of the fused SwiGLU backward exists only on the persistent kernel
bool ssi_gemm_swiglu_bwd_nn_supported(int64_t K, int64_t lddy, int64_t ldw) { return nt4_ok(K) && nt4_ld_ok(lddy, ldw); }
