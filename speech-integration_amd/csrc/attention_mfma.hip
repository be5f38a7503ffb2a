// Causal GQA flash attention on the gfx950 matrix cores (bf16, head_dim 64) — forward, dQ and dK/dV.
// Replaces F.scaled_dot_product_attention(is_causal=True) inside torchtune's MultiHeadAttention and its autograd
// (SURVEY.md §2.3 K5/K10).  qkv is the fused projection output [B*S, (H + 2 KV) * 64] after RoPE.
//
// Orientation (all three kernels): scores are produced TRANSPOSED or with the reduction index on the accumulator's ROW
// axis, so that the 32x32 accumulator of one v_mfma_f32_32x32x16_bf16 is directly the B operand of the next product
// (no LDS round trip, no lane shuffles for P):
//   forward : S^T[key][q] = K Q^T      -> P^T -> O^T[d][q]  += V^T[d][key] P^T[key][q]      (row statistics per LANE)
//   dQ      : S^T, dP^T[key][q] = V dO^T -> dS^T -> dQ^T[d][q] += K^T[d][key] dS^T[key][q]
//   dK/dV   : S[q][key] = Q K^T, dP[q][key] = dO V^T -> P, dS -> dV^T[d][key] += dO^T[d][q] P[q][key],
//             dK^T[d][key] += Q^T[d][q] dS[q][key]                                          (key on the lane, sums in regs)
// k-contiguous operands come from LDS by ds_read_b128, k-strided ones by ds_read_b64_tr_b16 (hardware transpose); tiles
// are [rows][64] bf16 (128-B rows) with a 16-B-chunk XOR swizzle chosen per tile for the way it is read.
// Workgroup = 4 waves; the waves of a workgroup share one kv head (K/V tiles staged once for the 4 query heads of a GQA
// group).  No atomics anywhere: dQ gets its own pass (recomputing S and dP) so every output has exactly one writer and
// results are bitwise reproducible.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <vector>
#include <type_traits>
#include "common_hip.h"

#ifndef DKV_RING
#define DKV_RING 6
#endif
#ifndef DKV_WAVES
#define DKV_WAVES 2
#endif
namespace {

// Workgroup barrier for the LDS-DMA rings.  __syncthreads() would do, except that hipcc puts `s_waitcnt vmcnt(0)` in front of
// its s_barrier: that waits for the prefetches of the NEXT steps as well and exposes their whole memory latency every step.
// Here the counted vmcnt wait for this step's pieces is written out by the caller; only LDS traffic is drained.
// Backward of the interleaved RoPE on the 4 consecutive head dimensions d0 .. d0+3 of one row (two adjacent pairs), applied to
// the gradient AFTER its rounding to bf16 and rounded again, i.e. exactly what ssi_rope_inplace(inverse) does to the stored
// tensor (torchtune applies RoPE as a separate bf16 -> fp32 -> bf16 op).  tb = table row of the position: [hd/2][cos, sin].
__device__ __forceinline__ bf16x4 unrope4(bf16x4 v, f32x4 cs) {  // cs = (cos, sin) of pairs d0/2 and d0/2 + 1
    const float x0 = (float)v[0], x1 = (float)v[1], x2 = (float)v[2], x3 = (float)v[3];
    bf16x4 o;
    o[0] = (bf16_t)(x0 * cs[0] + x1 * cs[1]);
    o[1] = (bf16_t)(x1 * cs[0] - x0 * cs[1]);
    o[2] = (bf16_t)(x2 * cs[2] + x3 * cs[3]);
    o[3] = (bf16_t)(x3 * cs[2] - x2 * cs[3]);
    return o;
}
__device__ __forceinline__ bf16x4 unrope4(bf16x4 v, const float* __restrict__ tb, int d0) {
    return unrope4(v, *reinterpret_cast<const f32x4*>(tb + d0));
}

// Workgroup -> (rank of the block inside its (batch, kv head) pair, pair).  Workgroups go to the 8 XCDs round-robin by
// blockIdx, and under the causal mask a block's work is proportional to its rank, so a plain "block = blockIdx % n" map hands
// XCD x only the blocks of rank x, x + 8, ...: 2.4x the work for XCD 0 as for XCD 7 at 16 blocks per pair, and the kernel lasts
// as long as XCD 0.  Here every XCD gets whole pairs (n_pairs / 8 of them: equal work, and a pair's K / V or Q / dO stay in one
// L2) and meets their blocks in rank order — with `rank` counting from the heaviest block, longest first across its pairs.
__device__ __forceinline__ void block_to_work(int n_blocks, int n_pairs, int& rank, int& pair) {
    const int i = (int)blockIdx.x;
    if (n_pairs % 8 == 0) {
        const int ppx = n_pairs / 8, xcd = i & 7, j = i >> 3;
        rank = j / ppx;
        pair = xcd * ppx + j % ppx;
    } else {
        rank = i % n_blocks;
        pair = i / n_blocks;
    }
}

__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// -DATTN_TRACE (debug build, tools/attn_trace.py): every workgroup leaves its start, the start and end of its tile loop and its end on the
// 100 MHz constant clock, where it ran (XCC, SE, CU) and its work (tiles / steps) in a device-side table: the occupancy timeline of a launch —
// per-workgroup cost against its tile count, idle slots, the tail.  Never part of the product build (the extra export would also fail
// tests/test_abi.py).
#ifdef ATTN_TRACE
constexpr int TRACE_MAX = 8192;
__device__ unsigned long long g_attn_trace[3][TRACE_MAX][6];
#define TRACE_BEGIN() const unsigned long long tr_t0_ = __builtin_amdgcn_s_memrealtime(); unsigned long long tr_ta_ = 0, tr_tb_ = 0
#define TRACE_LOOP_BEGIN() tr_ta_ = __builtin_amdgcn_s_memrealtime()   /* prologue issued (loads in flight), tile loop starts */
#define TRACE_LOOP_END() tr_tb_ = __builtin_amdgcn_s_memrealtime()     /* tile loop done, epilogue starts */
#define TRACE_END(k, work)                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < TRACE_MAX) {                                                                             \
        unsigned hw_, xcc_;                                                                                                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                      \
        g_attn_trace[k][blockIdx.x][0] = tr_t0_;                                                                                  \
        g_attn_trace[k][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();                                                        \
        g_attn_trace[k][blockIdx.x][2] = ((unsigned long long)xcc_ << 32) | hw_;                                                  \
        g_attn_trace[k][blockIdx.x][3] = (unsigned long long)(work);                                                              \
        g_attn_trace[k][blockIdx.x][4] = tr_ta_;                                                                                  \
        g_attn_trace[k][blockIdx.x][5] = tr_tb_;                                                                                  \
    }
#else
#define TRACE_BEGIN()
#define TRACE_LOOP_BEGIN()
#define TRACE_LOOP_END()
#define TRACE_END(k, work)
#endif

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_TAU = 5.545177444479562f;  // 8 ln 2

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// 16-B-chunk XOR swizzles of a [rows][64] bf16 tile (two 128-B rows per 256-B bank row):
//   SWZ_ROW  f = (row >> 1) & 7: the 8 same-parity rows of a ds_read_b128 lane group land on 8 different chunks   (row reads only)
//   SWZ_TR   f = 4 * bit 1 of row: rows r and r + 2 of a transposed 4-row block land on opposite halves of the row (transposed reads only)
//   SWZ_DUAL both at once: the three bits of (row >> 1) rotated so that bit 1 of the row becomes bit 2 of f — still 8 different values on
//            the row-read groups, and r / r + 2 differ in bit 2.  With SWZ_ROW a tile that is ALSO read transposed (K in dQ, Q and dO in
//            dK/dV) cost every ds_read_b64_tr_b16 a 2-way conflict (SQ_LDS_BANK_CONFLICT = one extra cycle per LDS instruction).
enum { SWZ_ROW = 0, SWZ_TR = 1, SWZ_DUAL = 2 };
template <int SWZ> __device__ __forceinline__ int swz(int row) {
    if (SWZ == SWZ_ROW) return (row >> 1) & 7;
    if (SWZ == SWZ_TR) return ((row >> 1) & 1) << 2;
    return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}

// 32 rows x 16 k fragment of a [rows][64] tile: lane l holds row = row_base + (l & 31), k = 16 ks + 8 (l >> 5) + j
template <int SWZ> __device__ __forceinline__ bf16x8 frag_row(const char* tile, int row_base, int ks, int lane) {
    const int row = row_base + (lane & 31);
    const int chunk = (2 * ks + (lane >> 5)) ^ swz<SWZ>(row);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + chunk * 16);
}

// transposed fragment: lane l holds column c = cbase + (l & 31) of tile rows kbase + {8 (j >> 2) + 4 (l >> 5) + (j & 3)}, j = 0..7
// (the k order in which a 32x32 accumulator, converted to bf16, presents itself as an MFMA operand)
template <int SWZ> __device__ __forceinline__ bf16x8 frag_tr(const char* tile, int kbase, int cbase, int lane) {
    const int G = lane >> 4, h = G >> 1, i = lane & 15, q = i >> 2, p = i & 3;
    const int chunk = ((cbase + 16 * (G & 1)) >> 3) + (p >> 1);
    const int r0 = kbase + 4 * h + q, r1 = r0 + 8;
    const char* a0 = tile + r0 * 128 + ((chunk ^ swz<SWZ>(r0)) * 16) + 8 * (p & 1);
    const char* a1 = tile + r1 * 128 + ((chunk ^ swz<SWZ>(r1)) * 16) + 8 * (p & 1);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

// one half (4 of the 8 k rows: half 0 = rows kbase + 4 (l >> 5) + 0..3, half 1 = those + 8) of frag_tr: one ds_read_b64_tr_b16
template <int SWZ> __device__ __forceinline__ s16x4 frag_tr_half(const char* tile, int kbase, int cbase, int lane, int half) {
    const int G = lane >> 4, h = G >> 1, i = lane & 15, q = i >> 2, p = i & 3;
    const int chunk = ((cbase + 16 * (G & 1)) >> 3) + (p >> 1);
    const int r = kbase + 4 * h + q + 8 * half;
    const char* a = tile + r * 128 + ((chunk ^ swz<SWZ>(r)) * 16) + 8 * (p & 1);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
}

// registers 8 s .. 8 s + 7 of a 32x32 accumulator as a bf16 operand fragment (k-step s)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)a[8 * s + j];
    return f;
}

__device__ __forceinline__ bf16x8 scale_frag(bf16x8 v, float s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] * s);
    return v;
}

// stage a [ROWS][64] bf16 tile global -> registers -> LDS (swizzled), split so the loads fly under compute (T14)
template <int ROWS, int NTHR> struct TileStage {
    static constexpr int N = ROWS * 8 / NTHR;  // 16-B chunks per thread
    u32x4 r[N];
    __device__ __forceinline__ void load(const bf16_t* g, int64_t ld, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NTHR;
            r[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)(c >> 3) * ld + (c & 7) * 8);
        }
    }
    template <int SWZ> __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * NTHR, row = c >> 3, chunk = (c & 7) ^ swz<SWZ>(row);
            *reinterpret_cast<u32x4*>(tile + row * 128 + chunk * 16) = r[i];
        }
    }
};

// LDS-DMA requests are written as inline asm, not as __builtin_amdgcn_global_load_lds: the compiler knows that the builtin writes
// LDS and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint (every ds_read_b64_tr_b16 here), which
// waits for the prefetches of the LATER tiles as well and turns a ring of N tiles into a ring of one.  With the asm form the only
// waits are the counted ones written out next to the ring barriers.  A request = one wave-instruction: lane l's 16 (or 4) bytes at
// rsrc base + voff(l) + soff go to LDS byte M0 + 16 l (4 l); the bank swizzle of a tile image is therefore applied on the per-lane
// SOURCE offset.  The buffer form keeps the per-lane part of the address a constant 32-bit VGPR and the moving part a scalar.
typedef __attribute__((address_space(3))) char lds_c;
constexpr unsigned BUF_RSRC_WORD3 = 0x00020000u;  // raw buffer, 32-bit data format
__device__ __forceinline__ u32x4 buffer_rsrc(const void* base) {  // stride 0, 2 GiB window
    const uint64_t a = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = 0x7fffffffu;
    r[3] = BUF_RSRC_WORD3;
    return r;
}
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, u32x4 rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4(unsigned lds_dst, unsigned voff, u32x4 rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// K and V tiles (64 keys x 64 d each, 8 KiB + 8 KiB) of one (batch, kv head) into ring slots: each of the 4 waves moves two 1-KiB
// pieces (8 rows x 128 B) of K and two of V = 4 requests per wave per tile.
// Waves per workgroup of the forward and dQ kernels: 4 (one 32-query block x the 4 heads of a GQA group, two such workgroups per CU).  With 8
// (-DATTN_NW=8: two query blocks share each K / V tile, every wave issues 2 LDS-DMA requests per tile instead of 4, one workgroup per CU) the
// forward took 197-201 us against 179 and the backward 553-561 against 549: what the halved request count saves, the 8-wave barrier and the
// loss of the second, unsynchronised workgroup cost again.
#ifndef ATTN_NW
#define ATTN_NW 4
#endif
constexpr int ANW = ATTN_NW, ANP = 8 / ATTN_NW;  // waves per workgroup, K (and V) pieces per wave and tile
template <int SWZ_K, int SWZ_V> struct KvTileDma {
    u32x4 rs;             // base = K rows of the batch, column block of the kv head
    unsigned vk[ANP], vv[ANP];  // per-lane source byte offsets of this wave's K and V pieces inside a tile
    unsigned lds_piece;   // LDS byte address of this wave's first piece in slot 0
    unsigned tile_bytes;  // source bytes from one tile to the next
    __device__ __forceinline__ void init(const bf16_t* kbase, int64_t ld, int kv_cols, const char* smem, int wave, int lane) {
        rs = buffer_rsrc(kbase);
#pragma unroll
        for (int p = 0; p < ANP; ++p) {
            const int row = (p * ANW + wave) * 8 + (lane >> 3);
            vk[p] = (unsigned)((row * ld + ((lane & 7) ^ swz<SWZ_K>(row)) * 8) * 2);
            vv[p] = (unsigned)((row * ld + kv_cols + ((lane & 7) ^ swz<SWZ_V>(row)) * 8) * 2);
        }
        lds_piece = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + (unsigned)wave * 1024u);
        tile_bytes = (unsigned)(64 * ld * 2);
    }
    __device__ __forceinline__ void tile(int t, unsigned slot_bytes) const {
        const unsigned soff = (unsigned)t * tile_bytes;
#pragma unroll
        for (int p = 0; p < ANP; ++p) {
            dma16(lds_piece + slot_bytes + p * ANW * 1024, vk[p], rs, soff);
            dma16(lds_piece + slot_bytes + 8192 + p * ANW * 1024, vv[p], rs, soff);
        }
    }
    // request i (0 .. 2 ANP - 1) of a tile alone: K piece i >> 1 (even i) or V piece i >> 1 (odd i)
    __device__ __forceinline__ void piece(int t, unsigned slot_bytes, int i) const {
        const unsigned soff = (unsigned)t * tile_bytes;
        const int p = i >> 1;
        if (i & 1) dma16(lds_piece + slot_bytes + 8192 + p * ANW * 1024, vv[p], rs, soff);
        else dma16(lds_piece + slot_bytes + p * ANW * 1024, vk[p], rs, soff);
    }
};

// A [64 rows][64] bf16 tile (64 consecutive rows of one head's column block) into LDS by ONE wave, SWZ_ROW image: 8 requests of 8 rows x
// 128 B — whole 128-B lines, where a fragment load straight from global memory (lane = row) touches 32 rows x 32 B per instruction.
struct RowTileDma {
    u32x4 rs;
    unsigned voff[2];  // per-lane source byte offset inside a request, for even / odd requests (the swizzle's bit 2 follows the request)
    unsigned step;     // source bytes from one request to the next
    __device__ __forceinline__ void init(const bf16_t* base, int64_t ld, int lane) {
        rs = buffer_rsrc(base);
#pragma unroll
        for (int par = 0; par < 2; ++par)  // swz<SWZ_ROW>(8 i + (l >> 3)) = (l >> 4) ^ 4 (i & 1)
            voff[par] = (unsigned)(((lane >> 3) * ld + ((lane & 7) ^ (lane >> 4) ^ (4 * par)) * 8) * 2);
        step = (unsigned)(8 * ld * 2);
    }
    __device__ __forceinline__ void request(int i, unsigned lds_tile) const { dma16(lds_tile + i * 1024, voff[i & 1], rs, (unsigned)i * step); }
};

// =====================================================================================================================
// forward
// =====================================================================================================================
// grid.x = B * KV * (S / (32 * QPW)),  QPW = ANW / rep q-blocks per workgroup; wave w: head kvh*rep + w % rep, q-block w / rep
__global__ __launch_bounds__(64 * ANW, ANW == 8 ? 1 : 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, int64_t ld, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, const int32_t* __restrict__ doc_start, int S, int H,
                                                       int KV) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 2 * 8192];  // ring of 3 x [K | V][64][64] bf16
    TRACE_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rep = H / KV, qpw = ANW / rep;
    const int nqb = S / (32 * qpw);
    // heavy q-blocks first
    int rank_, pair_;
    block_to_work(nqb, (int)(gridDim.x / nqb), rank_, pair_);
    const int qgrp = nqb - 1 - rank_;
    const int kvh = pair_ % KV;
    const int b = pair_ / KV;
    const int head = kvh * rep + wave % rep;
    const int q0 = (qgrp * qpw + wave / rep) * 32;
    const int q_last_wg = (qgrp * qpw + qpw - 1) * 32 + 31;
    const int nt = q_last_wg / 64 + 1;
    const int h = lane >> 5;
    const int64_t row0 = (int64_t)b * S;
    const bf16_t* kbase = qkv + row0 * ld + (int64_t)H * HD + (int64_t)kvh * HD;
    // packed rows: a query sees keys doc_start <= key <= query.  doc_start is non-decreasing along a row, so the first key
    // tile any row of the workgroup / wave needs, and whether a tile needs the document mask, follow from the end rows.
    const int qg_ = q0 + (lane & 31);
    const int ds = doc_start ? doc_start[row0 + qg_] : 0;                       // this lane's query
    const int ds_lo = doc_start ? doc_start[row0 + q0] : 0;                     // first row of the wave
    const int ds_hi = doc_start ? doc_start[row0 + q0 + 31] : 0;                // last row of the wave
    const int t_first = doc_start ? doc_start[row0 + qgrp * qpw * 32] / 64 : 0;  // first tile of the workgroup

    // Q as the B operand of S^T = K Q^T, pre-scaled by 1/sqrt(64) = 2^-3 (exact in bf16)
    bf16x8 qf[4];
    {
        const bf16_t* qrow = qkv + (row0 + q0 + (lane & 31)) * ld + (int64_t)head * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = scale_frag(*reinterpret_cast<const bf16x8*>(qrow + 16 * ks), 0.125f);
    }
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;  // reference max (scaled-score units) and this half-wave's partial row sum
    float mb = 0.f;                   // m in exp2 units (0 while the row has seen no key)
    const int qg = q0 + (lane & 31);

    // ring of 3 tile slots filled by LDS-DMA two tiles ahead (4 requests per wave per tile)
    KvTileDma<SWZ_ROW, SWZ_TR> kvdma;
    kvdma.init(kbase, ld, KV * HD, smem, wave, lane);
    kvdma.tile(t_first, 0);
    if (t_first + 1 < nt) kvdma.tile(t_first + 1, 16384);
    auto tile_step = [&](int t, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;  // compile-time ring slot: LDS addresses = hoisted lane base + immediate
        const char* kt = smem + BUF * 16384;
        const char* vt = kt + 8192;
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ANP) : "memory");  // own pieces of tile t landed (tile t+1 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();  // everybody's pieces landed; the slot of tile t-1 is free again
        if (t + 2 < nt) kvdma.tile(t + 2, ((BUF + 2) % 3) * 16384);
        const int k0 = t * 64;
        if (k0 <= q0 + 31 && k0 + 63 >= ds_lo) {  // wave-uniform: this tile intersects the visible range of the wave's rows
            // All 8 K fragments are requested before the first product and all 16 V fragments right behind the S^T products (they land
            // under the softmax).  Left to itself the compiler reads each fragment into the same registers right in front of its MFMA
            // (read, lgkmcnt(0), MFMA, read, ...), which exposes the LDS latency once per MFMA.
            bf16x8 kfr[2][4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kfr[kb][ks] = frag_row<SWZ_ROW>(kt, kb * 32, ks, lane);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][ks], qf[ks], sacc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 vfr[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int db = 0; db < 2; ++db) vfr[s][db] = frag_tr<SWZ_TR>(vt, s * 16, db * 32, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 63 > q0 || k0 < ds_hi) {  // edge tile: mask keys beyond the query or before its document
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kb * 32 + rowmap(r, h);
                        if (key > qg || key < ds) sacc[kb][r] = -INFINITY;
                    }
            }
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Deferred rescale: the reference maximum m of a row moves only when the tile's maximum exceeds it by more than
            // RESCALE_TAU, and then for the whole wave at once (wave-uniform branch), so most tiles skip the 32 multiplies of O^T and the
            // extra exponential.  With a stale m the probabilities of a tile are at most e^TAU = 256 instead of 1: same relative precision in
            // bf16, sums and O^T in fp32, and out = O / l, lse = m + log l do not depend on which m was used.
            if (__builtin_amdgcn_ballot_w64(mx > m + RESCALE_TAU) != 0) {
                const float mn = fmaxf(m, mx);
                // a row whose document starts after this tile has seen no key yet (m = mn = -inf): keep its state finite
                const float mref = mn == -INFINITY ? 0.f : mn;
                const float alpha = __builtin_amdgcn_exp2f((m - mref) * LOG2E);
                mb = mref * LOG2E;
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            }
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[kb][r] * LOG2E - mb);
                    sacc[kb][r] = p;
                    rs += p;
                }
            lsum += rs;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 pf = acc_frag(sacc[s >> 1], s & 1);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s][db], pf, oacc[db], 0, 0, 0);
            }
        }
    };
    TRACE_LOOP_BEGIN();
    for (int t = t_first; t < nt; t += 3) {
        tile_step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) tile_step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nt) tile_step(t + 2, std::integral_constant<int, 2>{});
    }
    TRACE_LOOP_END();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last tile's wait left nothing in flight towards LDS; once more on every path: kernel_lint R3)
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.f / ltot;
    bf16_t* orow = out + (row0 + qg) * ((int64_t)H * HD) + (int64_t)head * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(oacc[db][4 * g + e] * inv);
            *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * h) = v;
        }
    if (h == 0) lse[((int64_t)b * H + head) * S + qg] = m + logf(ltot);
    TRACE_END(0, nt - t_first);
}


// =====================================================================================================================
// backward: dQ   (same decomposition as the forward)
// =====================================================================================================================
__global__ __launch_bounds__(64 * ANW, ANW == 8 ? 1 : 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                          const int32_t* __restrict__ doc_start, const float* __restrict__ rope,
                                                          const int32_t* __restrict__ positions, int S, int H, int KV) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 2 * 8192];  // ring of 3 x [K | V][64][64] bf16
    TRACE_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rep = H / KV, qpw = ANW / rep;
    const int nqb = S / (32 * qpw);
    int rank_, pair_;
    block_to_work(nqb, (int)(gridDim.x / nqb), rank_, pair_);
    const int qgrp = nqb - 1 - rank_;
    const int kvh = pair_ % KV;
    const int b = pair_ / KV;
    const int head = kvh * rep + wave % rep;
    const int q0 = (qgrp * qpw + wave / rep) * 32;
    const int nt = ((qgrp * qpw + qpw - 1) * 32 + 31) / 64 + 1;
    const int h = lane >> 5;
    const int64_t row0 = (int64_t)b * S;
    const bf16_t* kbase = qkv + row0 * ld + (int64_t)H * HD + (int64_t)kvh * HD;
    const int qg = q0 + (lane & 31);
    // packed rows: see attn_fwd_kernel
    const int ds = doc_start ? doc_start[row0 + qg] : 0;
    const int ds_lo = doc_start ? doc_start[row0 + q0] : 0;
    const int ds_hi = doc_start ? doc_start[row0 + q0 + 31] : 0;
    const int t_first = doc_start ? doc_start[row0 + qgrp * qpw * 32] / 64 : 0;

    bf16x8 qf[4], dof[4];
    {
        const bf16_t* qrow = qkv + (row0 + qg) * ld + (int64_t)head * HD + 8 * h;
        const bf16_t* drow = dout + (row0 + qg) * ((int64_t)H * HD) + (int64_t)head * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = scale_frag(*reinterpret_cast<const bf16x8*>(qrow + 16 * ks), 0.125f);
            dof[ks] = *reinterpret_cast<const bf16x8*>(drow + 16 * ks);
        }
    }
    const float lq = lse[((int64_t)b * H + head) * S + qg];
    // delta = rowsum(dO * O) of this lane's query row: each half-wave holds half of the row (the dO fragments are already here);
    // written out for the dK/dV kernel, which runs after this one (every (row, head) belongs to exactly one wave)
    float dl = 0.f;
    {
        const bf16_t* orow = out + (row0 + qg) * ((int64_t)H * HD) + (int64_t)head * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(orow + 16 * ks);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)dof[ks][j];
        }
        dl += __shfl_xor(dl, 32, 64);
        if (h == 0) delta[((int64_t)b * H + head) * S + qg] = dl;
    }
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

    // -DDQ_STAMP (debug build, tools/dkv_stamps.py dq): cycle totals of wave 0 per phase of a tile, left in the wave's first dq row
#ifdef DQ_STAMP
    unsigned long long qs_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long qs_last = __builtin_readcyclecounter();
    const unsigned long long qs_begin = qs_last;
    int qs_tiles = 0;
#define QSTAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); qs_acc[i] += now_ - qs_last; qs_last = now_; }
#else
#define QSTAMP(i)
#endif
    KvTileDma<SWZ_DUAL, SWZ_ROW> kvdma;
    kvdma.init(kbase, ld, KV * HD, smem, wave, lane);
    kvdma.tile(t_first, 0);
    if (t_first + 1 < nt) kvdma.tile(t_first + 1, 16384);
    auto tile_step = [&](int t, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        const char* kt = smem + BUF * 16384;
        const char* vt = kt + 8192;
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ANP) : "memory");  // own pieces of tile t landed (tile t+1 may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();  // everybody's pieces landed; the slot of tile t-1 is free again
        if (t + 2 < nt) kvdma.tile(t + 2, ((BUF + 2) % 3) * 16384);
        QSTAMP(0)  // wait + barrier + the 4 requests of tile t+2
        const int k0 = t * 64;
        if (k0 <= q0 + 31 && k0 + 63 >= ds_lo) {
#ifdef DQ_STAMP
            ++qs_tiles;
#endif
            // fragment reads ahead of the products that use them (see attn_fwd_kernel)
            bf16x8 kfr[2][4], vfr[2][4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    kfr[kb][ks] = frag_row<SWZ_DUAL>(kt, kb * 32, ks, lane);
                    vfr[kb][ks] = frag_row<SWZ_ROW>(vt, kb * 32, ks, lane);
                }
            __builtin_amdgcn_sched_barrier(0);
#ifdef DQ_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            QSTAMP(1)  // 16 row-fragment reads landed
            f32x16 sacc[2], pacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[kb][r] = -lq; pacc[kb][r] = -dl; }  // S'^T = K Q^T - lse, dP'^T = V dO^T - delta
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][ks], qf[ks], sacc[kb], 0, 0, 0);
                    pacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kb][ks], dof[ks], pacc[kb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            QSTAMP(2)  // 16 S / dP MFMAs issued
            bf16x8 ktr[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int db = 0; db < 2; ++db) ktr[s][db] = frag_tr<SWZ_DUAL>(kt, s * 16, db * 32, lane);
            __builtin_amdgcn_sched_barrier(0);
            QSTAMP(3)  // 16 transposed reads issued
            if (k0 + 63 > q0 || k0 < ds_hi) {  // edge tile: keys beyond the query or before its document contribute nothing
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float p = __builtin_amdgcn_exp2f(sacc[kb][r] * LOG2E);
                        const int key = k0 + kb * 32 + rowmap(r, h);
                        if (key > qg || key < ds) p = 0.f;
                        sacc[kb][r] = p * pacc[kb][r];  // dS^T (the 1/sqrt(d) factor is applied once at the end)
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[kb][r] = __builtin_amdgcn_exp2f(sacc[kb][r] * LOG2E) * pacc[kb][r];
            }
            QSTAMP(4)  // exponentials (includes waiting for S / dP)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 dsf = acc_frag(sacc[s >> 1], s & 1);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[s][db], dsf, dq[db], 0, 0, 0);
            }
            QSTAMP(5)  // conversions + 8 dQ MFMAs issued
        }
    };
    TRACE_LOOP_BEGIN();
    for (int t = t_first; t < nt; t += 3) {
        tile_step(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) tile_step(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nt) tile_step(t + 2, std::integral_constant<int, 2>{});
    }
    TRACE_LOOP_END();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see attn_fwd_kernel)
    bf16_t* drow = dqkv + (row0 + qg) * ld + (int64_t)head * HD;
#ifdef DQ_STAMP
    unsigned long long qs_total = __builtin_readcyclecounter() - qs_begin;
#endif
    // rope != NULL: the gradient leaves in pre-RoPE space (backward of the rotation fused here, saves a pass over dqkv)
    const float* tb = rope ? rope + (int64_t)(positions ? positions[row0 + qg] : qg) * HD : nullptr;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(dq[db][4 * g + e] * 0.125f);
            if (tb) v = unrope4(v, tb, db * 32 + 8 * g + 4 * h);
            *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * g + 4 * h) = v;
        }
#ifdef DQ_STAMP
    if (wave == 0) {  // DEBUG BUILD ONLY: lane 0's row of head `head` carries the totals (overwrites the gradient there)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            float* dbg = reinterpret_cast<float*>(drow);
            for (int i = 0; i < 6; ++i) dbg[i] = (float)qs_acc[i];
            dbg[6] = (float)qs_total;
            dbg[7] = (float)qs_tiles;
            dbg[8] = (float)(nt - t_first);
        }
    }
#endif
    TRACE_END(1, nt - t_first);
}

// =====================================================================================================================
// backward: dQ — round 4: one wave per SIMD, hand-placed software pipeline, persistent workgroups
// (4 query heads per kv head; plain causal rows: S a multiple of 128, i.e. of 64 x 2, 4 or 8 query blocks per workgroup — the host takes the
//  largest count that divides S / 64 and fills the chip; packed rows: round 5, from a work plan, see VARLEN below)
// =====================================================================================================================
// The recipe of attn_bwd_dkv2_kernel (further down: read its header first) applied to dQ.  An ITEM = 64 queries x the 4 query heads of a kv
// head (wave w = head w), sweeping the 64-key tiles 0 .. its own; a UNIT = (32-key block kb, 32-query block qb) of a tile: 8 S^T / dP^T
// MFMAs (SP), 16 x { fma, exponential, multiply } + 8 packed conversions (SM), 4 dQ^T MFMAs (DQ).  A PERIOD = 12 MFMAs: DQ of unit u-1
// (MFMAs 0-3), SP of unit u+1 (4-11, S first), SM of unit u spread over the 12 gaps; four periods = one tile = one trip of the loop (one basic
// block, one barrier, four LDS-DMA requests per wave, 32 LDS reads: 8 per period, each >= 8 gaps ahead of its first use).  K / V row fragments
// and K transposed fragments of a key block serve both query blocks.  Q / dO operand fragments and the dQ sums live in accumulation
// registers; delta rides in as the C operand of the dP chain (a replicated register set that is never dead), lse as the addend of the
// exponent's fma (which also carries the 1/sqrt(d): the Q fragments stay as loaded).  Only the last tile of an item (its diagonal) needs the
// causal mask: a second, masked loop of one trip behind the first.
//
// With 400 registers per wave a CU holds ONE workgroup, so whatever an item does before and behind its tiles — waiting for its operands,
// converting them, storing dQ — is time the matrix pipe stands still: ~19 000 cycles per item against ~2 350 per tile and 16.5 tiles per item
// when every item was a workgroup (profiles/r04_dq2_stamps.txt).  Hence the workgroups are PERSISTENT: a workgroup walks DQ2_ITEMS = 8 (4, 2 for
// small launches) query blocks of one (batch, kv head) — blocks g, 2W-1-g, 2W+g, 4W-1-g, ... of the S/64, W = S/64/DQ2_ITEMS workgroups per
// pair, every workgroup the same number of tiles — and the Q, dO and O rows of the NEXT item are requested (LDS-DMA, whole 128-B lines, into per-wave images: no barrier)
// while the current item computes; its lse one item ahead into registers; the RoPE table rows for the store into LDS as well.
constexpr int DQ2_RING = 3;
constexpr int DQ2_STAGE = DQ2_RING * 16384;            // per wave: [Q | dO | O][64][64] bf16 images of the item's rows of its head
constexpr int DQ2_ROPE = DQ2_STAGE + 4 * 3 * 8192;     // [64 queries][64] fp32 table rows, 16-B chunks XOR (row & 15)
constexpr int DQ2_LDS = DQ2_ROPE + 64 * 256;           // 160 KiB: all of a CU's LDS

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    bf16x2_ v;
    v[0] = (bf16_t)a;
    v[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, v);
}

// The vector instructions of a dQ unit beside its 12 MFMAs: per pair j of accumulator elements two fmas (kind 0: 2 issue slots), two
// exponentials (kinds 1, 2: 2 slots each), two multiplies (kind 3) and a packed conversion (kind 4: 1 slot): 72 slots, dealt to the 12 gaps
// by their running slot count in an order that never lets an instruction read its predecessor's result: fma j+1, exp j, exp j, multiply
// j-1, conversion j-2.  (Packed fp32 — v_pk_fma_f32, v_pk_mul_f32 — would halve the fma / multiply slots, but beside a running MFMA one
// packed instruction costs ~14 cycles against ~4.5 for a scalar one: tools/micro/mfma_gap.hip, 63 cycles per gap for 4 of them.)
struct Dq2Plan {
    int n, kind[40], pair[40], gap[40];
};
constexpr Dq2Plan dq2_make_plan() {
    Dq2Plan p{};
    int n = 0;
    auto push = [&](int kind, int j) {
        if (j < 0 || j > 7) return;
        p.kind[n] = kind;
        p.pair[n] = j;
        ++n;
    };
    push(0, 0);
    for (int j = 0; j < 10; ++j) {
        push(0, j + 1);
        push(1, j);
        push(2, j);
        push(3, j - 1);
        push(4, j - 2);
    }
    p.n = n;
    int slots = 0;
    for (int i = 0; i < n; ++i) {
        const int g = slots * 12 / 72;
        p.gap[i] = g > 11 ? 11 : g;
        slots += p.kind[i] == 4 ? 1 : 2;
    }
    return p;
}
constexpr Dq2Plan DQ2_PLAN = dq2_make_plan();

// VARLEN (round 5; DQ2_ITEMS = 0): packed rows.  The items come from a host-built PLAN (ssi_attn_plan_build): an item = (row b, 64-query block
// q0 — a multiple of 64 —, document [dstart, dend)); a block that straddles a document boundary is two items.  A workgroup walks the items of
// one GROUP of the plan (groups of equal total work: longest-processing-time assignment on the host, an item's work = its key tiles + its
// fixed cost), heaviest first, for one kv head.  An item sweeps the key tiles dstart / 64 .. q0 / 64 of ITS document (tiles of other
// documents are skipped); masked are its diagonal tile and, when the document does not start on a 64-row boundary, its first tile (keys
// < dstart) — both by the one mask  dstart <= key <= query  in a masked loop of its own in front of / behind the plain loop.  Lanes whose query
// lies outside [dstart, dend) compute on whatever their row holds and store nothing (query = lane: their columns stay their own).  RoPE
// positions are query - dstart (the plan builder checks that input_pos runs 0, 1, 2, ... inside every document).
template <int DQ2_ITEMS, bool VARLEN = false>  // query blocks per workgroup: 8, 4 or 2 (the host takes the largest that fills the chip in whole rounds)
__global__ __launch_bounds__(256, 1) void attn_bwd_dq2_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                              const float* __restrict__ rope, int S, int H, int KV, int W,
                                                              const int4* __restrict__ groups, int group_stride, int table_len) {
    __shared__ __attribute__((aligned(16))) char smem[DQ2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    // workgroup -> ((batch, kv head) pair, group g of its query blocks); an XCD gets whole pairs (their K / V stay in one L2)
    int pair, g;
    const int4* gp = nullptr;  // VARLEN: this workgroup's group of the plan: [0].x = its item count, [1 ..] = the items
    if constexpr (VARLEN) {
        const int id = (int)blockIdx.x;
        pair = id % KV;  // consecutive workgroups = the kv heads of one group, i.e. (KV = 8) kv head = XCD
        g = id / KV;
        gp = groups + (int64_t)g * group_stride;
    } else {
        const int n_pairs = (int)gridDim.x / W, id = (int)blockIdx.x;
        if (n_pairs % 8 == 0) {
            const int ppx = n_pairs / 8, k = id >> 3;
            pair = (id & 7) * ppx + k / W;
            g = k % W;
        } else {
            pair = id / W;
            g = id % W;
        }
    }
    const int n_items = VARLEN ? gp[0].x : DQ2_ITEMS;
    const int kvh = pair % KV, b = VARLEN ? 0 : pair / KV;   // VARLEN: the row is the item's
    const int head = kvh * 4 + wave;
    const int64_t row0 = (int64_t)b * S, ldo = (int64_t)H * HD;
    const bf16_t* kbase = qkv + row0 * ld + (int64_t)H * HD + (int64_t)kvh * HD;
    // item i (heaviest first) -> query block: the pairs (2W-1-g, g) of the four 2W-blocks, from the top
    auto item_block = [&](int i) __attribute__((always_inline)) {
        const int u = (DQ2_ITEMS / 2 - 1) - (i >> 1);
        return u * 2 * W + ((i & 1) ? g : 2 * W - 1 - g);
    };
    // item i as (row offset of its batch row, query block, first key tile, document)
    struct Item { int64_t r0; int jq, t0, ds, de; };
    auto item_at = [&](int i) __attribute__((always_inline)) {
        Item it;
        if constexpr (VARLEN) {
            const int4 v = gp[1 + i];  // (uniform address: a scalar load)
            it.r0 = (int64_t)v.x * S, it.jq = v.y >> 6, it.t0 = v.z >> 6, it.ds = v.z, it.de = v.w;
        } else {
            it.r0 = row0, it.jq = item_block(i), it.t0 = 0, it.ds = 0, it.de = S;
        }
        return it;
    };

    KvTileDma<SWZ_DUAL, SWZ_ROW> kvdma;
    kvdma.init(kbase, ld, KV * HD, smem, wave, lane);
    // requests of a [64][64] image by one wave (see RowTileDma): per-lane source offsets for rows of stride ld (Q) and ldo (dO, O)
    unsigned vq[2], vo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int chunk = (lane & 7) ^ (lane >> 4) ^ (4 * par);
        vq[par] = (unsigned)(((lane >> 3) * ld + chunk * 8) * 2);
        vo[par] = (unsigned)(((lane >> 3) * ldo + chunk * 8) * 2);
    }
    const char* stage = smem + DQ2_STAGE + wave * (3 * 8192);
    const unsigned stage_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)stage);
    // Q, dO and O rows of item block jq of this wave's head: 24 requests
    auto request_stage = [&](int64_t r0_, int jq) __attribute__((always_inline)) {
        const int64_t r = r0_ + jq * 64;
        const u32x4 rq = buffer_rsrc(qkv + r * ld + (int64_t)head * HD), rd = buffer_rsrc(dout + r * ldo + (int64_t)head * HD),
                    ro = buffer_rsrc(out + r * ldo + (int64_t)head * HD);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dma16(stage_lds + i * 1024, vq[i & 1], rq, (unsigned)(i * 8 * ld * 2));
            dma16(stage_lds + 8192 + i * 1024, vo[i & 1], rd, (unsigned)(i * 8 * ldo * 2));
            dma16(stage_lds + 16384 + i * 1024, vo[i & 1], ro, (unsigned)(i * 8 * ldo * 2));
        }
    };
    // RoPE table rows q0 .. q0 + 63 (256 B each) of an item: 16 pieces of 4 rows, wave w the pieces w, w + 4, w + 8, w + 12; the 16-B chunk c
    // of row r lies at chunk c ^ (r & 15).  Without a table the requests read the head of qkv instead (and the store ignores them).
    const unsigned rope_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)(smem + DQ2_ROPE));
    const unsigned vrope = (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ ((4 * wave + (lane >> 4)) & 15)) * 16));
    const u32x4 rope_rs = buffer_rsrc(rope ? (const void*)rope : (const void*)qkv);
    auto lse_of = [&](int64_t r0_, int jq, int qb) __attribute__((always_inline)) {  // lse is [B][H][S]: r0_ = b S
        return lse[(r0_ * H + (int64_t)head * S) + jq * 64 + 32 * qb + (lane & 31)];
    };

    // ---- per-tile register state ------------------------------------------------------------------------------------------------------------
    bf16x8 qf[2][4], dof[2][4];    // B operands: lane = query q0 + 32 qb + (l & 31), d = 16 ks + 8 h + j
    f32x16 pdl[2];                 // -delta of the lane's query in all 16 registers: C operand of the dP^T chain
    float nlq[2];                  // -lse * log2(e): p = exp2(S^T * log2(e) / 8 + nlq)
    int qg[2];
    f32x16 dq[2][2];
    f32x16 sacc[2], pacc[2];       // [unit parity]: S^T and dP'^T of the unit in flight
    bf16x8 rowK[2][4], rowV[2][4]; // [kb]: K / V row fragments (A operands of the S^T / dP^T products)
    s16x4 ktrh[4][2][2];           // [k-step s][db][half]: K transposed fragments (A operands of the dQ^T products); s = 2 kb, 2 kb + 1
    u32x4 dsu[2][2];               // [unit parity][s2]: dS^T of a unit as bf16 operand fragments
    unsigned ring_cur, ring_nxt, ring_n2;  // byte offsets of the slots of tiles t, t+1, t+2
    constexpr float SCALE2 = LOG2E * 0.125f;  // log2(e) / sqrt(d)

    auto tr_frag = [&](const s16x4 (&hv)[2]) __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(hv[0], hv[1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // read i (0..7) of the row fragments of key block kb of the tile at `slot`: 0-3 K, 4-7 V
    auto read_rows = [&](unsigned slot, int kb, int i) __attribute__((always_inline)) {
        const char* kt = smem + slot;
        if (i < 4) rowK[kb][i] = frag_row<SWZ_DUAL>(kt, kb * 32, i, lane);
        else rowV[kb][i - 4] = frag_row<SWZ_ROW>(kt + 8192, kb * 32, i - 4, lane);
    };
    // read i (0..7) of the transposed fragments of key block kb: k-step 2 kb + (i >> 2), db (i >> 1) & 1, half i & 1 — the order of their use
    auto read_tr = [&](unsigned slot, int kb, int i) __attribute__((always_inline)) {
        const int sI = 2 * kb + (i >> 2), db = (i >> 1) & 1;
        ktrh[sI][db][i & 1] = frag_tr_half<SWZ_DUAL>(smem + slot, sI * 16, db * 32, lane, i & 1);
    };
    // S^T / dP^T product m (0..7) of unit (kb, qb) into register set `par`: the S chain first (see sp_mfma of attn_bwd_dkv2_kernel)
    auto sp_mfma = [&](int par, int kb, int qb, int m) __attribute__((always_inline)) {
        const int ks = m & 3;
        if (m == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(sacc[par]) : "v"(rowK[kb][0]), "a"(qf[qb][0]));
        else if (m == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(pacc[par]) : "v"(rowV[kb][0]), "a"(dof[qb][0]), "v"(pdl[qb]));
        else if (m > 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(pacc[par]) : "v"(rowV[kb][ks]), "a"(dof[qb][ks]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[par]) : "v"(rowK[kb][ks]), "a"(qf[qb][ks]));
    };
    // dQ^T product i (0..3) of unit (kb, qb) whose dS^T sits in dsu[par]: k-step s2 = i >> 1 of the key block, d block i & 1
    auto dq_mfma = [&](int par, int kb, int qb, int i) __attribute__((always_inline)) {
        const int s2 = i >> 1, db = i & 1;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dq[qb][db]) : "v"(tr_frag(ktrh[2 * kb + s2][db])), "v"(dsu[par][s2]));
    };
    // exponentials of unit (kb, qb) in register set `par`, gap g of 12: the instructions DQ2_PLAN puts there
    f32x2 ev[8], dsv[8];
    float pv[16];
    // the causal mask inside a DIAGONAL 32 x 32 block (kb == qb of the diagonal tile): key row rowmap(r, h) against query column l & 31 —
    // the same 16 lane masks for every item.  Of the other two blocks of that tile, (kb 0, qb 1) is all visible and (kb 1, qb 0) all masked.
    bool beyond[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) beyond[r] = rowmap(r, h) > (lane & 31);
    int ds_item = 0;  // VARLEN: first key of the item's document
    auto sm_gap = [&](auto edge_c, int par, int kb, int qb, int k0, int gap) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        if (!VARLEN && EDGE && kb == 1 && qb == 0) {  // nothing visible: dS^T = 0
            if (gap == 0) dsu[par][0] = dsu[par][1] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
#pragma unroll
        for (int i = 0; i < DQ2_PLAN.n; ++i) {
            if (DQ2_PLAN.gap[i] != gap) continue;
            const int j = DQ2_PLAN.pair[i], kind = DQ2_PLAN.kind[i];
            if (kind == 0) {
                ev[j][0] = fmaf(sacc[par][2 * j], SCALE2, nlq[qb]);
                ev[j][1] = fmaf(sacc[par][2 * j + 1], SCALE2, nlq[qb]);
            } else if (kind == 1 || kind == 2) {
                const int r = 2 * j + kind - 1;
                float p = __builtin_amdgcn_exp2f(ev[j][kind - 1]);
                if constexpr (VARLEN) {  // a masked tile of a packed row (the document's first or the item's diagonal): dstart <= key <= query
                    if (EDGE) {
                        const int key = k0 + 32 * kb + rowmap(r, h);
                        if (key > qg[qb] || key < ds_item) p = 0.f;
                    }
                } else if (EDGE && kb == qb && beyond[r]) p = 0.f;  // keys beyond the query contribute nothing
                pv[r] = p;
            } else if (kind == 3) {
                dsv[j][0] = pv[2 * j] * pacc[par][2 * j];  // dS^T (the 1/sqrt(d) factor is applied once at the end)
                dsv[j][1] = pv[2 * j + 1] * pacc[par][2 * j + 1];
            } else {
                dsu[par][j >> 2][j & 3] = pack_bf16(dsv[j][0], dsv[j][1]);
            }
        }
    };
    // one period: SM of unit `cur`, DQ of the unit before it, SP of the unit behind it; 8 LDS reads; optionally the ring barrier in front and
    // four LDS-DMA requests behind.  Units are (kb, qb, register set); k0 = first key of `cur`'s tile.
    struct Unit { int kb, qb, par; };
    auto period = [&](auto edge_c, auto sync_c, auto sp_c, Unit prev, Unit cur, Unit next, int k0, auto reads, auto tail, auto landed) __attribute__((always_inline)) {
        if (decltype(sync_c)::value) {
            // own requests of tile t+1 have landed (those of t+2 stay in flight) ... and everybody's; every wave is done with tile t
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            ring_barrier();
        }
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m < 4) dq_mfma(prev.par, prev.kb, prev.qb, m);
            else if (decltype(sp_c)::value) sp_mfma(next.par, next.kb, next.qb, m - 4);
            __builtin_amdgcn_sched_barrier(0);  // the MFMA first: the first multiply of a period reads the dP^T chain finished one MFMA ago
            sm_gap(edge_c, cur.par, cur.kb, cur.qb, k0, m);
            reads(m);
            tail(m);
            if (m == 11) landed();  // the period's 8 LDS reads (issued in gaps 0-7): ONE wait here instead of hipcc's one per first use
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // "these registers have been read": hipcc puts its s_waitcnt lgkmcnt in front, and none at the uses behind
    auto rows_landed = [&](int kb) __attribute__((always_inline)) {
        asm volatile("" ::"v"(rowK[kb][0]), "v"(rowK[kb][1]), "v"(rowK[kb][2]), "v"(rowK[kb][3]), "v"(rowV[kb][0]), "v"(rowV[kb][1]), "v"(rowV[kb][2]),
                     "v"(rowV[kb][3]));
    };
    auto tr_landed = [&](int kb) __attribute__((always_inline)) {
        asm volatile("" ::"v"(ktrh[2 * kb][0][0]), "v"(ktrh[2 * kb][0][1]), "v"(ktrh[2 * kb][1][0]), "v"(ktrh[2 * kb][1][1]), "v"(ktrh[2 * kb + 1][0][0]),
                     "v"(ktrh[2 * kb + 1][0][1]), "v"(ktrh[2 * kb + 1][1][0]), "v"(ktrh[2 * kb + 1][1][1]));
    };
    auto none = [&](int) __attribute__((always_inline)) {};
    using T_ = std::true_type;
    using F_ = std::false_type;
    const Unit U0{0, 0, 0}, U1{0, 1, 1}, U2{1, 0, 0}, U3{1, 1, 1};
    // hipcc does not know the asm statements above to be MFMAs.  Where it moves their registers itself — at the ends of the loops below — its
    // copies get no wait states: a copy reading a result still in the pipe, or an MFMA reading an accumulator register written just before it
    // (that one cost element 0 of a dQ block).  MFMA_DRAIN / MFMA_GUARD at every such place.
#define MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
#define MFMA_GUARD() asm volatile("s_nop 7" ::: "memory")

    // -DDQ2_STAMP (debug build, tools/attn_dq_check.py stamps): cycles of wave 0 per phase of an item, summed over the workgroup's items, left
    // in the first floats of its LAST item's first dq row (the lightest block of the group) together with the 100 MHz clock's count
#ifdef DQ2_STAMP
    unsigned long long stq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stq_last = __builtin_readcyclecounter();
    const unsigned long long stq_begin = stq_last, stq_rt0 = __builtin_amdgcn_s_memrealtime();
#define STAMPQ(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); stq_acc[i] += now_ - stq_last; stq_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMPQ(i) {}
#endif
    // ---- before the first item: its rows, its lse -------------------------------------------------------------------------------------------
    // the first three key tiles of an item of nt tiles; behind the last tile the last tile is requested again (into a slot nobody reads), so
    // that the counted vmcnt waits hold without a tail case and a trip has no branch
    auto request_first_tiles = [&](int t0, int nt) __attribute__((always_inline)) {  // tiles t0 .. nt - 1
        kvdma.tile(t0, 0);
        kvdma.tile(t0 + 1 < nt ? t0 + 1 : t0, 16384);
        kvdma.tile(t0 + 2 < nt ? t0 + 2 : nt - 1, 32768);
    };
    // K / V rows of the item's batch row (VARLEN: items of one group may lie in different rows)
    auto kv_rows = [&](int64_t r0_) __attribute__((always_inline)) {
        if constexpr (VARLEN) kvdma.rs = buffer_rsrc(qkv + r0_ * ld + (int64_t)H * HD + (int64_t)kvh * HD);
    };
    float lqn[2];
    {
        const Item i0 = item_at(0);
        kv_rows(i0.r0);
        request_first_tiles(i0.t0, i0.jq + 1);
        request_stage(i0.r0, i0.jq);
        lqn[0] = lse_of(i0.r0, i0.jq, 0);
        lqn[1] = lse_of(i0.r0, i0.jq, 1);
    }

    for (int it = 0; it < n_items; ++it) {
        const Item icur = item_at(it), inxt = item_at(it + 1 < n_items ? it + 1 : it);
        const int jq = icur.jq, jn = inxt.jq;
        const int q0 = jq * 64, nt = jq + 1;  // key tiles t0 .. jq; the last one holds the diagonal
        const int64_t rw0 = icur.r0;          // row offset of the item's batch row
        if constexpr (VARLEN) ds_item = icur.ds;
        const float lq0 = lqn[0], lq1 = lqn[1];
        // everything this wave has asked for is there: the item's rows (asked for an item ago), its first three tiles (asked for in front of
        // the store of the item before), that store
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(lq0), "v"(lq1) : "memory");
        STAMPQ(0)
        bf16x8 oraw[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[qb][ks] = frag_row<SWZ_ROW>(stage, 32 * qb, ks, lane);
                dof[qb][ks] = frag_row<SWZ_ROW>(stage + 8192, 32 * qb, ks, lane);
                oraw[qb][ks] = frag_row<SWZ_ROW>(stage + 16384, 32 * qb, ks, lane);
            }
        // in registers: the images are free for the next item's rows (this wave's own images: no barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(oraw[1][3]), "v"(dof[1][3]), "v"(qf[1][3]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        lqn[0] = lse_of(inxt.r0, jn, 0);
        lqn[1] = lse_of(inxt.r0, jn, 1);
        // 28 requests — the table rows of THIS item (nobody reads the old ones any more: barrier at the end of the item before) and the next
        // item's rows — dealt over the 8 steps of the delta sums: back to back, four waves' requests queue up in front of the CU's one
        // address unit (~150 cycles each where a request inside the tile loop costs 40)
        const int64_t rn = inxt.r0 + jn * 64;
        const u32x4 rq = buffer_rsrc(qkv + rn * ld + (int64_t)head * HD), rd = buffer_rsrc(dout + rn * ldo + (int64_t)head * HD),
                    ro = buffer_rsrc(out + rn * ldo + (int64_t)head * HD);
        auto request = [&](int i) __attribute__((always_inline)) {
            if (i < 4) {
                if constexpr (VARLEN) {  // table row of query q = its position q - dstart, kept inside the table for the lanes outside the document
                    const int pr = q0 - icur.ds + 4 * (wave + 4 * i) + (lane >> 4);
                    const int prc = pr < 0 ? 0 : (pr < table_len ? pr : table_len - 1);
                    dma16(rope_lds + (wave + 4 * i) * 1024, (rope ? (unsigned)prc * 256u : 0u) + (vrope & 255u), rope_rs, 0u);
                } else
                dma16(rope_lds + (wave + 4 * i) * 1024, vrope, rope_rs, (unsigned)((rope ? q0 * 256 : 0) + (wave + 4 * i) * 1024));
            } else {
                const int j = (i - 4) / 3, which = (i - 4) % 3;
                if (which == 0) dma16(stage_lds + j * 1024, vq[j & 1], rq, (unsigned)(j * 8 * ld * 2));
                else if (which == 1) dma16(stage_lds + 8192 + j * 1024, vo[j & 1], rd, (unsigned)(j * 8 * ldo * 2));
                else dma16(stage_lds + 16384 + j * 1024, vo[j & 1], ro, (unsigned)(j * 8 * ldo * 2));
            }
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            qg[qb] = q0 + 32 * qb + (lane & 31);
            float dl = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)oraw[qb][ks][e] * (float)dof[qb][ks][e];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = (4 * qb + ks) * 28 / 8; i < (4 * qb + ks + 1) * 28 / 8; ++i) request(i);
                __builtin_amdgcn_sched_barrier(0);
            }
            dl += __shfl_xor(dl, 32, 64);
            // for the dK / dV kernel, which runs after this one (every (row, head) belongs to exactly one wave)
            if (h == 0) delta[(rw0 * H + (int64_t)head * S) + qg[qb]] = dl;  // (a straddled block's two items write the same values)
            nlq[qb] = -(qb ? lq1 : lq0) * LOG2E;
#pragma unroll
            for (int r = 0; r < 16; ++r) pdl[qb][r] = -dl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {  // from here on the fragments LIVE in accumulation registers (see attn_bwd_dkv2_kernel)
                asm volatile("" : "=a"(qf[qb][ks]) : "0"(qf[qb][ks]));
                asm volatile("" : "=a"(dof[qb][ks]) : "0"(dof[qb][ks]));
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[qb][db][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) dsu[i][k2] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int sI = 0; sI < 4; ++sI)
#pragma unroll
            for (int db = 0; db < 2; ++db) ktrh[sI][db][0] = ktrh[sI][db][1] = s16x4{0, 0, 0, 0};  // the first period's dQ products add 0 * 0
        ring_cur = 0, ring_nxt = 16384, ring_n2 = 32768;
        STAMPQ(1)

        // ---- tile 0 is there for everybody (each wave waited for its own pieces above): row fragments of its first key block, transposed
        // fragments of the same, S^T / dP^T of unit 0
        ring_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) read_rows(0, 0, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) read_tr(0, 0, i);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_GUARD();
#pragma unroll
        for (int m = 0; m < 8; ++m) sp_mfma(0, 0, 0, m);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        STAMPQ(2)
        int t = icur.t0;
        if constexpr (VARLEN) {
            // ---- packed rows: the document's first tile when it holds keys of the document before (dstart off the 64-row grid) and is not
            // the diagonal tile: the masked form of a full trip.  A loop of zero or one trip (see below why a loop)
            const int t_head = ((icur.ds & 63) && t + 1 < nt) ? t + 1 : t;
            for (; t < t_head; ++t) {
                const int k0 = t * 64;
                const int t3 = t + 3 < nt ? t + 3 : nt - 1;
                MFMA_GUARD();
                period(T_{}, F_{}, T_{}, U3, U0, U1, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_rows(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { rows_landed(1); });
                period(T_{}, F_{}, T_{}, U0, U1, U2, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_tr(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { tr_landed(1); });
                period(T_{}, T_{}, T_{}, U1, U2, U3, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_rows(ring_nxt, 0, m); },
                       [&](int m) __attribute__((always_inline)) { if (m >= 8) kvdma.piece(t3, ring_cur, m - 8); }, [&]() __attribute__((always_inline)) { rows_landed(0); });
                period(T_{}, F_{}, T_{}, U2, U3, U0, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_tr(ring_nxt, 0, m); }, none, [&]() __attribute__((always_inline)) { tr_landed(0); });
                const unsigned c = ring_cur;
                ring_cur = ring_nxt;
                ring_nxt = ring_n2;
                ring_n2 = c;
            }
            MFMA_DRAIN();
        }
        // ---- the unmasked tiles: t0 .. nt - 2 -----------------------------------------------------------------------------------------------
        for (; t + 1 < nt; ++t) {
            const int k0 = t * 64;
            const int t3 = t + 3 < nt ? t + 3 : nt - 1;  // (a select)
            MFMA_GUARD();
            period(F_{}, F_{}, T_{}, U3, U0, U1, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_rows(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { rows_landed(1); });
            period(F_{}, F_{}, T_{}, U0, U1, U2, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_tr(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { tr_landed(1); });
            period(F_{}, T_{}, T_{}, U1, U2, U3, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_rows(ring_nxt, 0, m); },
                   [&](int m) __attribute__((always_inline)) { if (m >= 8) kvdma.piece(t3, ring_cur, m - 8); }, [&]() __attribute__((always_inline)) { rows_landed(0); });
            period(F_{}, F_{}, T_{}, U2, U3, U0, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_tr(ring_nxt, 0, m); }, none, [&]() __attribute__((always_inline)) { tr_landed(0); });
            const unsigned c = ring_cur;
            ring_cur = ring_nxt;
            ring_nxt = ring_n2;
            ring_n2 = c;
        }
        MFMA_DRAIN();
        STAMPQ(3)
        // ---- the diagonal tile, masked; no tile behind it.  Written as a second LOOP (of one trip): straight-line code here would be entered
        // from the loop above or around it, the accumulation registers of the two ways in would meet at its entry, and hipcc moves them there.
        // Two loops in sequence keep their registers (as in attn_bwd_dkv2_kernel).
        for (; t < nt; ++t) {
            const int k0 = t * 64;
            MFMA_GUARD();
            period(T_{}, F_{}, T_{}, U3, U0, U1, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_rows(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { rows_landed(1); });
            period(T_{}, F_{}, F_{}, U0, U1, U2, k0, [&](int m) __attribute__((always_inline)) { if (m < 8) read_tr(ring_cur, 1, m); }, none, [&]() __attribute__((always_inline)) { tr_landed(1); });
            period(T_{}, F_{}, T_{}, U1, U2, U3, k0, none, none, [&]() __attribute__((always_inline)) {});
            period(T_{}, F_{}, F_{}, U2, U3, U0, k0, none, none, [&]() __attribute__((always_inline)) {});
        }
        MFMA_DRAIN();
#pragma unroll
        for (int i = 0; i < 4; ++i) dq_mfma(U3.par, U3.kb, U3.qb, i);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_DRAIN();
        STAMPQ(4)
        // ---- the store.  Every request of this wave has landed (table rows; the next item's rows; the tiles asked for beyond the last) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_barrier();  // ... and every other wave's pieces of the table rows
        f32x4 rcs[2][2][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q = 32 * qb + (lane & 31);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    rcs[qb][db][gg] = *reinterpret_cast<const f32x4*>(smem + DQ2_ROPE + q * 256 + (((8 * db + 2 * gg + h) ^ (q & 15)) * 16));
        }
        // everybody has its table rows in registers and is done with the ring: the next item's requests may overwrite both
        ring_barrier();
        kv_rows(inxt.r0);
        request_first_tiles(inxt.t0, jn + 1);  // (behind the last item: its own once more — waited for at the end of the kernel)
        STAMPQ(5)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            bf16_t* drow = dqkv + (rw0 + qg[qb]) * ld + (int64_t)head * HD;
            const bool mine = !VARLEN || (qg[qb] >= icur.ds && qg[qb] < icur.de);  // packed rows: queries of other documents are other items'
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(dq[qb][db][4 * gg + e] * 0.125f);
                    // rope != NULL: the gradient leaves in pre-RoPE space (backward of the rotation fused here, saves a pass over dqkv)
                    if (rope) v = unrope4(v, rcs[qb][db][gg]);
                    if (mine) *reinterpret_cast<bf16x4*>(drow + db * 32 + 8 * gg + 4 * h) = v;
                }
        }
        STAMPQ(6)
#ifdef DQ2_STAMP
        if (it == n_items - 1 && wave == 0) {  // DEBUG BUILD ONLY: overwrites the first floats of the item's first dq row
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                float* dbg = reinterpret_cast<float*>(dqkv + (rw0 + q0) * ld + (int64_t)head * HD);
                for (int i = 0; i < 7; ++i) dbg[i] = (float)stq_acc[i];
                dbg[7] = (float)(__builtin_readcyclecounter() - stq_begin);
                dbg[8] = (float)(__builtin_amdgcn_s_memrealtime() - stq_rt0);
                dbg[9] = (float)g;
            }
        }
#endif
    }
    // The last item asked for its own rows once more (landed before its store) — and for its first three tiles once more, in front of its
    // store: nothing of this workgroup may be in flight towards LDS when it ends (the LDS goes to the next workgroup on this CU).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =====================================================================================================================
// backward: dK, dV
// =====================================================================================================================
// Workgroup = (b, kv head, 128-key group); wave w owns keys [key0 + 32 w, +32) and keeps their dK^T / dV^T in accumulators
// while the workgroup sweeps, for each of the `rep` query heads of the kv head in turn, the Q / dO tiles (32 queries) from
// the diagonal to the end of the sequence.  The tile of a step is staged ONCE for all four waves (LDS-DMA, double
// buffered, one step ahead), so Q and dO cross the L2 -> CU path once per 128 keys instead of once per 32, and the sum
// over the query heads of the group happens in registers: no cross-wave reduction, no partial buffers, one writer per
// output element.
// HSPLIT (round 4): a workgroup sweeps `heads_per_wg` of the group's query heads instead of all `rep` of them; its dK / dV sums leave as fp32
// partial rows in `partial` ([slot = head / heads_per_wg][B * S][KV][dK 64 | dV 64]) and attn_dkv_head_reduce_kernel adds the slots in a fixed order.  For
// launches whose workgroups cannot fill the chip: the longest workgroup IS the launch (B = 2, S = 2048: 256 workgroups, the heaviest with
// 256 steps: 176 us per layer where 65 is the launch's share of the chip), and a workgroup's steps are queries x heads.
template <bool HSPLIT>
__global__ __launch_bounds__(256, DKV_WAVES) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, int64_t ld,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, bf16_t* __restrict__ dqkv,
                                                           const int32_t* __restrict__ doc_end, const float* __restrict__ rope,
                                                           const int32_t* __restrict__ positions, int S, int H, int KV,
                                                           float* __restrict__ partial, int heads_per_wg) {
    // ring of RING step buffers: [Q tile 4 KiB | dO tile 4 KiB | lse 128 B | delta 128 B]; requests run RING-1 steps ahead
    constexpr int SB = 8192 + 256;
    constexpr int RING = DKV_RING;
    __shared__ __attribute__((aligned(16))) char smem[RING * SB];
    TRACE_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rep_all = H / KV;
    const int rep = HSPLIT ? heads_per_wg : rep_all;  // query heads this workgroup sweeps
    const int n_slots = rep_all / rep;                 // workgroups (and partial rows) per key group
    const int ngrp = S / 128;
    int kgrp, pair_, head0 = 0, slot = 0;  // low key groups (most work) are dispatched first
    if (HSPLIT) {
        int r;
        block_to_work(ngrp * n_slots, (int)(gridDim.x / (ngrp * n_slots)), r, pair_);
        kgrp = r / n_slots;
        slot = r % n_slots;
        head0 = slot * rep;
    } else {
        block_to_work(ngrp, (int)(gridDim.x / ngrp), kgrp, pair_);
    }
    const int kvh = pair_ % KV;
    const int b = pair_ / KV;
    const int h = lane >> 5;
    const int64_t row0 = (int64_t)b * S;
    const int64_t ldo = (int64_t)H * HD;
    const int key0 = kgrp * 128 + wave * 32;
    const int kg = key0 + (lane & 31);

    // -K * 2^-3 and -V as B operands (lane holds row key0 + (l & 31), d = 16 ks + 8 h + j).  With the operands negated and
    // +lse / +delta as the initial accumulators, the chains deliver  lse - S  and  delta - dP,  so that
    // P = exp2(-(lse - S) log2 e) needs one multiply (by a negative constant) and -dS = P (delta - dP) one more: no
    // subtractions, no zero-initialisation.  dK accumulates with the opposite sign and is flipped by the final scale.
    bf16x8 kf[4], vf[4];
    {
        const bf16_t* krow = qkv + (row0 + kg) * ld + (int64_t)H * HD + (int64_t)kvh * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = scale_frag(*reinterpret_cast<const bf16x8*>(krow + 16 * ks), -0.125f);
            vf[ks] = scale_frag(*reinterpret_cast<const bf16x8*>(krow + (int64_t)KV * HD + 16 * ks), -1.0f);
        }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

    // packed rows: key k is seen by the queries k <= q < doc_end[k] (doc_end is non-decreasing along a row): the tile loop stops
    // at the end of the document of the group's last key, a tile needs the document mask iff it reaches past the end of the
    // document of the wave's first key, and is dead for the wave from the end of the document of its last key on.
    const int de = doc_end ? doc_end[row0 + kg] : S;                      // this lane's key
    const int de_lo = doc_end ? doc_end[row0 + key0] : S;                 // first key of the wave
    const int de_hi = doc_end ? doc_end[row0 + key0 + 31] : S;            // last key of the wave
    const int q_end = doc_end ? doc_end[row0 + kgrp * 128 + 127] : S;     // last key of the group
    const int qb_first = kgrp * 4;                       // first 32-query tile that sees any key of the group
    const int per_head = (q_end + 31) / 32 - qb_first;   // tiles per query head
    const int n_steps = per_head * rep;
    // step -> (head of the group, query tile); each wave moves one 1-KiB piece of Q and one of dO per step.  Steps are issued in
    // order, so (head, tile) and the three source addresses advance incrementally: the per-step `step / per_head`, `step % per_head` and
    // 64-bit address arithmetic cost 67 scalar instructions per step and wave before (SQ_INSTS_SALU), a fifth of the step's issue
    int iss_qt = 0;                                  // query tile of the next request inside its head
    const int irow = wave * 8 + (lane >> 3), ichunk = (lane & 7) ^ swz<SWZ_DUAL>(wave * 8 + (lane >> 3));
    const u32x4 rs_q = buffer_rsrc(qkv + row0 * ld + (int64_t)(kvh * rep_all + head0) * HD);    // Q columns of the group's first head, this batch
    const u32x4 rs_do = buffer_rsrc(dout + row0 * ldo + (int64_t)(kvh * rep_all + head0) * HD);
    const unsigned voff_q = (unsigned)((irow * ld + ichunk * 8) * 2), voff_do = (unsigned)((irow * ldo + ichunk * 8) * 2);
    unsigned soff_q = (unsigned)(qb_first * 32 * ld * 2), soff_do = (unsigned)(qb_first * 32 * ldo * 2);  // scalar, advanced per request
    const float* iss_rc = (lane < 32 ? lse : delta) + ((int64_t)b * H + kvh * rep_all + head0) * S + qb_first * 32 + (lane & 31);
    const unsigned lds_piece = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + (unsigned)wave * 1024u);
    const unsigned lds_rc = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + 8192u);
    auto issue = [&](int step) {
        const unsigned buf = (unsigned)(step % RING) * SB;
        dma16(lds_piece + buf, voff_q, rs_q, soff_q);
        dma16(lds_piece + buf + 4096, voff_do, rs_do, soff_do);
        // row constants of the tile: lanes 0-31 fetch lse[q0 + l], lanes 32-63 delta[q0 + l - 32] (every wave issues the same
        // 256-B request so that all waves count 3 requests per step); two arrays, hence per-lane 64-bit addresses
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_rc + buf), "v"(iss_rc) : "memory");
        if (++iss_qt == per_head) {  // next head of the group: back to the first query tile, one head further
            iss_qt = 0;
            soff_q += (unsigned)(HD * 2) - (unsigned)((per_head - 1) * 32 * ld * 2);
            soff_do += (unsigned)(HD * 2) - (unsigned)((per_head - 1) * 32 * ldo * 2);
            iss_rc += (int64_t)S - (int64_t)(per_head - 1) * 32;
        } else {
            soff_q += (unsigned)(32 * ld * 2);
            soff_do += (unsigned)(32 * ldo * 2);
            iss_rc += 32;
        }
    };
    int cur_qt = 0;  // query tile of the step being computed (steps run in order too)
    // -DDKV_STAMP (debug build, tools/dkv_stamps.py): cycle totals of wave 0 per phase of a step, left in the workgroup's first dq row
#ifdef DKV_STAMP
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_readcyclecounter();
    const unsigned long long st_begin = st_last;
    int st_steps = 0;
#define STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define STAMP(i)
#endif
    // one step on ring buffer BUF (compile-time, so every LDS address is a hoisted per-lane base + an immediate)
    auto do_step = [&](int step, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        const int q0 = (qb_first + cur_qt) * 32;
        if (++cur_qt == per_head) cur_qt = 0;
        const char* qt = smem + BUF * SB;
        const char* dt = qt + 4096;
        const float* rcs = reinterpret_cast<const float*>(qt + 8192);
        if constexpr (BUF % 2 == 0) {
            // one barrier per TWO steps: own requests of this step and the next have landed (later ones may stay in flight) ...
            if (step + RING - 3 < n_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (RING - 4)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail: fewer requests are in flight than the constant assumes
            ring_barrier();  // ... and everybody else's; the buffers of steps -1 and -2 are free again
            if (step + RING - 2 < n_steps) issue(step + RING - 2);
            if (step + RING - 1 < n_steps) issue(step + RING - 1);
        }
        STAMP(0)  // wait + barrier + the two requests the barrier made room for
        if (q0 + 31 < key0 || q0 >= de_hi) return;  // wave-uniform: no query of the tile sees any key of this wave
#ifdef DKV_STAMP
        ++st_steps;
#endif
        f32x16 sacc, pacc;  // rows = queries q0 + rowmap(r, h): row constants come in runs of 4
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(rcs + 4 * h + 8 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(rcs + 32 + 4 * h + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sacc[4 * g + e] = l4[e]; pacc[4 * g + e] = d4[e]; }
        }
        // fragment reads ahead of the products that use them (see attn_fwd_kernel)
        bf16x8 qfr[4], dfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qfr[ks] = frag_row<SWZ_DUAL>(qt, 0, ks, lane);
            dfr[ks] = frag_row<SWZ_DUAL>(dt, 0, ks, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef DKV_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        STAMP(1)  // row constants + fragment reads landed
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], sacc, 0, 0, 0);
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr[ks], vf[ks], pacc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2)  // S / dP MFMAs issued
        bf16x8 dtr[2][2], qtr[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dtr[s2][db] = frag_tr<SWZ_DUAL>(dt, s2 * 16, db * 32, lane);
                qtr[s2][db] = frag_tr<SWZ_DUAL>(qt, s2 * 16, db * 32, lane);
            }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3)  // transposed reads issued
        if (q0 < key0 + 32 || q0 + 31 >= de_lo) {  // edge tile: keys beyond the query or of an earlier document contribute nothing
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(sacc[r] * -LOG2E);
                const int q = q0 + rowmap(r, h);
                if (kg > q || q >= de) p = 0.f;
                sacc[r] = p;
                pacc[r] *= p;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(sacc[r] * -LOG2E);
                sacc[r] = p;
                pacc[r] *= p;
            }
        }
        STAMP(4)  // exponentials (includes waiting for S / dP)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = acc_frag(sacc, s2), dsf = acc_frag(pacc, s2);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dtr[s2][db], pf, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[s2][db], dsf, dk[db], 0, 0, 0);
            }
        }
        STAMP(5)  // conversions + dV / dK MFMAs issued
    };
#pragma unroll
    for (int i = 0; i < RING - 2; ++i)
        if (i < n_steps) issue(i);
    static_assert(RING % 2 == 0 && RING >= 4 && RING <= 10, "the barrier cadence (one per two steps) needs an even ring");
    TRACE_LOOP_BEGIN();
    for (int step = 0; step < n_steps; step += RING) {
        do_step(step, std::integral_constant<int, 0>{});
        if (step + 1 < n_steps) do_step(step + 1, std::integral_constant<int, 1>{});
        if (step + 2 < n_steps) do_step(step + 2, std::integral_constant<int, 2>{});
        if (step + 3 < n_steps) do_step(step + 3, std::integral_constant<int, 3>{});
        if constexpr (RING > 4) {
            if (step + 4 < n_steps) do_step(step + 4, std::integral_constant<int, 4>{});
            if (step + 5 < n_steps) do_step(step + 5, std::integral_constant<int, 5>{});
        }
        if constexpr (RING > 6) {
            if (step + 6 < n_steps) do_step(step + 6, std::integral_constant<int, 6>{});
            if (step + 7 < n_steps) do_step(step + 7, std::integral_constant<int, 7>{});
        }
        if constexpr (RING > 8) {
            if (step + 8 < n_steps) do_step(step + 8, std::integral_constant<int, 8>{});
            if (step + 9 < n_steps) do_step(step + 9, std::integral_constant<int, 9>{});
        }
    }
    TRACE_LOOP_END();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see attn_fwd_kernel)
#ifdef DKV_STAMP
    if (lane == 0 && wave == 0) {  // DEBUG BUILD ONLY: overwrites the first floats of the workgroup's first dq row
        float* dbg = reinterpret_cast<float*>(dqkv + (row0 + kgrp * 128) * ld);
        for (int i = 0; i < 6; ++i) dbg[i] = (float)st_acc[i];
        dbg[6] = (float)(__builtin_readcyclecounter() - st_begin);
        dbg[7] = (float)st_steps;
        dbg[8] = (float)n_steps;
    }
#endif
    if constexpr (HSPLIT) {  // raw fp32 sums of this head: [head][row][kv head][dK 64 | dV 64]; scale, RoPE backward and rounding happen after the heads are added
        const int64_t t_rows = (int64_t)(gridDim.x / (ngrp * n_slots)) / KV * S;  // B * S
        float* prow = partial + (((int64_t)slot * t_rows + row0 + kg) * KV + kvh) * 128;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 vk, vv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vk[e] = dk[db][4 * g + e]; vv[e] = dv[db][4 * g + e]; }
                *reinterpret_cast<f32x4*>(prow + db * 32 + 8 * g + 4 * h) = vk;
                *reinterpret_cast<f32x4*>(prow + 64 + db * 32 + 8 * g + 4 * h) = vv;
            }
        TRACE_END(2, n_steps);
        return;
    }
    // lane = key, registers = d (runs of 4): 8-byte stores into the k and v column blocks of dqkv
    bf16_t* krow_out = dqkv + (row0 + kg) * ld + (int64_t)H * HD + (int64_t)kvh * HD;
    bf16_t* vrow_out = krow_out + (int64_t)KV * HD;
    const float* tb = rope ? rope + (int64_t)(positions ? positions[row0 + kg] : kg) * HD : nullptr;  // dK leaves in pre-RoPE space
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 vk, vv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vk[e] = (bf16_t)(dk[db][4 * g + e] * -0.125f);  // dk holds -sum dS Q
                vv[e] = (bf16_t)dv[db][4 * g + e];
            }
            if (tb) vk = unrope4(vk, tb, db * 32 + 8 * g + 4 * h);
            *reinterpret_cast<bf16x4*>(krow_out + db * 32 + 8 * g + 4 * h) = vk;
            *reinterpret_cast<bf16x4*>(vrow_out + db * 32 + 8 * g + 4 * h) = vv;
        }
    TRACE_END(2, n_steps);
}


// =====================================================================================================================
// backward: dK, dV — round 4: one wave per SIMD, hand-placed software pipeline
// =====================================================================================================================
// Same algebra, same operand images and the same summation order as attn_bwd_dkv_kernel (results are bit-identical), rebuilt around what
// its trace said (profiles/LAB_NOTES.md, round 3): a wave was bound by its own chain  fragment reads -> S / dP -> exponentials -> dV / dK,
// which two unsynchronised waves per SIMD overlapped only by chance (matrix pipe 41 % busy).  Here a wave has the SIMD to itself
// (__launch_bounds__(256, 1): the whole 512-entry register file) and overlaps the chain with itself:
//   * a wave owns 64 keys = two 32-key blocks kb; a workgroup = 256 keys of one (batch, kv head).  A UNIT = (query tile t, kb) is what a
//     step of the old kernel was: 8 S / dP products (SP), the exponentials (SM), 8 dV / dK products (DKV).  Q / dO row and transposed
//     fragments are read once per TILE and serve both units: half the LDS reads per product;
//   * a PERIOD = 16 MFMAs carries three units at once: SP of unit u+1, SM of unit u spread over the 16 MFMA gaps (per gap: one scale, one
//     exponential, one multiply, one packed conversion = 20 issue cycles beside the MFMA's 8, MI355X_MICROARCH.md "vector-instruction ISSUE cost"),
//     DKV of unit u-1.  Units alternate kb, so S / dP need one register set per kb and no double buffer;
//   * every MFMA is inline asm with its register class pinned (S / dP results in arch VGPRs where the vector ALU reads them, dK / dV sums and
//     the K / V operand fragments in accumulation registers) and every gap is closed by sched_barrier(0): hipcc allocates, the order is ours;
//   * the vector issue port is the scarce unit (8 + 20 of a gap's 32 cycles are taken), so the 32 LDS reads of a tile are SPREAD: one per gap
//     (two in 8 of the 32 gaps), each a register's last use behind and >= 8 gaps ahead of its first use; the LDS-DMA requests (ring of 12
//     tiles, one barrier and nine requests per wave per four tiles, counted vmcnt) go one per gap into the one half-period per tile that carries no LDS reads.  Bunched two
//     per gap in half of the gaps (first build) the reads cost 6-11 cycles each (in-kernel stamps).
// A wave does not skip the tiles in front of its keys (the old kernel's `return`): with one wave per SIMD nothing else could use the slot,
// and wave 0 of the workgroup needs every tile anyway — they run masked (p = 0 adds exact zeros).
constexpr int DKV2_RING = 12;
constexpr int DKV2_SB = 8192 + 256;
constexpr int DKV2_MAX_STEPS = 2048;  // tiles per workgroup = (S / 32) * rep at most: S <= 16384 at rep = 4

// VARLEN (round 5): packed rows.  The work comes from a host-built PLAN (ssi_attn_plan_build): an item = (row b, first key k0 — a multiple of
// 32 —, document [dstart, dend)) = the up to 256 keys k0 .. k0 + 255 of ONE document, items sorted by work, heaviest first; a workgroup =
// (item, kv head).  Because an item never leaves its document, everything that made packed rows expensive in the 128-key kernel is uniform
// here: the query tiles are those from k0 to the END OF THE DOCUMENT (tiles of other documents are skipped, not masked — they are simply
// not in the tile table), and the masked tiles are the 8 on the diagonal, with the plain rows' mask  key <= query.  The document's last tile,
// when the document does not end on a 32-row boundary, holds queries of the NEXT document: they are taken out by their row constant, not by
// a mask — the lanes that fetch lse[q] for q >= dend fetch 1e30 instead (one word of the plan's header), so P = exp2((S - lse) log2 e) = 0
// exactly and dS = P (dP - delta) = 0 for those rows at no cost in the loops (first build: a second condition in the mask, 1.5 compares and
// a scalar instruction per element more in every masked tile).  Lanes whose key lies outside [dstart, dend) — the head of the first item
// of a document that does not start on a 32-row boundary, the tail of its last item — compute on clamped rows and store nothing (key = lane:
// whatever they accumulate stays in their own columns).
template <bool VARLEN>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv2_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, const float* __restrict__ rope,
                                                               const int32_t* __restrict__ positions, int S, int H, int KV,
                                                               const int4* __restrict__ items, const float* __restrict__ lse_beyond,
                                                               float* __restrict__ partial) {
    constexpr int SB = DKV2_SB, RING = DKV2_RING;
    __shared__ __attribute__((aligned(16))) char smem[RING * SB + DKV2_MAX_STEPS * 4];  // ring of [Q tile 4 KiB | dO tile 4 KiB | lse 128 B | delta 128 B], tile table
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rep_all = H / KV;
    // query heads this workgroup sweeps: all of the kv head's, or (VARLEN) the item's share of them — a heavy item of the plan is split over
    // the query heads (2 or 4 workgroups whose fp32 sums meet in attn_dkv_plan_reduce_kernel), so that a launch is not as long as its longest document
    int rep = rep_all, head0 = 0, pslot = -1;
    int kvh, b, k0, dstart = 0, dend = S;
    if constexpr (VARLEN) {  // workgroup -> (item, kv head): consecutive workgroups = the kv heads of one item, i.e. (KV = 8) one per XCD
        const int id = (int)blockIdx.x;
        kvh = id % KV;
        const int4 it = items[2 * (id / KV)], ih = items[2 * (id / KV) + 1];  // (uniform address: scalar loads)
        b = it.x, k0 = it.y, dstart = it.z, dend = it.w;
        head0 = ih.x, rep = ih.y, pslot = ih.z;
    } else {
        const int ngrp = S / 256;
        int kgrp, pair_;  // low key groups (most work) are dispatched first
        block_to_work(ngrp, (int)(gridDim.x / ngrp), kgrp, pair_);
        kvh = pair_ % KV;
        b = pair_ / KV;
        k0 = kgrp * 256;
    }
    const int h = lane >> 5;
    const int64_t row0 = (int64_t)b * S;
    const int64_t ldo = (int64_t)H * HD;
    const int key0 = k0 + wave * 64;

    // operands and row constants exactly as in attn_bwd_dkv_kernel: -K * 2^-3 and -V as B operands, +lse / +delta as initial accumulators
    bf16x8 kf[2][4], vf[2][4];
    int kg[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        kg[kb] = key0 + 32 * kb + (lane & 31);
        const int krow_i = VARLEN ? (kg[kb] < S ? kg[kb] : S - 1) : kg[kb];  // (an item's last keys may lie beyond the row: not stored)
        const bf16_t* krow = qkv + (row0 + krow_i) * ld + (int64_t)H * HD + (int64_t)kvh * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[kb][ks] = scale_frag(*reinterpret_cast<const bf16x8*>(krow + 16 * ks), -0.125f);
            vf[kb][ks] = scale_frag(*reinterpret_cast<const bf16x8*>(krow + (int64_t)KV * HD + 16 * ks), -1.0f);
            // from here on the fragments LIVE in accumulation registers: an "a" input alone makes hipcc keep them in arch VGPRs and copy
            // them over (4 v_accvgpr_write) in front of every MFMA that names them
            asm volatile("" : "=a"(kf[kb][ks]) : "0"(kf[kb][ks]));
            asm volatile("" : "=a"(vf[kb][ks]) : "0"(vf[kb][ks]));
        }
    }
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[kb][i][r] = 0.f; dv[kb][i][r] = 0.f; }

    const int qb_first = k0 / 32;                                        // first 32-query tile that sees any key of the group
    const int per_head = (VARLEN ? (dend + 31) / 32 : S / 32) - qb_first;  // tiles per query head (plain rows: >= 8)
    // tiles of the two loops, each a multiple of 4 (a trip): plain rows come with rep % 4 == 0; an item of the plan that sweeps 1 or 2 heads
    // is padded with DUMMY tiles — any tile's Q / dO under lse = 1e30 for all its rows, i.e. P = 0, dS = 0: exact zeros added

    // ---- LDS-DMA requests: as in attn_bwd_dkv_kernel, three per tile and wave, issued part by part ----------------------------------------
    // Tile order: the MASKED tiles of every head first (the 8 tiles on the group's diagonal), then the rest of every head.  Two plain loops, one per form of the exponentials — not an if / else per period and not two inner loops taking turns:
    // wherever register tuples defined in different places meet (a diamond, a loop nest), hipcc's phi elimination splits them into scalars in
    // arch VGPRs and copies them into the accumulation registers in front of every MFMA (1 500 v_accvgpr moves and 400 scratch accesses in
    // the loop of the first build).  The order of the sums over the tiles differs from attn_bwd_dkv_kernel's, so the two kernels agree to
    // rounding, not bit for bit; each is reproducible run to run.
    // masked tiles per head: the group's diagonal (8 tiles; with VARLEN fewer when the document ends inside it)
    const int n_edge = VARLEN ? (per_head < 8 ? per_head : 8) : 8;
    const int n_masked_real = n_edge * rep, n_rest_real = (per_head - n_edge) * rep;
    const int n_masked = VARLEN ? (n_masked_real + 3) & ~3 : n_masked_real;          // tiles of the first loop
    const int n_steps = n_masked + (VARLEN ? (n_rest_real + 3) & ~3 : n_rest_real);
    // tile i of the sequence -> (head << 16) | tile of the head, looked up in a table in LDS behind the ring (built once per workgroup): the
    // requests run 6-7 tiles ahead of the products and cross heads and loops at other times, and a cursor kept in scalar registers by selects
    // cost ~50 scalar instructions per trip, all in front of its first MFMA
    int* seq_tab = reinterpret_cast<int*>(smem + RING * SB);
    for (int i = tid; i < n_steps; i += 256) {
        const int j = i < n_masked ? i : i - n_masked, len = i < n_masked ? n_edge : per_head - n_edge;
        int w = 0x8000;  // dummy: tile 0 of the first head, bit 15 = "its rows see nothing"
        if (!VARLEN || j < (i < n_masked ? n_masked_real : n_rest_real)) w = ((j / len) << 16) | ((i < n_masked ? 0 : n_edge) + j % len);
        seq_tab[i] = w;
    }
    const int irow = wave * 8 + (lane >> 3), ichunk = (lane & 7) ^ swz<SWZ_DUAL>(wave * 8 + (lane >> 3));
    const u32x4 rs_q = buffer_rsrc(qkv + row0 * ld + (int64_t)(kvh * rep_all + head0) * HD);
    const u32x4 rs_do = buffer_rsrc(dout + row0 * ldo + (int64_t)(kvh * rep_all + head0) * HD);
    const unsigned voff_q = (unsigned)((irow * ld + ichunk * 8) * 2), voff_do = (unsigned)((irow * ldo + ichunk * 8) * 2);
    const float* rc_base = (lane < 32 ? lse : delta) + ((int64_t)b * H + kvh * rep_all + head0) * S + (lane & 31);
    const unsigned lds_piece = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + (unsigned)wave * 1024u);
    const unsigned lds_rc = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_c*)smem + 8192u);
    // Requests are issued for EVERY ring position, also behind the last tile (the last tile again; the bytes go to a slot nobody reads): the
    // counted vmcnt waits hold without a tail case, and there is no branch inside a trip — hipcc sinks the pure vector instructions of a gap
    // across any basic-block boundary towards their users, which undoes the placement.
    // One request = M0 (LDS destination) written one gap AHEAD of the load that uses it (issue_m0 then issue_go, as in gemm_nt4dma: written
    // right in front of the load, every request stalls the wave's issue).
    auto seq_at = [&](int step) __attribute__((always_inline)) { return seq_tab[step < n_steps ? step : n_steps - 1]; };  // every lane reads the same word
    // Per trip of 4 tiles a wave issues 9 requests: the row constants of ONE of the four tiles (tile + wave: 256 B, lse | delta) FIRST, then its
    // Q and dO pieces of the four tiles (every wave used to fetch every tile's constants: 12 requests; a request costs its wave ~40 cycles)
    auto issue_m0 = [&](unsigned buf, int part) __attribute__((always_inline)) {  // buf = byte offset of the tile's ring slot
        const unsigned dst = part == 0 ? lds_piece + buf : part == 1 ? lds_piece + buf + 4096 : lds_rc + buf;
        asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
    };
    auto issue_go = [&](int w, int part) __attribute__((always_inline)) {  // w = table word of the tile
        const int qrow = (qb_first + (w & 0x7fff)) * 32, hoff = (w >> 16) * (HD * 2);
        if (part == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff_q), "s"(rs_q), "s"((unsigned)(qrow * (int)ld * 2 + hoff)) : "memory");
        else if (part == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff_do), "s"(rs_do), "s"((unsigned)(qrow * (int)ldo * 2 + hoff)) : "memory");
        else {
            const float* src = rc_base + ((w >> 16) * S + qrow);
            if constexpr (VARLEN) {  // queries of the next document (the document's last tile): lse = 1e30 -> P = 0, dS = 0
                const int seen_until = (w & 0x8000) ? 0 : dend;  // (a scalar select, no branch: a trip stays one basic block)
                if (lane < 32 && qrow + lane >= seen_until) src = lse_beyond;
            }
            asm volatile("global_load_lds_dword %0, off" ::"v"(src) : "memory");
        }
    };

    // ---- per-tile register state ------------------------------------------------------------------------------------------------------------
    f32x16 sacc[2], pacc[2];          // S' = lse - S and dP' = delta - dP of the unit in flight per key block
    f32x16 rcl, rcd;                  // lse / delta of the tile whose S / dP products come next (rows = queries rowmap(r, h))
    bf16x8 qfr[4], dfr[4];            // Q / dO row fragments of that tile
    s16x4 dtrh[2][2][2], qtrh[2][2][2];  // [s2][db][half]: dO / Q transposed fragments of the tile whose dV / dK products come next
    u32x4 pfu[2][2], dsu[2][2];       // [kb][s2]: P and -dS of a unit as bf16 operand fragments
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pfu[i][j] = u32x4{0u, 0u, 0u, 0u};
            dsu[i][j] = u32x4{0u, 0u, 0u, 0u};
            dtrh[i][j][0] = dtrh[i][j][1] = s16x4{0, 0, 0, 0};  // the first period's dV / dK products add 0 * 0
            qtrh[i][j][0] = qtrh[i][j][1] = s16x4{0, 0, 0, 0};
        }

    // Ring offsets of a trip as three scalars (RING = 12 is not a power of two and a trip's 4 tiles never wrap: t % 4 == 0): bytes of the slot of
    // tile t (trip start), of tile t+4 (next trip's first) and of tile t+8 (first requested); tile t+i of the trip sits i * SB further on.
    unsigned ring_cur = 0, ring_nxt = 4 * SB, ring_req = 8 * SB;
    auto tile_base = [&](int i) __attribute__((always_inline)) { return smem + (i < 4 ? ring_cur + i * SB : ring_nxt); };  // i = tile - trip start, 0..4
    // read i (0..15) of the 16 row reads of a tile: 0-3 lse (rows 4h + 8i ..+3), 4-7 delta, 8-11 Q row fragments, 12-15 dO row fragments
    auto read_rows = [&](const char* qt, int i) __attribute__((always_inline)) {
        const float* rcs = reinterpret_cast<const float*>(qt + 8192);
        if (i < 4) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(rcs + 4 * h + 8 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) rcl[4 * i + e] = l4[e];
        } else if (i < 8) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(rcs + 32 + 4 * h + 8 * (i - 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) rcd[4 * (i - 4) + e] = d4[e];
        } else if (i < 12) qfr[i - 8] = frag_row<SWZ_DUAL>(qt, 0, i - 8, lane);
        else dfr[i - 12] = frag_row<SWZ_DUAL>(qt + 4096, 0, i - 12, lane);
    };
    // read i (0..15) of the 16 transposed reads of a tile, in the order the dV / dK products use them: fragment i >> 1, half i & 1
    auto read_tr = [&](const char* qt, int i) __attribute__((always_inline)) {
        const int j = i >> 1, s2 = j >> 2, db = (j >> 1) & 1;
        if (j & 1) qtrh[s2][db][i & 1] = frag_tr_half<SWZ_DUAL>(qt, s2 * 16, db * 32, lane, i & 1);
        else dtrh[s2][db][i & 1] = frag_tr_half<SWZ_DUAL>(qt + 4096, s2 * 16, db * 32, lane, i & 1);
    };
    auto tr_frag = [&](const s16x4 (&hv)[2]) {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(hv[0], hv[1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // S / dP product m (0..7) of key block kb: the S chain first (m = 0..3 = k-slices), then the dP chain; the first of a chain takes the row
    // constants as C.  S first because the exponentials of the next period start with S: its last product is 4 MFMAs (128 cycles) old when the
    // first scale reads it, dP's last product >= 52 cycles when the first multiply does (asm MFMAs are invisible to hipcc's hazard
    // recogniser; an MFMA result needs 44).
    auto sp_mfma = [&](int kb, int m) __attribute__((always_inline)) {
        const int ks = m & 3;
        if (m == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(sacc[kb]) : "v"(qfr[0]), "a"(kf[kb][0]), "v"(rcl));
        else if (m == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(pacc[kb]) : "v"(dfr[0]), "a"(vf[kb][0]), "v"(rcd));
        else if (m > 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(pacc[kb]) : "v"(dfr[ks]), "a"(vf[kb][ks]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sacc[kb]) : "v"(qfr[ks]), "a"(kf[kb][ks]));
    };
    // dV / dK product j (0..7) of key block kb, in attn_bwd_dkv_kernel's order: for s2: for db: dV, dK
    auto dkv_mfma = [&](int kb, int j) __attribute__((always_inline)) {
        const int s2 = j >> 2, db = (j >> 1) & 1;
        if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dk[kb][db]) : "v"(tr_frag(qtrh[s2][db])), "v"(dsu[kb][s2]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dv[kb][db]) : "v"(tr_frag(dtrh[s2][db])), "v"(pfu[kb][s2]));
    };
    // exponentials of unit (tile at q0, kb), gap g of 16: scale + exponential of element g, -dS of element g - 1, one packed conversion
    float pv[16], dsv[16];
    auto sm_gap = [&](auto edge_c, int kb, int q0, int g) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value;
        {
            float p = __builtin_amdgcn_exp2f(sacc[kb][g] * -LOG2E);
            if (EDGE) {
                const int q = q0 + rowmap(g, h);
                if (kg[kb] > q) p = 0.f;  // keys beyond the query contribute nothing
            }
            pv[g] = p;
        }
        if (g >= 1) dsv[g - 1] = pacc[kb][g - 1] * pv[g - 1];
        if (g >= 2 && !(g & 1)) { const int j = (g - 2) >> 1; pfu[kb][j >> 2][j & 3] = pack_bf16(pv[g - 2], pv[g - 1]); }
        if (g >= 3 && (g & 1)) { const int j = (g - 3) >> 1; dsu[kb][j >> 2][j & 3] = pack_bf16(dsv[g - 3], dsv[g - 2]); }
        if (g == 15) {
            dsv[15] = pacc[kb][15] * pv[15];
            pfu[kb][1][3] = pack_bf16(pv[14], pv[15]);
            dsu[kb][1][3] = pack_bf16(dsv[14], dsv[15]);
        }
    };

    // -DDKV2_STAMP (debug build, tools/attn_dkv_check.py stamps): cycle totals of wave 0 per half-period, by kind of tile (with / without the
    // barrier), left in the first floats of the workgroup's first dq row.  The stamp waits for the LDS reads in flight: read shares, not lengths.
#ifdef DKV2_STAMP
    unsigned long long st2_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st2_last = __builtin_readcyclecounter();
    const unsigned long long st2_begin = st2_last;
#define STAMP2(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); st2_acc[i] += now_ - st2_last; st2_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP2(i) {}
#endif
#ifdef DKV2_STAMP_GAPS  // debug build: cycles per GAP of the periods of a tile without barrier / requests (wave 0), 32 totals
    unsigned long long sg_acc[32];
    for (int i = 0; i < 32; ++i) sg_acc[i] = 0;
    unsigned long long sg_last = __builtin_readcyclecounter();
#define STAMPG(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); sg_acc[i] += now_ - sg_last; sg_last = now_; __builtin_amdgcn_sched_barrier(0); }
#define STAMPG_RESET() { __builtin_amdgcn_sched_barrier(0); sg_last = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#define STAMPG_ON 1
#else
#define STAMPG_ON 0
#define STAMPG(i) {}
#define STAMPG_RESET() {}
#endif
    // period A of tile t: SM of unit (t, 0);  MFMAs 0-7 = dV / dK of (t-1, 1), 8-15 = S / dP of (t, 1) — the products whose results the vector
    // ALU needs come LAST, so that at most ~1.4 S / dP register sets are live at any time (first-half S / dP put 296 registers in flight and the
    // fragments into scratch).  Behind MFMA 7: the ring barrier (every second tile).  LDS reads: one transposed read of tile t per gap (fragment
    // j right behind the last use of tile t-1's fragment j), and from gap 10 on also the row constants of tile t+1
    auto period_a = [&](auto edge_c, auto sync_c, auto pos_c, int t, int q0) __attribute__((always_inline)) {
        constexpr int POS = decltype(pos_c)::value;  // position of tile t in its trip
        const char* ct_ = tile_base(POS);
        const char* nt_ = tile_base(POS + 1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < 8) dkv_mfma(1, m);
            else sp_mfma(1, m - 8);
            sm_gap(edge_c, 0, q0, m);
            if (m == 7 && decltype(sync_c)::value) {
                __builtin_amdgcn_sched_barrier(0);
                // own requests of tiles t+1 .. t+4 have landed — the Q / dO pieces of t+5, t+6, t+7 stay in flight; the row constants of
                // t+4 .. t+7 were this wave's FIRST request of the last trip ...
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                ring_barrier();  // ... and everybody's; every wave is done with tiles t-4 .. t-1: their slots are free
            }
            read_tr(ct_, m);                      // fragment m >> 1 of tile t-1 had its last use in MFMA m >> 1
            if (m >= 10) read_rows(nt_, m - 10);  // lse 0-3, delta 0-1 of tile t+1
            // An MFMA reads its C operand over its whole run and hipcc does not know the asm is one: left to itself it handed the registers of
            // lse / delta (dead to it behind MFMA 8 / 12) to the very next vector instruction, and the products ran on a half-overwritten C.
            // Keep them alive for two more gaps (64 cycles).
            if (m == 9) asm volatile("" ::"v"(rcl));
            if (m == 13) asm volatile("" ::"v"(rcd));
            __builtin_amdgcn_sched_barrier(0);
            if (m == 7) STAMP2(decltype(sync_c)::value ? 0 : 4)
            if (STAMPG_ON && !decltype(sync_c)::value && t % 4 == 3) STAMPG(m)
        }
        STAMP2(decltype(sync_c)::value ? 1 : 5)
    };
    // period B of tile t: SM of unit (t, 1);  MFMAs 0-7 = dV / dK of (t, 0), 8-15 = S / dP of (t+1, 0);  row constants and row fragments of tile
    // t+1 under the first half (constants first: they are the C operands of MFMAs 8 and 9), the LDS-DMA requests of tiles t+RING-2, t+RING-1
    // (every second tile) under the second, which carries no LDS reads
    auto period_b = [&](auto edge_c, auto issue_c, int t, int q0) __attribute__((always_inline)) {
        const char* nt_ = tile_base(decltype(issue_c)::value + 1);
        int wv0 = 0, wv1 = 0, w0 = 0, w1 = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < 8) dkv_mfma(0, m);
            else sp_mfma(0, m - 8);
            sm_gap(edge_c, 1, q0, m);
            if (m < 2) read_rows(nt_, 6 + m);  // delta 2-3
            if (m < 8) read_rows(nt_, 8 + m);  // Q / dO row fragments, each 8 gaps ahead of its product
            {   // requests for tile t + RING - 4 (its slot was freed by this trip's barrier); in the trip's first tile also the row constants
                constexpr int KIND = decltype(issue_c)::value;  // position of tile t in its trip
                const int rt = t + RING - 4;
                if (m == 0) { wv0 = seq_at(rt); if (KIND == 0) wv1 = seq_at(rt + wave); }
                if (m == 6) { w0 = __builtin_amdgcn_readfirstlane(wv0); if (KIND == 0) w1 = __builtin_amdgcn_readfirstlane(wv1); }
                if (KIND == 0) {
                    if (m == 8) issue_go(w1, 2);
                    if (m == 9 || m == 10) issue_go(w0, m - 9);
                    if (m == 7) issue_m0(ring_req + (unsigned)wave * SB, 2);
                    if (m == 8 || m == 9) issue_m0(ring_req, m - 8);
                } else {
                    if (m == 9 || m == 10) issue_go(w0, m - 9);
                    if (m == 8 || m == 9) issue_m0(ring_req + KIND * SB, m - 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m == 7) STAMP2(decltype(issue_c)::value == 0 ? 2 : 6)
            if (decltype(issue_c)::value == 3) STAMPG(16 + m)
        }
        STAMP2(decltype(issue_c)::value == 0 ? 3 : 7)
        if (decltype(issue_c)::value == 2) STAMPG_RESET()
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------------------------
    // Everything this wave has loaded from global memory is consumed HERE: hipcc does not see the LDS-DMA requests below, and its wait for a
    // value first used inside the loops (packed rows' document ends, when this kernel still took them) was `s_waitcnt vmcnt(0)` in every trip.
    asm volatile("" ::"v"(kg[0]), "v"(kg[1]) : "memory");
    __syncthreads();  // the table is complete (nothing is in flight yet that a vmcnt(0) could drain)
    // tiles 0 .. RING-5: this wave's two row-constant requests first (tiles wave and wave + 4), then its Q / dO pieces
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = __builtin_amdgcn_readfirstlane(seq_at(wave + 4 * i));
        issue_m0((unsigned)(wave + 4 * i) * SB, 2);
        asm volatile("s_nop 0" ::: "memory");
        issue_go(w, 2);
    }
#pragma unroll
    for (int i = 0; i < RING - 4; ++i) {
        const int w = __builtin_amdgcn_readfirstlane(seq_at(i));
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            issue_m0((unsigned)i * SB, part);
            asm volatile("s_nop 0" ::: "memory");
            issue_go(w, part);
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING - 4) - 2) : "memory");  // this wave's constants and its pieces of tile 0 have landed
    ring_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) read_rows(smem, i);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");  // the K / V fragments reach the asm MFMAs through v_accvgpr_write: let the last one land
#pragma unroll
    for (int m = 0; m < 8; ++m) sp_mfma(0, m);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // asm MFMAs are opaque to the hazard recogniser: S / dP of unit (0, 0) must have landed

    // ---- main loops: four tiles per trip (one barrier, nine requests per wave); a trip is ONE basic block ------------------------------------
    using T_ = std::true_type;
    using F_ = std::false_type;
    int cur_qt = 0;  // tile of the head in the masked loop (the other loop needs no query positions)
    auto trip = [&](auto edge_c, int t) __attribute__((always_inline)) {
        int q0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            q0[i] = (qb_first + cur_qt) * 32;
            cur_qt = cur_qt + 1 == n_edge ? 0 : cur_qt + 1;
        }
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        using P3 = std::integral_constant<int, 3>;
        period_a(edge_c, T_{}, P0{}, t, q0[0]);
        period_b(edge_c, P0{}, t, q0[0]);
        period_a(edge_c, F_{}, P1{}, t + 1, q0[1]);
        period_b(edge_c, P1{}, t + 1, q0[1]);
        period_a(edge_c, F_{}, P2{}, t + 2, q0[2]);
        period_b(edge_c, P2{}, t + 2, q0[2]);
        period_a(edge_c, F_{}, P3{}, t + 3, q0[3]);
        period_b(edge_c, P3{}, t + 3, q0[3]);
        ring_cur = ring_nxt;
        ring_nxt = ring_req;
        ring_req = ring_req == 8 * SB ? 0u : ring_req + 4 * SB;
    };
    int t = 0;
    for (; t < n_masked; t += 4) trip(T_{}, t);
    for (; t < n_steps; t += 4) trip(F_{}, t);
    // ---- drain: dV / dK of the last unit ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 8; ++j) dkv_mfma(1, j);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");  // results landed; no request of this wave is left in flight towards LDS
#ifdef DKV2_STAMP
    const unsigned long long st2_total = __builtin_readcyclecounter() - st2_begin;
#endif

    if constexpr (VARLEN) {
        if (pslot >= 0) {  // an item split over the query heads: raw fp32 sums [slot][kv head][key of the item][dK 64 | dV 64]; scale, RoPE backward
#pragma unroll             // and rounding happen after the heads are added (attn_dkv_plan_reduce_kernel)
            for (int kb = 0; kb < 2; ++kb) {
                float* prow = partial + (((int64_t)pslot * KV + kvh) * 256 + (kg[kb] - k0)) * 128;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 vk, vv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vk[e] = dk[kb][db][4 * g + e]; vv[e] = dv[kb][db][4 * g + e]; }
                        *reinterpret_cast<f32x4*>(prow + db * 32 + 8 * g + 4 * h) = vk;
                        *reinterpret_cast<f32x4*>(prow + 64 + db * 32 + 8 * g + 4 * h) = vv;
                    }
            }
            return;
        }
    }
    const float* tb0 = rope ? rope : nullptr;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (VARLEN && (kg[kb] < dstart || kg[kb] >= dend)) continue;  // another item's key (or none)
        bf16_t* krow_out = dqkv + (row0 + kg[kb]) * ld + (int64_t)H * HD + (int64_t)kvh * HD;
        bf16_t* vrow_out = krow_out + (int64_t)KV * HD;
        const float* tb = tb0 ? tb0 + (int64_t)(positions ? positions[row0 + kg[kb]] : kg[kb]) * HD : nullptr;  // dK leaves in pre-RoPE space
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 vk, vv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vk[e] = (bf16_t)(dk[kb][db][4 * g + e] * -0.125f);  // dk holds -sum dS Q
                    vv[e] = (bf16_t)dv[kb][db][4 * g + e];
                }
                if (tb) vk = unrope4(vk, tb, db * 32 + 8 * g + 4 * h);
                *reinterpret_cast<bf16x4*>(krow_out + db * 32 + 8 * g + 4 * h) = vk;
                *reinterpret_cast<bf16x4*>(vrow_out + db * 32 + 8 * g + 4 * h) = vv;
            }
    }
#ifdef DKV2_STAMP
    if (wave == 0 && lane == 0) {  // DEBUG BUILD ONLY: overwrites the first floats of the workgroup's first dq row
        float* dbg = reinterpret_cast<float*>(dqkv + (row0 + k0) * ld);
        for (int i = 0; i < 8; ++i) dbg[i] = (float)st2_acc[i];
        dbg[8] = (float)st2_total;
        dbg[9] = (float)n_steps;
        dbg[10] = (float)(k0 / 256);
    }
#endif
#ifdef DKV2_STAMP_GAPS
    if (wave == 0 && lane == 0) {  // DEBUG BUILD ONLY
        float* dbg = reinterpret_cast<float*>(dqkv + (row0 + k0) * ld);
        for (int i = 0; i < 32; ++i) dbg[i] = (float)sg_acc[i];
        dbg[32] = (float)n_steps;
    }
#endif
}

}  // namespace

#ifdef ATTN_TRACE
extern "C" int ssi_debug_attn_trace(void* dst_host, int kernel) {  // debug build only: copy one kernel's table to the host
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_attn_trace), sizeof(unsigned long long) * TRACE_MAX * 6,
                                    sizeof(unsigned long long) * TRACE_MAX * 6 * (size_t)kernel, hipMemcpyDeviceToHost);
}
#endif

// Adds the per-head partial rows of the HSPLIT form in head order (fixed: reproducible), then does what the unsplit kernel's epilogue does:
// dK * -2^-3 (the sums carry the opposite sign), optional RoPE backward, rounding, stores.  One thread per (row, kv head, 4 columns).
__global__ __launch_bounds__(256) void attn_dkv_head_reduce_kernel(const float* __restrict__ partial, int rep, int64_t t_rows, int KV,
                                                                   bf16_t* __restrict__ dqkv, int64_t ld, int H, const float* __restrict__ rope,
                                                                   const int32_t* __restrict__ positions, int S) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, kvh, c4) with c4 = 0..31: dK columns 4 c4 .. (c4 < 16), dV columns (c4 - 16) * 4 ..
    if (i >= t_rows * KV * 32) return;
    const int c4 = (int)(i & 31), kvh = (int)((i >> 5) % KV);
    const int64_t row = (i >> 5) / KV;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int hd = 0; hd < rep; ++hd) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (((int64_t)hd * t_rows + row) * KV + kvh) * 128 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    bf16x4 o;
    bf16_t* dst;
    if (c4 < 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(sum[e] * -0.125f);
        if (rope) o = unrope4(o, rope + (int64_t)(positions ? positions[row] : (int)(row % S)) * HD, c4 * 4);
        dst = dqkv + row * ld + (int64_t)H * HD + (int64_t)kvh * HD + c4 * 4;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)sum[e];
        dst = dqkv + row * ld + (int64_t)(H + KV) * HD + (int64_t)kvh * HD + (c4 - 16) * 4;
    }
    *reinterpret_cast<bf16x4*>(dst) = o;
}

// The items of a plan that were split over the query heads: adds their slots' fp32 rows in slot order (fixed: reproducible), then the epilogue of
// attn_bwd_dkv2_kernel.  red = {b, k0, dstart, dend}, {first slot, slots, 0, 0} per split 256-key chunk; one thread per (key, 4 columns).
__global__ __launch_bounds__(256) void attn_dkv_plan_reduce_kernel(const float* __restrict__ partial, const int4* __restrict__ red, int KV,
                                                                   bf16_t* __restrict__ dqkv, int64_t ld, int H, const float* __restrict__ rope,
                                                                   const int32_t* __restrict__ positions, int S) {
    const int chunk = (int)blockIdx.x / (KV * 32), kvh = ((int)blockIdx.x / 32) % KV;
    const int4 it = red[2 * chunk], is = red[2 * chunk + 1];
    const int i = ((int)blockIdx.x % 32) * 256 + (int)threadIdx.x;  // (key of the chunk, c4): dK columns 4 c4 .. (c4 < 16), dV columns 4 (c4 - 16) ..
    const int key = i >> 5, c4 = i & 31, kg = it.y + key;
    if (kg < it.z || kg >= it.w) return;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < is.y; ++sl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (((int64_t)(is.x + sl) * KV + kvh) * 256 + key) * 128 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += v[e];
    }
    const int64_t row = (int64_t)it.x * S + kg;
    bf16x4 o;
    bf16_t* dst;
    if (c4 < 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(sum[e] * -0.125f);
        if (rope) o = unrope4(o, rope + (int64_t)(positions ? positions[row] : kg) * HD, c4 * 4);
        dst = dqkv + row * ld + (int64_t)H * HD + (int64_t)kvh * HD + c4 * 4;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)sum[e];
        dst = dqkv + row * ld + (int64_t)(H + KV) * HD + (int64_t)kvh * HD + (c4 - 16) * 4;
    }
    *reinterpret_cast<bf16x4*>(dst) = o;
}

// Which backward kernels ssi_attn_bwd_mfma may take (ssi_set_attn_impl; process-global like ssi_set_impl).  The environment variables
// SSI_ATTN_DQ / SSI_ATTN_DKV give the INITIAL values, read once under C++ static initialisation — never on the launch path.
static int attn_env_mode(const char* name, int max_mode) {
    const char* s = getenv(name);
    const int v = (s && s[0] >= '0' && s[0] <= '9' && !s[1]) ? s[0] - '0' : 0;
    return v <= max_mode ? v : 0;
}
static std::atomic<int>& attn_mode(int which) {
    static std::atomic<int> modes[2] = {{attn_env_mode("SSI_ATTN_DQ", SSI_ATTN_MODE_NEW)}, {attn_env_mode("SSI_ATTN_DKV", SSI_ATTN_MODE_NO_HEAD_SPLIT)}};
    return modes[which];
}
extern "C" int ssi_set_attn_impl(int which, int mode) {
    if (which != SSI_ATTN_KERNEL_DQ && which != SSI_ATTN_KERNEL_DKV) return -1;
    const int max_mode = which == SSI_ATTN_KERNEL_DQ ? SSI_ATTN_MODE_NEW : SSI_ATTN_MODE_NO_HEAD_SPLIT;
    if (mode < 0 || mode > max_mode) return attn_mode(which).load(std::memory_order_relaxed);
    return attn_mode(which).exchange(mode, std::memory_order_relaxed);
}

// fp32 workspace the head-split dK / dV form wants for this shape (0: the launch fills the chip without it, or a single head per kv head)
// Workgroups per key group of the split form: all `rep` heads apart below 512 workgroups (two fit a CU: 512 fill the chip once, and the heaviest
// of them — queries x heads steps — is as long as the launch), two halves below 1024 (one long packed row: B = 1, S = 11 520 gives 720 workgroups
// whose heaviest sweeps a whole document x 4 heads = 256 steps where the chip's share per slot is 127); 1 = unsplit.
static int dkv_head_slots(int64_t batch, int64_t seq, int n_heads, int n_kv) {
    const int rep = n_heads / n_kv;
    const int64_t wgs = batch * n_kv * (seq / 128);
    if (rep <= 1 || seq % 128) return 1;
    if (wgs < 512) return rep;
    if (wgs < 1024 && rep % 2 == 0) return 2;
    return 1;
}
int64_t ssi_attn_mfma_bwd_workspace_bytes(int64_t batch, int64_t seq, int n_heads, int n_kv) {
    const int slots = dkv_head_slots(batch, seq, n_heads, n_kv);
    return slots <= 1 ? 0 : (int64_t)slots * batch * seq * n_kv * 128 * (int64_t)sizeof(float);
}

bool ssi_attn_mfma_supported(int64_t ld, int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype) {
    if (dtype != SSI_BF16 || head_dim != HD) return false;
    const int rep = n_heads / n_kv;
    if (rep != 1 && rep != 2 && rep != 4) return false;
    const int wg_rows = 32 * ANW / rep < 128 ? 128 : 32 * ANW / rep;  // query rows per forward workgroup; dK/dV: 128-key groups
    if (seq < wg_rows || seq % wg_rows != 0) return false;
    if (ld % 8 != 0 || batch <= 0) return false;
    if (batch * n_kv * (seq / 32) > (1LL << 30)) return false;
    return true;
}

int ssi_attn_fwd_mfma(const void* qkv, int64_t ld, void* out, float* lse, const int32_t* doc_start, int64_t batch, int64_t seq,
                      int n_heads, int n_kv, void* stream) {
    const int rep = n_heads / n_kv, qpw = ANW / rep;
    const unsigned grid = (unsigned)(batch * n_kv * (seq / (32 * qpw)));
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(64 * ANW), 0, (hipStream_t)stream, (const bf16_t*)qkv, ld, (bf16_t*)out, lse,
                       doc_start, (int)seq, n_heads, n_kv);
    SSI_LAUNCH_CHECK();
    return SSI_OK;
}

// ---- work plan for packed rows (ABI v7) --------------------------------------------------------------------------------------------------
// Layout of a plan (int32 words; built on the HOST by ssi_attn_plan_build, copied to the device by the caller; the first SSI_ATTN_PLAN_HEADER
// words are also what the launch needs on the host):
//   [0] magic  [1] n_dkv_items  [2] word offset of the dK/dV items  [3] n_dq_groups  [4] word offset of the dQ groups  [5] words per dQ group
//   [6] batch  [7] seq  [8] n_heads  [9] n_kv  [10] total words  [11] n_docs  [12] 1e30f (read by attn_bwd_dkv2_kernel<true>)
//   [13] n_reduce  [14] word offset of the reduce list  [15] fp32 partial slots (workspace = slots x n_kv x 256 x 128 x 4 bytes)
//   dK/dV items: n_dkv_items x [{b, k0, dstart, dend}, {first query head, query heads, partial slot or -1, 0}], heaviest first
//   reduce list: n_reduce x [{b, k0, dstart, dend}, {first slot, slots, 0, 0}]: the 256-key chunks that were split over the query heads
//   dQ groups  : n_dq_groups x [{n_items, load, 0, 0}, cap x {b, q0, dstart, dend}] (see attn_bwd_dq2_kernel<0, true>)
constexpr int32_t PLAN_MAGIC = 0x53534950;  // "SSIP"
static_assert(SSI_ATTN_PLAN_HEADER == 16, "plan header");

extern "C" int64_t ssi_attn_plan_words(int64_t batch, int64_t seq, int64_t n_docs) {
    if (batch <= 0 || seq <= 0 || n_docs <= 0) return 0;
    const int64_t dkv = batch * seq / 256 + 2 * n_docs, dq = batch * seq / 64 + 2 * n_docs;
    return SSI_ATTN_PLAN_HEADER + 8 * 4 * dkv /* split 4 ways */ + 8 * dkv /* reduce list */ + 4 * (dq + 512 /* group headers */ + dq /* slack of the fixed group stride */);
}

// Returns the number of words written (> 0), 0 when the pipelined kernels do not take this batch (the caller then passes no plan and the
// round-1..3 kernels run), < 0 on a bad argument.
extern "C" int64_t ssi_attn_plan_build(const int32_t* host_doc_row, const int32_t* host_doc_start, const int32_t* host_doc_end, int64_t n_docs,
                                       int64_t batch, int64_t seq, int n_heads, int n_kv, int flags, int32_t* host_plan, int64_t plan_words) {
    if (!host_doc_row || !host_doc_start || !host_doc_end || !host_plan || n_docs <= 0 || batch <= 0 || seq <= 0 || n_heads <= 0 || n_kv <= 0) return -1;
    if (plan_words < ssi_attn_plan_words(batch, seq, n_docs)) return -1;
    const bool force = (flags & SSI_ATTN_PLAN_FORCE) != 0;
    if (n_heads != 4 * n_kv || seq % 128 != 0 || seq > (1 << 24)) return 0;  // the pipelined kernels: 4 query heads per kv head
    struct It { int32_t b, r0, ds, de, work, head0 = 0, heads = 4, pslot = -1; };
    std::vector<It> dkv, dq;
    std::vector<int64_t> covered((size_t)batch, 0);
    int64_t keys = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        const int32_t b = host_doc_row[d], ds = host_doc_start[d], de = host_doc_end[d];
        if (b < 0 || b >= batch || ds < 0 || de <= ds || de > seq) return -1;
        covered[(size_t)b] += de - ds;
        keys += de - ds;
        for (int32_t k0 = ds & ~31; k0 < de; k0 += 256) {
            const int per_head = (de + 31) / 32 - k0 / 32;
            if (per_head * 4 + 8 > DKV2_MAX_STEPS) return 0;  // a document longer than the tile table
            dkv.push_back({b, k0, ds, de, per_head * 4});
        }
        for (int32_t q0 = ds & ~63; q0 < de; q0 += 64) dq.push_back({b, q0, ds, de, (q0 >> 6) - (ds >> 6) + 1});
    }
    for (int64_t b = 0; b < batch; ++b)
        if (covered[(size_t)b] != seq) return -1;  // the documents must tile every row (overlaps are the caller's bug; gaps are caught here)
    // stable sorts by work, heaviest first (ties keep document order: reproducible plans)
    auto by_work = [](const It& x, const It& y) { return x.work > y.work; };
    std::stable_sort(dq.begin(), dq.end(), by_work);
    // dK/dV is one workgroup per (item, kv head) and CU: a launch lasts at least as long as its heaviest item.  Items above the chip's share
    // per CU are split over the query heads (2 x 2 or 4 x 1 heads; fp32 partial sums, added by a reduction pass over those chunks only)
    int64_t total = 0;
    for (const It& it : dkv) total += it.work;
    const int64_t share = std::max<int64_t>(1, total * n_kv / 256), n_chunks = (int64_t)dkv.size();
    const bool split_all = (flags & SSI_ATTN_PLAN_SPLIT_ALL) != 0;
    const int64_t split_pct = 115;  // a chunk is split from 1.15 x the share on (measured: 80 / 60 / 45 % split more chunks and LOSE 2-10 %, LAB_NOTES round 5)
    std::vector<It> items, red;
    int32_t n_slots = 0;
    for (const It& it : dkv) {
        int ways = 1;
        if (split_all) ways = (it.r0 / 256) % 2 ? 2 : 4;
        else if (it.work * 100 > share * split_pct && it.work > 32) ways = it.work * 100 > share * 2 * split_pct ? 4 : 2;
        if (ways == 1) { items.push_back(it); continue; }
        It r = it;
        r.pslot = n_slots, r.heads = ways;  // (reduce entry: first slot, slots)
        red.push_back(r);
        for (int w = 0; w < ways; ++w) {
            It part = it;
            part.heads = 4 / ways, part.head0 = w * (4 / ways), part.pslot = n_slots++, part.work = it.work / ways;
            items.push_back(part);
        }
    }
    std::stable_sort(items.begin(), items.end(), by_work);
    dkv.swap(items);
    if (!force) {
        // many short documents: an item has room for 256 keys (dK/dV) / 64 queries (dQ) whatever the document holds
        if (n_chunks * 256 > 2 * keys + 2048 || (int64_t)dq.size() * 64 > 2 * keys + 2048) return 0;
    }
    // dQ: groups of equal load for persistent workgroups, one round of the chip (256 workgroups over n_kv heads), longest processing time first
    const int fixed_cost = 6;  // an item's cost outside its tiles, in tiles (14 000 of ~2 400 cycles)
    int n_groups = (int)std::min<int64_t>((int64_t)dq.size(), std::max<int64_t>(1, 256 / n_kv));
    std::vector<std::vector<It>> groups((size_t)n_groups);
    std::vector<int64_t> load((size_t)n_groups, 0);
    for (const It& it : dq) {
        int g = 0;
        for (int j = 1; j < n_groups; ++j)
            if (load[(size_t)j] < load[(size_t)g]) g = j;
        groups[(size_t)g].push_back(it);
        load[(size_t)g] += it.work + fixed_cost;
    }
    size_t cap = 0;
    for (const auto& g : groups) cap = std::max(cap, g.size());
    const int64_t gstride = 4 * (1 + (int64_t)cap);
    const int64_t dkv_off = SSI_ATTN_PLAN_HEADER, red_off = dkv_off + 8 * (int64_t)dkv.size(), dq_off = red_off + 8 * (int64_t)red.size();
    const int64_t words = dq_off + gstride * n_groups;
    if (words > plan_words) return -1;
    int32_t* hd = host_plan;
    for (int i = 0; i < SSI_ATTN_PLAN_HEADER; ++i) hd[i] = 0;
    hd[0] = PLAN_MAGIC, hd[1] = (int32_t)dkv.size(), hd[2] = (int32_t)dkv_off, hd[3] = n_groups, hd[4] = (int32_t)dq_off, hd[5] = (int32_t)gstride;
    hd[6] = (int32_t)batch, hd[7] = (int32_t)seq, hd[8] = n_heads, hd[9] = n_kv, hd[10] = (int32_t)words, hd[11] = (int32_t)n_docs;
    { const float big = 1e30f; memcpy(&hd[12], &big, sizeof(float)); }  // the "log-sum-exp" of a query that belongs to another document: P = 0
    hd[13] = (int32_t)red.size(), hd[14] = (int32_t)red_off, hd[15] = n_slots;
    int32_t* w = host_plan + dkv_off;
    for (const It& it : dkv) { w[0] = it.b, w[1] = it.r0, w[2] = it.ds, w[3] = it.de, w[4] = it.head0, w[5] = it.heads, w[6] = it.pslot, w[7] = 0; w += 8; }
    for (const It& it : red) { w[0] = it.b, w[1] = it.r0, w[2] = it.ds, w[3] = it.de, w[4] = it.pslot, w[5] = it.heads, w[6] = w[7] = 0; w += 8; }
    for (int g = 0; g < n_groups; ++g) {
        w = host_plan + dq_off + gstride * g;
        w[0] = (int32_t)groups[(size_t)g].size(), w[1] = (int32_t)load[(size_t)g], w[2] = w[3] = 0;
        w += 4;
        for (size_t i = 0; i < cap; ++i, w += 4) {
            if (i < groups[(size_t)g].size()) { const It& it = groups[(size_t)g][i]; w[0] = it.b, w[1] = it.r0, w[2] = it.ds, w[3] = it.de; }
            else w[0] = w[1] = w[2] = w[3] = 0;
        }
    }
    return words;
}

extern "C" int64_t ssi_attn_plan_workspace_bytes(const int32_t* host_plan_header) {
    if (!host_plan_header || host_plan_header[0] != PLAN_MAGIC) return -1;
    return (int64_t)host_plan_header[15] * host_plan_header[9] * 256 * 128 * (int64_t)sizeof(float);
}

static std::atomic<int> g_last_dispatch{0};
void ssi_attn_note_dispatch(int v) { g_last_dispatch.store(v, std::memory_order_relaxed); }
extern "C" int ssi_attn_last_dispatch(void) { return g_last_dispatch.load(std::memory_order_relaxed); }

// plan_dev: the plan in device memory, host_plan_header: its first SSI_ATTN_PLAN_HEADER words on the host (both NULL: no plan)
int ssi_attn_bwd_mfma(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                      const int32_t* doc_start, const int32_t* doc_end, const float* rope, int64_t table_len, const int32_t* positions,
                      int64_t batch, int64_t seq, int n_heads, int n_kv, void* workspace, int64_t workspace_bytes, const int32_t* plan_dev,
                      const int32_t* host_plan_header, void* stream) {
    auto st = (hipStream_t)stream;
    const int rep = n_heads / n_kv, qpw = ANW / rep;
    int used = 0;
    const int selq = attn_mode(SSI_ATTN_KERNEL_DQ).load(std::memory_order_relaxed);
    const int sel = attn_mode(SSI_ATTN_KERNEL_DKV).load(std::memory_order_relaxed);
    // packed rows with a plan: the document-aware forms of the pipelined kernels (mode OLD sends either kernel back to the round-1..3 one)
    const int32_t* ph = (plan_dev && host_plan_header) ? host_plan_header : nullptr;
    if (ph) {
        // (plain causal rows — no document arrays — take a plan whose documents are the rows themselves: the same work, dealt out by load)
        if (ph[0] != PLAN_MAGIC || ph[6] != batch || ph[7] != seq || ph[8] != n_heads || ph[9] != n_kv || rep != 4 || ph[1] <= 0 || ph[3] <= 0 ||
            (rope && table_len <= 0) || ((!doc_start || !doc_end) && (ph[11] != batch || positions))) {
            ssi_set_error("ssi_attn_varlen_bwd_plan: the plan does not belong to this batch (magic %x, batch %d, seq %d, heads %d / %d)", ph[0], ph[6],
                          ph[7], ph[8], ph[9]);
            return SSI_ERR_ARG;
        }
    }
    // dQ: the pipelined one-wave-per-SIMD kernel (persistent workgroups of 8, 4 or 2 query blocks: the largest count whose workgroups fill
    // the chip in whole rounds of 256, or in many rounds) for plain causal rows of 4 query heads per kv head; ssi_set_attn_impl(DQ, OLD)
    // keeps the round-1..3 kernel, NEW forces this one (8 blocks per workgroup if S allows, else 4, 2) whatever the fill
    int dq2_items = 0;
    if (!doc_start && !positions && rep == 4 && seq % 128 == 0 && selq != SSI_ATTN_MODE_OLD) {
        const int64_t nqb = seq / 64;
        for (int it = 8; it >= 2 && !dq2_items; it >>= 1) {
            if (nqb % it) continue;
            const int64_t grid = batch * n_kv * (nqb / it);
            if (grid % 256 == 0 || grid >= 1024) dq2_items = it;
        }
        if (!dq2_items && selq == SSI_ATTN_MODE_NEW) dq2_items = nqb % 8 == 0 ? 8 : nqb % 4 == 0 ? 4 : 2;
    }
    if (ph && selq != SSI_ATTN_MODE_OLD) {
        hipLaunchKernelGGL((attn_bwd_dq2_kernel<0, true>), dim3((unsigned)(ph[3] * n_kv)), dim3(256), 0, st, (const bf16_t*)qkv, ld, (const bf16_t*)out,
                           (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, rope, (int)seq, n_heads, n_kv, 0,
                           reinterpret_cast<const int4*>(plan_dev + ph[4]), ph[5] / 4, (int)std::min<int64_t>(table_len, 1 << 30));
        used |= SSI_ATTN_USED_DQ2 | SSI_ATTN_USED_PLAN;
    } else if (dq2_items) {
        const int w = (int)(seq / 64 / dq2_items);
        const dim3 grid((unsigned)(batch * n_kv * w));
        auto kern = dq2_items == 8 ? attn_bwd_dq2_kernel<8> : dq2_items == 4 ? attn_bwd_dq2_kernel<4> : attn_bwd_dq2_kernel<2>;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, (const bf16_t*)qkv, ld, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv,
                           rope, (int)seq, n_heads, n_kv, w, (const int4*)nullptr, 0, 0);
        used |= SSI_ATTN_USED_DQ2 | (dq2_items << 8);
    }
    else
        hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(batch * n_kv * (seq / (32 * qpw)))), dim3(64 * ANW), 0, st, (const bf16_t*)qkv,
                           ld, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, doc_start, rope, positions, (int)seq, n_heads, n_kv);
    SSI_LAUNCH_CHECK();
    // dK / dV: the pipelined one-wave-per-SIMD kernel where its shape assumptions hold (256-key groups, an even number of tiles per group);
    // ssi_set_attn_impl(DKV, OLD) keeps the round-1..3 kernel
    // (plain causal rows, or packed rows with a plan; packed rows without one keep the 128-key kernel, whose waves skip the tiles outside
    //  their keys' documents — at B = 2, S = 8192 with documents of 440-1100 tokens the fixed 256-key groups of the plain form, masking
    //  instead of skipping, took 409 us against 329)
    // ... and only where its 256-key workgroups (one per CU at a time) can be balanced over the 256 CUs: the heaviest one walks (S / 32) * rep
    // tiles, the chip's share per CU is the total over 256.  B = 8, S = 2048: 256 against 288; B = 2, S = 2048: 256 against 72 — there the
    // 128-key kernel (two workgroups per CU, half the granularity) is faster.  Mode NEW forces this kernel whatever the balance.
    const int64_t ngrp2 = seq / 256, per0 = seq / 32;
    const int64_t total_tiles = batch * n_kv * rep * (ngrp2 * per0 - 8 * ngrp2 * (ngrp2 - 1) / 2);
    const bool balanced = per0 * rep * 256 <= total_tiles * 23 / 20 || sel == SSI_ATTN_MODE_NEW;
    const bool v2 = !doc_end && seq % 256 == 0 && rep % 4 == 0 && (seq / 32) * rep <= DKV2_MAX_STEPS && balanced && sel != SSI_ATTN_MODE_OLD;
    if (ph && sel != SSI_ATTN_MODE_OLD) {
        const int64_t want = ssi_attn_plan_workspace_bytes(ph);
        if (want > 0 && (!workspace || workspace_bytes < want || ((uintptr_t)workspace & 15))) {
            ssi_set_error("ssi_attn_varlen_bwd_plan: the plan splits %d chunks over the query heads and needs %lld bytes of workspace (got %lld)", ph[13],
                          (long long)want, (long long)workspace_bytes);
            return SSI_ERR_WORKSPACE;
        }
        hipLaunchKernelGGL(attn_bwd_dkv2_kernel<true>, dim3((unsigned)(ph[1] * n_kv)), dim3(256), 0, st, (const bf16_t*)qkv, ld, (const bf16_t*)dout, lse,
                           delta, (bf16_t*)dqkv, rope, positions, (int)seq, n_heads, n_kv, reinterpret_cast<const int4*>(plan_dev + ph[2]),
                           reinterpret_cast<const float*>(plan_dev + 12), (float*)workspace);
        used |= SSI_ATTN_USED_DKV2 | SSI_ATTN_USED_PLAN;
        if (ph[13] > 0) {
            SSI_LAUNCH_CHECK();
            hipLaunchKernelGGL(attn_dkv_plan_reduce_kernel, dim3((unsigned)(ph[13] * n_kv * 32)), dim3(256), 0, st, (const float*)workspace,
                               reinterpret_cast<const int4*>(plan_dev + ph[14]), n_kv, (bf16_t*)dqkv, ld, n_heads, rope, positions, (int)seq);
            used |= SSI_ATTN_USED_HEAD_SPLIT;
        }
    } else if (v2) {
        hipLaunchKernelGGL(attn_bwd_dkv2_kernel<false>, dim3((unsigned)(batch * n_kv * (seq / 256))), dim3(256), 0, st, (const bf16_t*)qkv, ld,
                           (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, rope, positions, (int)seq, n_heads, n_kv, (const int4*)nullptr,
                           (const float*)nullptr, (float*)nullptr);
        used |= SSI_ATTN_USED_DKV2;
    } else {
        // small launches: one workgroup per query head + a reduction, when the caller brought the workspace (mode NO_HEAD_SPLIT: never)
        const int64_t want = ssi_attn_mfma_bwd_workspace_bytes(batch, seq, n_heads, n_kv);
        if (want > 0 && workspace && workspace_bytes >= want && ((uintptr_t)workspace & 15) == 0 && sel != SSI_ATTN_MODE_NO_HEAD_SPLIT) {
            const int slots = dkv_head_slots(batch, seq, n_heads, n_kv);
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, dim3((unsigned)(batch * n_kv * (seq / 128) * slots)), dim3(256), 0, st, (const bf16_t*)qkv, ld,
                               (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, doc_end, rope, positions, (int)seq, n_heads, n_kv, (float*)workspace,
                               rep / slots);
            SSI_LAUNCH_CHECK();
            hipLaunchKernelGGL(attn_dkv_head_reduce_kernel, dim3((unsigned)ssi_cdiv(batch * seq * n_kv * 32, 256)), dim3(256), 0, st,
                               (const float*)workspace, slots, batch * seq, n_kv, (bf16_t*)dqkv, ld, n_heads, rope, positions, (int)seq);
            used |= SSI_ATTN_USED_HEAD_SPLIT;
        } else {
            hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, dim3((unsigned)(batch * n_kv * (seq / 128))), dim3(256), 0, st, (const bf16_t*)qkv, ld,
                               (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, doc_end, rope, positions, (int)seq, n_heads, n_kv, (float*)nullptr, rep);
        }
    }
    SSI_LAUNCH_CHECK();
    ssi_attn_note_dispatch(used | 0x10000);  // bit 16: an MFMA backward ran
    return SSI_OK;
}
