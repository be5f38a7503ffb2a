/*
 * ssi_hip.h — C ABI of libssi_hip.so: the MI355X (gfx950) kernels under the speech-integration training hot path.
 *
 * The reference (anilkeshwani/speech-integration) reaches its GPU arithmetic only through Python:
 * ssi/trainer.py -> ssi/loss.py -> torchtune 0.5.0 modules -> ATen.  It has no FFI of its own, so the drop-in boundary
 * is the Python API (speech-integration_amd/ssi mirrors it); THIS header is the native seam underneath it, one entry per
 * vendor op the reference hits.  Each entry cites the reference call site it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller (PyTorch's allocator) owns all buffers,
 *     including workspaces; the library allocates nothing persistent and never synchronises the device.
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered.  Compute entries keep no per-call host state and may
 *     be called from any host thread (autograd runs backward on another thread than forward); their one-time set-up (LDS size
 *     attribute, CU count) is done under C++ thread-safe static initialisation.
 *   - PROCESS-GLOBAL state, stated here because it is not per call: (1) ssi_set_impl, (2) ssi_set_gemm_tile_order and (4) ssi_set_attn_impl
 *     are switches for the whole process (atomics; meant to be set once at start-up — tests, A/B runs, the data-parallel trainer — not
 *     flipped while another thread is launching; (4) takes its initial values from the environment variables SSI_ATTN_DQ / SSI_ATTN_DKV,
 *     read ONCE at first use, never on the launch path; the kernel choice moves results at the 1e-4 level — another summation order);
 *     (5) ssi_attn_last_dispatch reports the choice of the most recent attention backward of the process (diagnostic);
 *     (3) with SSI_TILES_DYNAMIC the persistent GEMM draws tiles from 16 scheduler slots in
 *     device memory handed out round-robin per launch, shared by all streams of the process: more than 16 persistent GEMMs in
 *     flight at once on one device would share a slot (the trainer has at most two).
 *   - return 0 on success; SSI_ERR_* otherwise (never throws).  ssi_last_error() gives a thread-local message.
 *   - dtype: SSI_F32 or SSI_BF16 selects the storage type of activations/weights; reductions, softmax, norms and
 *     accumulators are always fp32 (the reference's rounding points).
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef SSI_HIP_H
#define SSI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSI_ABI_VERSION 8 /* 8: + ssi_ce_fwd_weighted (per-row loss weights: an accumulation window run as ONE batch keeps the reference's
                           *    per-micro-batch normalisation, ssi/data/window.py)
                           * 7: + ssi_set_attn_impl, ssi_attn_last_dispatch (no getenv on the launch path), ssi_attn_plan_* and ssi_attn_varlen_bwd_plan
                           *    (packed rows on the pipelined backward kernels, host-built work plan)
                           * 6: + ssi_attn_bwd_workspace_bytes, ssi_attn_varlen_bwd_ws (attention backward with a caller-owned workspace)
                           * 5: ssi_doc_ranges takes n_clamped (out-of-table positions are counted, not only clamped)
                           * 4: + ssi_gemm_batched
                           * 2: + ssi_attn_varlen_fwd/bwd(_rope), ssi_set_gemm_tile_order, NN form of ssi_gemm_swiglu_bwd
                           * 3: ssi_ce_reduce takes `vocab` and reports out-of-range labels in out[3]; + ssi_doc_ranges; ssi_rmsnorm_bwd takes accumulate_dscale; + ssi_lmhead_ce_fwd/bwd */

enum { SSI_F32 = 0, SSI_BF16 = 1 };
enum { SSI_OK = 0, SSI_ERR_ARG = 1, SSI_ERR_UNSUPPORTED = 2, SSI_ERR_WORKSPACE = 3, SSI_ERR_HIP = 1000 /* + hipError_t */ };
/* GEMM operand layouts (row-major storage):
 *   NT: A[M,K]  B[N,K]   C = A * B^T   forward linear  y = x W^T      (F.linear, torchtune nn.Linear / TiedLinear)
 *   NN: A[M,K]  B[K,N]   C = A * B     data gradient   dx = dy W
 *   TN: A[K,M]  B[K,N]   C = A^T * B   weight gradient dW = dy^T x                                                  */
enum { SSI_GEMM_NT = 0, SSI_GEMM_NN = 1, SSI_GEMM_TN = 2 };
/* which implementation ssi_gemm may use: AUTO picks the MFMA kernel when shape/dtype allow, else the generic one */
/* Order in which the workgroups of the persistent MFMA GEMM take output tiles.  STATIC (default): tile b, b+G, ... — optimal
 * when the GEMM has the GPU to itself.  DYNAMIC: tiles are drawn from per-XCD counters, so a workgroup that starts late because
 * another kernel holds its CU (RCCL during the data-parallel gradient exchange) costs its share of tiles instead of a round. */
enum { SSI_TILES_STATIC = 0, SSI_TILES_DYNAMIC = 1 };
int ssi_set_gemm_tile_order(int mode);

enum { SSI_IMPL_AUTO = 0, SSI_IMPL_GENERIC = 1, SSI_IMPL_MFMA = 2,
       SSI_IMPL_MFMA_WG8 = 3 /* debug / A-B runs: MFMA paths, but the NT GEMM restricted to the 8-wave LDS-DMA kernel */ };

int ssi_abi_version(void);
const char* ssi_last_error(void);
/* force an implementation for kernels that have both a generic and an MFMA path (tests); returns previous value */
int ssi_set_impl(int impl);

/* ---- K1  nn.Embedding (torchtune TransformerDecoder.tok_embeddings; called from ssi/loss.py:8) ------------------- */
/* out[t,:] = table[tokens[t],:] */
int ssi_embed_fwd(const int64_t* tokens, const void* table, void* out, int64_t n_tok, int64_t dim, int64_t vocab,
                  int dtype, void* stream);
/* dtable[v,:] += sum_{t: tokens[t]==v} dout[t,:]   (deterministic: sorted segmented sum, fp32 accumulate, no atomics) */
int64_t ssi_embed_bwd_workspace_bytes(int64_t vocab);
int ssi_embed_bwd(const int64_t* tokens, const void* dout, void* dtable, int64_t n_tok, int64_t dim, int64_t vocab,
                  int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- K2  RMSNorm (torchtune.modules.RMSNorm; sa_norm / mlp_norm / norm) ------------------------------------------- */
/* y = (x32 * rsqrt(mean(x32^2)+eps)).to(dtype) * scale ; rstd[row] saved for backward */
int ssi_rmsnorm_fwd(const void* x, const void* scale, void* y, float* rstd, int64_t rows, int64_t dim, float eps,
                    int dtype, void* stream);
/* dx = d(rmsnorm)/dx . dy (+ dres if non-null);  dscale (+)= sum_rows dy * xhat  (two-stage deterministic reduce; accumulate_dscale
 * == 0 overwrites: the first micro-batch of an accumulation window writes the gradients, nothing zeroes them beforehand).
 * partials: fp32 workspace of ssi_rmsnorm_bwd_workspace_bytes(rows, dim) bytes. */
int64_t ssi_rmsnorm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int ssi_rmsnorm_bwd(const void* dy, const void* x, const void* scale, const float* rstd, const void* dres, void* dx,
                    void* dscale, int accumulate_dscale, int64_t rows, int64_t dim, int dtype, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* ---- K4  Llama3ScaledRoPE (adjacent-pair rotation, fp32 math; torchtune MultiHeadAttention.pos_embeddings) -------- */
/* x: [rows, ld] rows = b*seq_len + s; rotates heads [0, n_heads_rot) of width head_dim in place; position of a row is
 * positions[row] if positions != NULL else row % seq_len.  table: [table_len, head_dim/2, 2] fp32 (cos, sin).
 * inverse != 0 applies the transpose rotation (the backward of the forward rotation). */
int ssi_rope_inplace(void* x, int64_t ld, int64_t rows, int64_t seq_len, int n_heads_rot, int head_dim,
                     const float* table, int64_t table_len, const int32_t* positions, int inverse, int dtype,
                     void* stream);

/* C[M, N] = A[M, K] B[N, K]^T followed by ssi_rope_inplace(C, heads [0, n_heads_rot)): the QKV projection (K3) with the rotation
 * (K4) in the GEMM epilogue on the MFMA path (bf16, head_dim 64, n_heads_rot * 64 a multiple of 256), two launches otherwise. */
int ssi_gemm_rope(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                  int64_t seq_len, int n_heads_rot, int head_dim, const float* rope_table, int64_t table_len,
                  const int32_t* positions, int dtype, void* stream);

/* ---- K5  causal GQA attention (F.scaled_dot_product_attention(is_causal=True) inside torchtune MultiHeadAttention) -- */
/* qkv: [B*S, ld] with q heads at column 0, k heads at n_heads*head_dim, v heads at (n_heads+n_kv)*head_dim.
 * out: [B*S, n_heads*head_dim].  lse: [B, n_heads, S] fp32 (natural-log sum-exp of scaled scores). */
int ssi_attn_fwd(const void* qkv, int64_t ld, void* out, float* lse, int64_t batch, int64_t seq, int n_heads,
                 int n_kv, int head_dim, int dtype, void* stream);
/* dqkv (same layout as qkv) is fully overwritten. delta: fp32 workspace [B, n_heads, S]. */
int ssi_attn_bwd(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                 float* delta, int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype,
                 void* stream);

/* Packed rows (several documents per row; SURVEY.md §8f rank 1, the block-causal mask torchtune's padded_collate_packed builds
 * from `seq_lens`, reference stub ssi/data/__init__.py:66-73,202-205): doc_start / doc_end are int32 [B*S]; for position s of
 * row b, doc_start[b*S+s] is the first position and doc_end[b*S+s] one past the last position of the document holding s (both
 * relative to the row, non-decreasing along it).  A query sees the keys doc_start <= key <= query.  Both NULL = plain causal
 * attention (= ssi_attn_fwd / ssi_attn_bwd).  Key tiles outside the documents of a query block are skipped, not masked. */
int ssi_attn_varlen_fwd(const void* qkv, int64_t ld, void* out, float* lse, const int32_t* doc_start, const int32_t* doc_end,
                        int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype, void* stream);
int ssi_attn_varlen_bwd(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                        float* delta, const int32_t* doc_start, const int32_t* doc_end, int64_t batch, int64_t seq,
                        int n_heads, int n_kv, int head_dim, int dtype, void* stream);

/* input_pos [batch, seq] (int64; restarts at 0 with every document of a packed row) -> the int32 [batch*seq] arrays the varlen entries
 * take: positions (clamped to [0, max_pos]: the RoPE table is never indexed past its end), doc_start, doc_end.  Position 0 of a row
 * always starts a document.  n_clamped (may be NULL): one int32 the kernel ADDS the number of positions outside [0, max_pos] to — the
 * caller zeroes it and reads it with whatever it reads back anyway, so that a data bug is reported instead of silently clamped.
 * One launch (the reference has no packed path: ssi/data/__init__.py:66-73 raises). */
int ssi_doc_ranges(const int64_t* input_pos, int64_t batch, int64_t seq, int64_t max_pos, int32_t* positions, int32_t* doc_start,
                   int32_t* doc_end, int32_t* n_clamped, void* stream);

/* Same, followed by the backward of the RoPE rotation on the q and k heads of dqkv (= ssi_rope_inplace(dqkv, ..., inverse=1)): the
 * MFMA kernels apply it in their epilogues, which saves a pass over dqkv.  rope_table / table_len / positions as in ssi_rope_inplace. */
int ssi_attn_varlen_bwd_rope(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                             float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                             int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                             int head_dim, int dtype, void* stream);

/* The same backward with a caller-owned workspace (any of the forms above: doc_start / doc_end and rope_table may be NULL).
 * ssi_attn_bwd_workspace_bytes: what this shape can use, 0 = nothing.  With at least that many bytes (16-byte aligned), launches whose
 * workgroups cannot fill the chip — the reference's default micro-batch of 2 x 2048 rows gives dK / dV 256 workgroups, the heaviest as long as
 * the launch — run dK / dV as one workgroup per query head with fp32 partial rows in the workspace and a reduction over the heads in fixed
 * order (reproducible; the sums over the heads are taken in another order than without workspace). */
int64_t ssi_attn_bwd_workspace_bytes(int64_t batch, int64_t seq, int n_heads, int n_kv, int head_dim, int dtype);
int ssi_attn_varlen_bwd_ws(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                           float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                           int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                           int head_dim, int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* Which kernels the attention backward may take (process-global, see "PROCESS-GLOBAL state" above).  `which`: SSI_ATTN_KERNEL_DQ or
 * SSI_ATTN_KERNEL_DKV.  `mode`: AUTO = the dispatcher's choice by shape; OLD = the round-1..3 kernels (attn_bwd_dq_kernel / the 128-key
 * attn_bwd_dkv_kernel); NEW = the pipelined one-wave-per-SIMD kernels wherever their shape rules allow, whatever the fill / balance;
 * NO_HEAD_SPLIT (dK / dV only) = AUTO without the per-query-head split of small launches.  Returns the previous mode (an invalid `mode`
 * only reads it; an invalid `which` returns -1).  Initial values: environment SSI_ATTN_DQ / SSI_ATTN_DKV = 0..3, read once. */
enum { SSI_ATTN_KERNEL_DQ = 0, SSI_ATTN_KERNEL_DKV = 1 };
enum { SSI_ATTN_MODE_AUTO = 0, SSI_ATTN_MODE_OLD = 1, SSI_ATTN_MODE_NEW = 2, SSI_ATTN_MODE_NO_HEAD_SPLIT = 3 };
int ssi_set_attn_impl(int which, int mode);
/* Kernels the most recent MFMA attention backward of this process launched (diagnostic; tests assert the dispatch with it, the secondary
 * bench lines name the path a number came from): bit 0 dQ pipelined (attn_bwd_dq2_kernel), bit 1 dK/dV pipelined (attn_bwd_dkv2_kernel),
 * bit 2 dK/dV split over the query heads, bit 3 the document-aware (plan) forms of the pipelined kernels; bits 8-11 query blocks per
 * persistent dQ workgroup.  0 before any call and after a call that took the generic kernels. */
enum { SSI_ATTN_USED_DQ2 = 1, SSI_ATTN_USED_DKV2 = 2, SSI_ATTN_USED_HEAD_SPLIT = 4, SSI_ATTN_USED_PLAN = 8 };
int ssi_attn_last_dispatch(void);

/* Work plan for packed rows on the pipelined backward kernels (the Trainer's default path: ssi/data/unpad.py turns every right-padded batch of
 * the reference's collate function, ssi/data/__init__.py:139-199, into one packed row; plans/Feature - Packed Dataset Support.md:1-96).  The
 * document bounds are known on the HOST wherever batches are collated, so the plan is built there (no device sync, in the prefetch thread)
 * and travels to the device with the batch: every document is cut into dK/dV items (up to 256 keys of ONE document) and dQ items (a
 * 64-query block of ONE document), sorted by work, the dQ items dealt to persistent workgroups of equal load.  Tiles outside an item's
 * document are not in its tile list at all (skipped, not masked); the heaviest items start first whatever their place in the row.
 *   host_doc_row / _start / _end: n_docs documents as (row b, first position, one past the last position) — they must tile every row.
 *   flags: SSI_ATTN_PLAN_FORCE = build the plan even where the round-1..3 kernels are expected to be faster (tests).
 *   ssi_attn_plan_words: upper bound of a plan's size in int32 words.  ssi_attn_plan_build: words written (> 0); 0 = the pipelined kernels
 *   do not take this batch (other than 4 query heads per kv head, a document longer than 16 384 tokens, mostly tiny documents) — pass no
 *   plan then; < 0 = bad arguments.  The first SSI_ATTN_PLAN_HEADER words are the header the launch
 *   reads on the host.  RoPE positions are taken to be (position - document start): build no plan for batches whose input_pos does
 *   anything else. */
enum { SSI_ATTN_PLAN_HEADER = 16, SSI_ATTN_PLAN_FORCE = 1, SSI_ATTN_PLAN_SPLIT_ALL = 2 /* tests: every dK/dV chunk split over the query heads */ };
int64_t ssi_attn_plan_words(int64_t batch, int64_t seq, int64_t n_docs);
/* dK/dV chunks whose work exceeds the chip's share per compute unit (long documents) are split over the query heads: fp32 partial sums in
 * the caller's workspace, added by a reduction pass over those chunks only.  ssi_attn_plan_workspace_bytes: what ssi_attn_varlen_bwd_plan needs
 * in `workspace` for this plan (0: nothing was split; -1: not a plan header). */
int64_t ssi_attn_plan_workspace_bytes(const int32_t* host_plan_header);
int64_t ssi_attn_plan_build(const int32_t* host_doc_row, const int32_t* host_doc_start, const int32_t* host_doc_end, int64_t n_docs,
                            int64_t batch, int64_t seq, int n_heads, int n_kv, int flags, int32_t* host_plan, int64_t plan_words);
/* ssi_attn_varlen_bwd_ws + a plan: plan = the plan in DEVICE memory, host_plan_header = its first SSI_ATTN_PLAN_HEADER words on the host
 * (both NULL = ssi_attn_varlen_bwd_ws).  doc_start / doc_end stay required for packed rows (kernels a mode switch sends back to the
 * round-1..3 forms read them); plain causal rows (doc_start = doc_end = positions = NULL) may bring a plan whose documents are the rows
 * themselves — the persistent dQ workgroups then get their query blocks by load instead of by a fixed pattern.  Results equal the plan-less
 * call's to the rounding of another summation order; reproducible for a given plan. */
int ssi_attn_varlen_bwd_plan(const void* qkv, int64_t ld, const void* out, const void* dout, const float* lse, void* dqkv,
                             float* delta, const int32_t* doc_start, const int32_t* doc_end, const float* rope_table,
                             int64_t table_len, const int32_t* positions, int64_t batch, int64_t seq, int n_heads, int n_kv,
                             int head_dim, int dtype, void* workspace, int64_t workspace_bytes, const int32_t* plan,
                             const int32_t* host_plan_header, void* stream);

/* ---- K7  SwiGLU elementwise (torchtune FeedForward: w2(silu(w1 x) * w3 x)) ------------------------------------------ */
/* gu: [rows, 2*inter] = [gate | up]; act[rows, inter] = silu(gate) * up */
int ssi_swiglu_fwd(const void* gu, void* act, int64_t rows, int64_t inter, int dtype, void* stream);
int ssi_swiglu_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int64_t inter, int dtype, void* stream);

/* ---- K3/K6/K7/K8/K10  dense GEMMs (F.linear and its autograd) ------------------------------------------------------- */
/* C = (accumulate ? C : 0) + alpha * (alpha_dev ? *alpha_dev : 1) * op(A) op(B) + (R ? R : 0)
 * A, B, C, R share `dtype`; accumulation is fp32.  R (residual) has C's shape and ldc.  The MFMA path needs
 * dtype == SSI_BF16, M % 256 == 0, N % 256 == 0, K % 64 == 0 and 16-byte aligned rows; otherwise the generic path runs
 * (or SSI_ERR_UNSUPPORTED if SSI_IMPL_MFMA was forced). */
int ssi_gemm(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
             void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev, int accumulate, int dtype,
             void* stream);

/* Split-K variant for long-K / small-output contractions (weight gradients): `splits` K-slices into fp32 slabs in
 * `workspace` (ssi_gemm_splitk_workspace_bytes), then one reduction pass with the same epilogue as ssi_gemm. */
int64_t ssi_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits);
int ssi_gemm_splitk(int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int64_t ldc, const void* R, float alpha, const float* alpha_dev, int accumulate, int dtype,
                    int splits, void* workspace, int64_t workspace_bytes, void* stream);

/* `batch` GEMMs of one shape in ONE launch: problem b works on A + b strideA, B + b strideB, C + b strideC (strides in elements, no
 * residual).  For contractions whose output grid cannot fill the 256 CUs but which come in independent copies: the weight gradients
 * of the square projections (F.linear's autograd for attn.output_proj: 64 output tiles, fused q/k/v_proj: 96) of several LAYERS,
 * which the model defers until a group of layers has finished its backward — at full K, with no fp32 partials and no reduction pass.
 * MFMA path: SSI_GEMM_TN, bf16, ssi_gemm's shape rules, K % 128 == 0; anything else runs as `batch` ssi_gemm calls (same results). */
int ssi_gemm_batched(int layout, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t strideA, const void* B,
                     int64_t ldb, int64_t strideB, void* C, int64_t ldc, int64_t strideC, float alpha, const float* alpha_dev,
                     int accumulate, int dtype, void* stream);

/* Fused SwiGLU GEMMs (torchtune FeedForward and its autograd): the elementwise stage rides in the GEMM epilogue on the MFMA
 * path (bf16, M % 256 == 0, I % 256 == 0, K % 64 == 0); otherwise the unfused kernels run.  Same rounding points either way.
 *   fwd: GU[M, 2I] = X[M, K] W13[2I, K]^T  (W13 = [gate rows | up rows]),  ACT[M, I] = silu(gate) * up
 *   bwd: DGU[M, 2I] = swiglu_backward(DY[M, K] W2, GU);  W2 as [I, K] (layout SSI_GEMM_NT) or [K, I] (SSI_GEMM_NN);
 *        dact_ws: [M, I] scratch used only by the unfused fallback. */
int ssi_gemm_swiglu_fwd(int64_t M, int64_t inter, int64_t K, const void* X, int64_t ldx, const void* W13, int64_t ldw, void* GU,
                        int64_t ldgu, void* ACT, int64_t ldact, int dtype, void* stream);
int ssi_gemm_swiglu_bwd(int layout, int64_t M, int64_t inter, int64_t K, const void* DY, int64_t lddy, const void* W2, int64_t ldw,
                        const void* GU, int64_t ldgu, void* DGU, int64_t lddgu, void* dact_ws, int dtype, void* stream);

/* dst[c][r] = src[r][c] for a [rows, cols] matrix (both dims multiples of 8).  Keeps [in, out] copies of the projection
 * weights so that the data-gradient GEMMs use the k-contiguous operand form (refreshed once per optimizer step). */
int ssi_transpose(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols, int dtype,
                  void* stream);

/* ---- K9  CEWithChunkedOutputLoss (ssi/trainer.py:300; F.cross_entropy(logits.float(), reduction="sum")) -------------- */
/* logits: [rows, ld] (columns [0, vocab) are real, [vocab, ld) padding).  labels: already shifted (ssi/loss.py:16).
 * row_loss[r] = lse(logits[r]) - logits[r, label] (0 if ignored).  If write_grad: logits[r, :] is OVERWRITTEN by
 * softmax(logits[r]) - onehot(label) (0 for ignored rows and for padding columns) — the unscaled d(sum NLL)/dlogits. */
int ssi_ce_fwd(void* logits, int64_t ld, const int64_t* labels, int64_t rows, int64_t vocab, int64_t ignore_index,
               float* row_loss, float* row_lse, int write_grad, int dtype, void* stream);
/* bf16 rows of up to 196 608 columns are held in registers by one workgroup per CU: the log-sum-exp and the gradient come from ONE
 * read of the row (2 passes over the logits, not 3).  A label that is neither `ignore_index` nor inside [0, vocab) gives a zero
 * loss and a zero gradient row here and is COUNTED by ssi_ce_reduce (out[3]): the caller raises when it next reads back. */
/* out[0] = sum(row_loss) / n_valid (NaN if n_valid == 0, as the reference), out[1] = sum(row_loss), out[2] = n_valid (labels that
 * are not ignored and lie in [0, vocab) — the predicate ssi_ce_fwd uses), out[3] = number of out-of-range labels.  out: 4 floats. */
int ssi_ce_reduce(const float* row_loss, const int64_t* labels, int64_t rows, int64_t vocab, int64_t ignore_index, float* out,
                  void* stream);
/* ssi_ce_fwd with a weight per row (NULL: all 1, and then bit-identical to ssi_ce_fwd): row_loss[r] = w[r] (lse - logit[label]), gradient row
 * w[r] (softmax - onehot); w >= 0.  The reference normalises every micro-batch of an accumulation window by ITS OWN count of shifted labels and
 * weights it by its own count of unshifted ones (ssi/trainer.py:385-395 with ssi/loss.py:16-22); a window that runs as one batch carries that
 * ratio per row here.  In the register-resident bf16 kernel the weight is an additive term of the exponent: no cost per element. */
int ssi_ce_fwd_weighted(void* logits, int64_t ld, const int64_t* labels, const float* row_weight, int64_t rows, int64_t vocab,
                        int64_t ignore_index, float* row_loss, float* row_lse, int write_grad, int dtype, void* stream);

/* ---- K8 + K9 as one entry per direction: tied LM head (TiedLinear over tok_embeddings, ssi/loss.py:8-14) + chunked CE (trainer.py:300) ----
 * fwd: logits_ws[rows, vocab_pad] = hidden[rows, dim] table[vocab_pad, dim]^T (table rows >= vocab are zero padding), then ssi_ce_fwd
 *      on it (write_grad: logits_ws becomes softmax - onehot in place) and ssi_ce_reduce into stats[4].  Equals the reference's
 *      CEWithChunkedOutputLoss(model(tokens), shifted_labels) for any chunk count.
 * bwd: d_hidden = alpha * dlogits table;  d_table (+)= alpha * dlogits^T hidden   (alpha_dev: device scalar = upstream grad / n_valid).
 * The [rows, vocab_pad] logits stay in HBM between the two calls (the caller's workspace, 4.37 GB at T = 16 384): recomputing the logit
 * tiles in backward instead costs another 2 rows vocab_pad dim flop (8.95 TFLOP = 6 ms per step at 1.5 PFLOP/s) to save the 1.8 ms the
 * register-resident CE kernel takes — measured and decided in DESIGN.md §4. */
int ssi_lmhead_ce_fwd(const void* hidden, int64_t ldh, const void* table, int64_t ldt, const int64_t* labels, int64_t rows,
                      int64_t dim, int64_t vocab, int64_t vocab_pad, int64_t ignore_index, void* logits_ws, int64_t ldl,
                      float* row_loss, float* stats, int write_grad, int dtype, void* stream);
int ssi_lmhead_ce_bwd(const void* dlogits, int64_t ldl, const void* hidden, int64_t ldh, const void* table, int64_t ldt,
                      const float* alpha_dev, int64_t rows, int64_t dim, int64_t vocab_pad, void* d_hidden, int64_t lddh,
                      void* d_table, int64_t lddt, int accumulate_d_table, int dtype, void* stream);

/* ---- K14 count_token_types + valid-label count (ssi/train_utils.py:150-165, ssi/trainer.py:388,391) ---------------- */
/* ranges: n_ranges inclusive [lo, hi] pairs (device int64).  out[0..n_ranges) = per-range counts,
 * out[n_ranges] = count(tokens != pad_id), out[n_ranges+1] = count(labels != ignore_index).  One launch, no host sync. */
int ssi_count_tokens(const int64_t* tokens, const int64_t* labels, int64_t n, const int64_t* ranges, int n_ranges,
                     int64_t pad_id, int64_t ignore_index, int64_t* out, void* stream);

/* ---- K11-K13 scale_grads / clip_grad_norm_ / AdamW (ssi/trainer.py:404-409, ssi/optimizer.py:8-17) ----------------- */
int ssi_scale_inplace(void* x, int64_t n, float scale, const float* scale_dev, int dtype, void* stream);
/* out[0] = sum(x^2) in fp32, deterministic two-stage; workspace >= ssi_sumsq_workspace_bytes(n) */
int64_t ssi_sumsq_workspace_bytes(int64_t n);
int ssi_sumsq(const void* x, int64_t n, int dtype, float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* Decoupled-weight-decay Adam on flat buffers, fp32 op-math, one rounding on store (torch fused AdamW semantics):
 *   g = grad * (grad_scale_dev ? *grad_scale_dev : 1);  p -= lr*wd*p;  m = m + (1-b1)(g-m);  v = b2 v + (1-b2) g g;
 *   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),  bc1 = 1-b1^step, bc2 = 1-b2^step.  zero_grad bit 0: grad <- 0; bit 1 (ABI v7): the
 *   launch changes nothing when *grad_scale_dev is inf or NaN (1 / 0 label tokens of an accumulation window: the reference skips that window's
 *   optimizer step, ssi/trainer.py:399-403; an update issued before the host has read the count back skips by itself). */
int ssi_adamw_step(void* param, void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, const float* grad_scale_dev, int zero_grad,
                   int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSI_HIP_H */
