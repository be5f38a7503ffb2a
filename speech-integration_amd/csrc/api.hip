// ABI version + thread-local error text for libssi_hip.so.
#include <stdarg.h>
#include "common_hip.h"

static thread_local char g_err[512] = "";

void ssi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ssi_abi_version(void) { return SSI_ABI_VERSION; }
extern "C" const char* ssi_last_error(void) { return g_err; }

// ---- K8 + K9: tied LM head + cross-entropy as ONE entry per direction (SURVEY.md §8b lists `lmhead_ce_{fwd,bwd}` among the exports) -------
// Each is a fixed launch sequence over the kernels above; the logits workspace is the caller's (see the header for why the logits are
// kept in HBM between the two calls instead of being recomputed).
extern "C" int ssi_lmhead_ce_fwd(const void* hidden, int64_t ldh, const void* table, int64_t ldt, const int64_t* labels, int64_t rows,
                                 int64_t dim, int64_t vocab, int64_t vocab_pad, int64_t ignore_index, void* logits_ws, int64_t ldl,
                                 float* row_loss, float* stats, int write_grad, int dtype, void* stream) {
    SSI_CHECK_ARG(hidden && table && labels && logits_ws && row_loss && stats && rows >= 0 && dim > 0 && vocab > 0 && vocab_pad >= vocab &&
                  ldl >= vocab_pad);
    if (rows == 0) return SSI_OK;
    if (int rc = ssi_gemm(SSI_GEMM_NT, rows, vocab_pad, dim, hidden, ldh, table, ldt, logits_ws, ldl, nullptr, 1.f, nullptr, 0, dtype, stream)) return rc;
    if (int rc = ssi_ce_fwd(logits_ws, ldl, labels, rows, vocab, ignore_index, row_loss, nullptr, write_grad, dtype, stream)) return rc;
    return ssi_ce_reduce(row_loss, labels, rows, vocab, ignore_index, stats, stream);
}

extern "C" int ssi_lmhead_ce_bwd(const void* dlogits, int64_t ldl, const void* hidden, int64_t ldh, const void* table, int64_t ldt,
                                 const float* alpha_dev, int64_t rows, int64_t dim, int64_t vocab_pad, void* d_hidden, int64_t lddh,
                                 void* d_table, int64_t lddt, int accumulate_d_table, int dtype, void* stream) {
    SSI_CHECK_ARG(dlogits && hidden && table && d_hidden && d_table && rows >= 0 && dim > 0 && vocab_pad > 0);
    if (rows == 0) return SSI_OK;
    // d_hidden = alpha * dlogits @ E ;  dE (+)= alpha * dlogits^T @ hidden
    if (int rc = ssi_gemm(SSI_GEMM_NN, rows, dim, vocab_pad, dlogits, ldl, table, ldt, d_hidden, lddh, nullptr, 1.f, alpha_dev, 0, dtype, stream)) return rc;
    return ssi_gemm(SSI_GEMM_TN, vocab_pad, dim, rows, dlogits, ldl, hidden, ldh, d_table, lddt, nullptr, 1.f, alpha_dev, accumulate_d_table, dtype, stream);
}
