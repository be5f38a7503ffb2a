// placeholder until the MFMA flash attention lands
#include "common.cuh"
bool ssi_attn_mfma_supported(int64_t, int64_t, int64_t, int, int, int, int) { return false; }
int ssi_attn_fwd_mfma(const void*, int64_t, void*, float*, int64_t, int64_t, int, int, void*) { return SSI_ERR_UNSUPPORTED; }
int ssi_attn_bwd_mfma(const void*, int64_t, const void*, const void*, const float*, void*, float*, int64_t, int64_t, int, int, void*) { return SSI_ERR_UNSUPPORTED; }
