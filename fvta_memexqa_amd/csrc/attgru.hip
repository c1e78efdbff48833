// AttentionGRUCell (placeholder until the kernels land).
#include "fvta_common.h"
extern "C" int fvta_attgru_fwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                               const float* bg, const float* Wc, const float* Wi, const float* bi, float* new_h,
                               float* saved, fvta_stream_t stream) {
  fvta_set_error("fvta_attgru_fwd: not built yet");
  return FVTA_ERR_UNSUPPORTED;
}
extern "C" int fvta_attgru_bwd(int32_t B, int32_t d, const float* inputs, const float* state, const float* Wg,
                               const float* Wc, const float* Wi, const float* saved, const float* d_new_h,
                               float* d_inputs, float* d_state, float* dWg, float* dbg, float* dWc, float* dWi,
                               float* dbi, void* workspace, fvta_stream_t stream) {
  fvta_set_error("fvta_attgru_bwd: not built yet");
  return FVTA_ERR_UNSUPPORTED;
}
