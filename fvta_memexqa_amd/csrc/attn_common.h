// Shared layout of the focal-attention kernels (forward + backward).
#pragma once
#include "fvta_common.h"

namespace fvta {

// Bilinear form of the similarity logits:
//   x[t,j] = rs[t] * sum_c h[t,c] * Qs[j,c]  +  (Rh.h[t] + R2.h[t]^2)  +  ct[j]
// with Qs[j,c] = U[c] q[j,c] (simi 1-3) or q[j,c]/|q[j]| (simi 4, rs = 1/|h|),
// ct[j] = Cq.q[j] + C2.q[j]^2 + b.
struct AttnShape {
  int N, K, T, JQ, w;
  int simi, feat_order, add_tanh, use_mask;
  int W4;      // w / 4
  int JT;      // 32-wide column tiles covering JQ
  int JP;      // 32 * JT
  int nsplit;  // workgroups per (n,k) in the forward main kernel
  int bsplit;  // workgroups per (n,k) in the backward main kernel
  int gk;      // backward: consecutive k of one n that share a workgroup (and one dQs slab); > 1 only if bsplit == 1
  int ng;      // backward: ceil(K / gk) groups per n
};

inline AttnShape attn_shape(const fvta_attn_desc* d, bool use_mask) {
  AttnShape s;
  s.N = d->N; s.K = d->K; s.T = d->T; s.JQ = d->JQ; s.w = d->w;
  s.simi = d->simi; s.feat_order = d->feat_order; s.add_tanh = d->add_tanh;
  s.use_mask = use_mask ? 1 : 0;
  s.W4 = d->w / 4;
  s.JT = (d->JQ + 31) / 32;
  s.JP = 32 * s.JT;
  const int nk = d->N * d->K;
  int ns = (3072 + nk - 1) / nk;
  const int maxs = (d->T + 63) / 64;
  if (ns > maxs) ns = maxs;
  const int mins = (d->T + 1007) / 1008;  // the 16-row forward kernel keeps a split's row list (<= 1024) in LDS
  if (ns < mins) ns = mins;
  if (ns < 1) ns = 1;
  s.nsplit = ns;
  int bs = (1024 + nk - 1) / nk;
  const int maxb = (d->T + 255) / 256;
  const int minb = (d->T + 959) / 960;  // a backward chunk must fit the 1024-row LDS sort
  if (bs > maxb) bs = maxb;
  if (bs < minb) bs = minb;
  if (bs < 1) bs = 1;
  s.bsplit = bs;
  // short photo streams (the metric shape: 150 rows per (n,k)): one workgroup takes several k of the same n, whose
  // rows are contiguous in [N,K,T,w] -- one slab, one sort, fewer flushes.  ~512 workgroups (2 per CU) when possible.
  int gk = 1;
  if (bs == 1 && d->T <= 512) {
    gk = nk / 512;
    const int cap = 1024 / d->T;  // BWD_CHMAX rows in the LDS sort
    if (gk > cap) gk = cap;
    if (gk > d->K) gk = d->K;
    if (gk > 16) gk = 16;
    if (gk < 1) gk = 1;
  }
  s.gk = gk;
  s.ng = (d->K + gk - 1) / gk;
  return s;
}

#ifdef __HIPCC__
// The softmax logit of a row under time_warp_att: amax * tscale as a ROUNDED fp32 product.  Written as a plain `a * b - m`
// the compiler contracts it into fma(a, b, -m); for masked rows a = -1e30 and m is the (rounded) maximum of the same
// products, so the unrounded product differs from m by ~1e30 * 2^-24 = 6e22 and exp() of that is inf.
// (an opaque v_mul_f32: HIP's __fmul_rn is a plain `x * y` and contracts just the same)
__device__ __forceinline__ float tw_logit(float amax, float tscale) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(amax), "v"(tscale));
  return r;
}
#endif

// per-channel vectors of the bilinear form, float [5][w]: U, Rh, R2, Cq, C2
enum { VEC_U = 0, VEC_RH = 1, VEC_R2 = 2, VEC_CQ = 3, VEC_C2 = 4, VEC_COUNT = 5 };

// "saved" buffer: forward state the backward needs (all device side)
struct AttnSaved {
  float* amax;     // [N,K,T] max_j of the masked (tanh'd) logits  (= a_logits_maxed, model_v2.py:268)
  uint8_t* jmax;   // [N,K,T] first arg-max j
  int32_t* idx;    // [N,K,T] compacted valid row list (identity when the row is fully masked)
  int32_t* cnt;    // [N,K] rows in the list
  int32_t* allmasked;  // [N,K] 1: no valid (t,j) pair -> softmax goes uniform over all T
  float* M;        // [N,K] max_t amax  (= reduce_max [3,2], model_v2.py:278)
  float* Mz;       // [N,K] max_t of the inner softmax's logits z = amax * tscale (time_warp_att, model_v2.py:269-275); = M without it
  float* L;        // [N,K] sum_t exp(z - Mz)
  float* r;        // [N,K] softmax_k(M)
  float* u;        // [N,K,w] inner softsel result
  float* Qs;       // [N][W4][JP][4] pre-scaled question, MFMA-B friendly
  uint16_t* Qh;    // [N][2][W4][JP][4] fp16 split of Qs: hi = rtz(Qs), lo = rtz((Qs - hi) * 2^11)  (16-row forward kernel)
  float* ct;       // [N][JP]
  float* vecs;     // [5][w]
  uint64_t* qvalid;  // [N][JT<=2 -> 2] valid-j bit masks
  size_t bytes;
};

inline AttnSaved attn_saved_view(const AttnShape& s, void* p) {
  FvtaCarver c(p);
  AttnSaved v;
  const size_t nkt = (size_t)s.N * s.K * s.T, nk = (size_t)s.N * s.K;
  v.amax = c.take<float>(nkt);
  v.jmax = c.take<uint8_t>(nkt);
  v.idx = c.take<int32_t>(nkt);
  v.cnt = c.take<int32_t>(nk);
  v.allmasked = c.take<int32_t>(nk);
  v.M = c.take<float>(nk);
  v.Mz = c.take<float>(nk);
  v.L = c.take<float>(nk);
  v.r = c.take<float>(nk);
  v.u = c.take<float>(nk * s.w);
  v.Qs = c.take<float>((size_t)s.N * s.W4 * s.JP * 4);
  v.Qh = c.take<uint16_t>((size_t)s.N * 2 * s.W4 * s.JP * 4);
  v.ct = c.take<float>((size_t)s.N * s.JP);
  v.vecs = c.take<float>((size_t)VEC_COUNT * s.w);
  v.qvalid = c.take<uint64_t>((size_t)s.N * 2);
  v.bytes = c.off;
  return v;
}

}  // namespace fvta
