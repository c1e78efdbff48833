// Library-level entry points: version, per-thread error string, and the opt-in
// HIP-event profiling hook bench.py uses to time kernels live on the launch stream.
#include <stdarg.h>

#include <vector>

#include "fvta_common.h"
#include "fvta_prof.h"

static thread_local char g_err[512] = "";

void fvta_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int fvta_version(void) { return 100; }
extern "C" const char* fvta_last_error(void) { return g_err; }

// sizeof of the descriptor structs as THIS library was compiled: a binding (ctypes, cgo, JNI) checks its own layout
// against it at load time instead of corrupting a call silently
extern "C" int64_t fvta_abi_struct_bytes(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(fvta_attn_desc);
    case 1: return (int64_t)sizeof(fvta_lstm_desc);
    case 2: return (int64_t)sizeof(fvta_scorer_desc);
    case 3: return (int64_t)sizeof(fvta_timewarp_desc);
    case 4: return (int64_t)sizeof(fvta_embed_desc);
    case 5: return (int64_t)sizeof(fvta_imgtrans_desc);
    default: return -1;
  }
}

// ---- profiling brackets ------------------------------------------------------
namespace {
struct Bracket {
  hipEvent_t a, b;
  int id;
  int launches;
};
thread_local bool g_on = false;
thread_local std::vector<Bracket> g_open;
}  // namespace

void fvta_prof_begin(int id, hipStream_t s) {
  if (!g_on) return;
  Bracket br;
  if (hipEventCreate(&br.a) != hipSuccess || hipEventCreate(&br.b) != hipSuccess) return;
  br.id = id;
  br.launches = 0;
  (void)hipEventRecord(br.a, s);
  g_open.push_back(br);
}

void fvta_prof_end(int id, int launches, hipStream_t s) {
  if (!g_on) return;
  for (size_t i = g_open.size(); i-- > 0;)
    if (g_open[i].id == id && g_open[i].launches == 0) {
      g_open[i].launches = launches;
      (void)hipEventRecord(g_open[i].b, s);
      return;
    }
}

extern "C" int fvta_profile_enable(int32_t on) {
  g_on = on != 0;
  return FVTA_OK;
}

// Sum of elapsed ms and launch count of every closed bracket of `id`; consumes them.
namespace fvta { int wreg_read_stamp(int i, long long* v); }  // lstm_wreg.hip (diagnostics, -DFVTA_WREG_STAMP builds)
namespace fvta { int wreg_set_mode(int mode); }  // lstm_wreg.hip

extern "C" int fvta_lstm_kernel_select(int32_t mask) { return fvta::wreg_set_mode(mask); }

namespace fvta { extern long long g_bwd_step_counts[3]; }  // lstm_wreg_bwd.hip
extern "C" int fvta_lstm_bwd_kernel_counts(int64_t* counts) {
  FVTA_CHECK_ARG(counts, "lstm_bwd_kernel_counts: null pointer");
  for (int i = 0; i < 3; ++i) {
    counts[i] = fvta::g_bwd_step_counts[i];
    fvta::g_bwd_step_counts[i] = 0;
  }
  return FVTA_OK;
}

extern "C" int fvta_profile_collect(int32_t id, double* total_ms, int64_t* launches) {
  FVTA_CHECK_ARG(total_ms && launches, "profile_collect: null pointer");
  if (id >= 200000) {  // diagnostics: shader-clock stamp id - 200000 of lstm_fwd_wreg_bf16 (tools/r03_wreg_stamps.py)
    long long v = 0;
    const int e = fvta::wreg_read_stamp(id - 200000, &v);
    *total_ms = 0;
    *launches = v;
    return e;
  }
  double ms = 0;
  int64_t n = 0;
  std::vector<Bracket> keep;
  for (auto& br : g_open) {
    if (br.id != id || br.launches == 0) {
      keep.push_back(br);
      continue;
    }
    float t = 0.f;
    (void)hipEventSynchronize(br.b);
    if (hipEventElapsedTime(&t, br.a, br.b) == hipSuccess) {
      ms += t;
      n += br.launches;
    }
    (void)hipEventDestroy(br.a);
    (void)hipEventDestroy(br.b);
  }
  g_open.swap(keep);
  *total_ms = ms;
  *launches = n;
  return FVTA_OK;
}

// ---- achievable-HBM probe (bench.py): a read-only stream over `bytes` of device memory, 16 B per lane, fully
// coalesced, non-temporal -- the rate the roofline fractions can be held against next to the nominal 8 TB/s.
namespace fvta {
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const probe_f32x4* __restrict__ p, size_t n16, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  probe_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    probe_f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345678e30f) sink[0] = acc[0];  // never true: keeps the loads alive
}
// read `nr` streams and write `nw` streams of n16 float4 each, all out of / into one buffer (stream k at base + k * n16):
// the achievable rate of a MIXED read/write stream set, which is what the LSTM step epilogues are (5 reads : 2 writes)
__global__ __launch_bounds__(256) void hbm_mix_probe_kernel(probe_f32x4* __restrict__ base, size_t n16, int nr, int nw) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    probe_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nr; ++k) acc += __builtin_nontemporal_load(base + (size_t)k * n16 + i);
    for (int k = 0; k < nw; ++k) __builtin_nontemporal_store(acc, base + (size_t)(nr + k) * n16 + i);
  }
}
// one wave that waits `ticks` of the 100 MHz wall clock: two of these on two streams finish in one wait when the
// streams own different hardware queues and in two when they share one (fvta_probe_spin)
__global__ void spin_probe_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace fvta

extern "C" int fvta_probe_spin(int64_t microseconds, fvta_stream_t stream) {
  FVTA_CHECK_ARG(microseconds > 0 && microseconds <= 100000, "probe_spin: 1..100000 us");
  hipLaunchKernelGGL(fvta::spin_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  FVTA_CHECK_LAUNCH("spin_probe");
  return FVTA_OK;
}

extern "C" int fvta_probe_hbm_mix(void* buf, size_t bytes_per_stream, int32_t nread, int32_t nwrite, fvta_stream_t stream) {
  FVTA_CHECK_ARG(buf && bytes_per_stream >= 16 && nread >= 0 && nwrite >= 0 && nread + nwrite > 0, "probe_hbm_mix: bad arguments");
  hipLaunchKernelGGL(fvta::hbm_mix_probe_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                     (fvta::probe_f32x4*)buf, bytes_per_stream / 16, nread, nwrite);
  FVTA_CHECK_LAUNCH("hbm_mix_probe");
  return FVTA_OK;
}

extern "C" int fvta_probe_hbm_read(const void* buf, size_t bytes, float* sink, fvta_stream_t stream) {
  FVTA_CHECK_ARG(buf && sink && bytes >= 16, "probe_hbm_read: bad arguments");
  hipLaunchKernelGGL(fvta::hbm_read_probe_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream,
                     (const fvta::probe_f32x4*)buf, bytes / 16, sink);
  FVTA_CHECK_LAUNCH("hbm_read_probe");
  return FVTA_OK;
}
