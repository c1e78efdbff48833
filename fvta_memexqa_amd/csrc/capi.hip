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
namespace fvta { int lstm_read_stamp(int i, long long* v); }  // lstm_bf16.hip (diagnostics)

extern "C" int fvta_profile_collect(int32_t id, double* total_ms, int64_t* launches) {
  FVTA_CHECK_ARG(total_ms && launches, "profile_collect: null pointer");
  if (id >= 100000) {  // diagnostics: shader-clock stamp id - 100000 of the LSTM step kernel (tools/lstm_phases.py)
    long long v = 0;
    const int e = fvta::lstm_read_stamp(id - 100000, &v);
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
