// Error reporting + the optional HIP-event profiler shared by every entry point of libcat_hip.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace cat {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return -5;
  }
  return 0;
}

// ---- profiler: one (start, stop) event pair per instrumented entry-point call, on the caller's stream ----------
struct ProfRec {
  const char* fam;
  double flops, bytes;
  hipEvent_t a, b;
};
struct ProfAgg {
  std::string name;
  int64_t count;
  double ms, flops, bytes;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;
static std::vector<ProfAgg> g_aggs;

ProfScope::ProfScope(const char* fam, double flops, double bytes, void* stream) : active_(g_prof_on), stream_(stream), idx_(-1) {
  if (!active_) return;
  ProfRec r{fam, flops, bytes, nullptr, nullptr};
  hipEventCreate(&r.a);
  hipEventCreate(&r.b);
  hipEventRecord(r.a, (hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  idx_ = (int)g_recs.size();
  g_recs.push_back(r);
}
ProfScope::~ProfScope() {
  if (!active_ || idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_recs[idx_].b, (hipStream_t)stream_);
}
}  // namespace cat

extern "C" {
const char* cat_hip_last_error(void) { return cat::g_err; }
int cat_hip_version(void) { return 2; }

void cat_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(cat::g_prof_mu);
  cat::g_prof_on = on != 0;
}

// Synchronise, fold all records into per-family aggregates, free the events.  Returns the number of families.
int cat_prof_collect(void) {
  hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(cat::g_prof_mu);
  std::map<std::string, cat::ProfAgg> m;
  for (auto& r : cat::g_recs) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    auto& a = m[r.fam];
    a.name = r.fam;
    a.count += 1;
    a.ms += ms;
    a.flops += r.flops;
    a.bytes += r.bytes;
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  cat::g_recs.clear();
  cat::g_aggs.clear();
  for (auto& kv : m) cat::g_aggs.push_back(kv.second);
  return (int)cat::g_aggs.size();
}

int cat_prof_family(int i, char* name, int cap, int64_t* count, double* ms, double* flops) {
  if (i < 0 || i >= (int)cat::g_aggs.size()) return -22;
  const auto& a = cat::g_aggs[i];
  snprintf(name, cap, "%s", a.name.c_str());
  *count = a.count;
  *ms = a.ms;
  *flops = a.flops;
  return 0;
}
double cat_prof_family_bytes(int i) { return (i < 0 || i >= (int)cat::g_aggs.size()) ? 0.0 : cat::g_aggs[i].bytes; }
}
