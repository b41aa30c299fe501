// Optional in-library launch timing: HIP events recorded on the launch stream around every conv launch,
// aggregated per kernel instantiation.  Used by bench.py for the live roofline figure (the same per-kernel
// average that `rocprofv3 --kernel-trace --stats` reports).  Off by default; not thread-safe.
#include <vector>

#include "common.h"
#include "prof.h"

namespace rgbm {

struct ProfRec { hipEvent_t e0, e1; int variant; double flops, bytes; };
static bool g_on = false;
static int g_only = -1;      // >= 0: only launches of this row are bracketed (prof_select)
static bool g_open = false;    // the last prof_begin_launch recorded its first event
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_enabled() { return g_on; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void prof_begin_launch(hipStream_t s, int variant, double flops, double bytes) {
  g_open = false;
  if (!g_on || (g_only >= 0 && variant != g_only)) return;
  ProfRec r;
  r.e0 = get_event(); r.e1 = get_event(); r.variant = variant; r.flops = flops; r.bytes = bytes;
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  g_open = true;
}

void prof_end_launch(hipStream_t s) {
  if (!g_on || !g_open || g_recs.empty()) return;
  (void)hipEventRecord(g_recs.back().e1, s);
  g_open = false;
}

int prof_select(int variant) {
  g_only = variant;
  return 0;
}

int prof_start() {
  for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_recs.clear();
  g_on = true;
  return 0;
}

// stats[v*4 + {0,1,2,3}] = {launches, total ms, total algorithmic flops, total algorithmic bytes}
int prof_stop(double* stats, int n_variants) {
  g_on = false;
  for (int i = 0; i < n_variants * 4; ++i) stats[i] = 0.0;
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) { set_error("prof: event sync failed"); return -2; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { set_error("prof: elapsed failed"); return -2; }
    if (r.variant >= 0 && r.variant < n_variants) {
      stats[r.variant * 4 + 0] += 1.0;
      stats[r.variant * 4 + 1] += (double)ms;
      stats[r.variant * 4 + 2] += r.flops;
      stats[r.variant * 4 + 3] += r.bytes;
    }
    g_pool.push_back(r.e0); g_pool.push_back(r.e1);
  }
  g_recs.clear();
  return 0;
}

}  // namespace rgbm
