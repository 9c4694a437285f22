// Launch-duration probe: HIP events recorded on the launch stream around selected kernels, so
// bench.py can report the *live* average duration of a kernel inside the timed training loop
// (torch.cuda.Event would do the same from Python but cannot bracket a single launch inside a fused
// C-ABI call).  Disabled by default: zero overhead unless obman_prof_enable(1) was called.
#include <vector>

#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace {
struct Rec { int id; hipEvent_t a, b; };
std::vector<Rec> g_pool;
size_t g_used = 0;
bool g_on = false;
constexpr size_t POOL = 8192;
}  // namespace

ObmanProfScope::ObmanProfScope(int id, hipStream_t st) : rec_(-1), st_(st) {
  if (!g_on || g_used >= g_pool.size()) return;
  rec_ = (int)g_used++;
  g_pool[rec_].id = id;
  (void)hipEventRecord(g_pool[rec_].a, st_);
}
ObmanProfScope::~ObmanProfScope() {
  if (rec_ >= 0) (void)hipEventRecord(g_pool[rec_].b, st_);
}

extern "C" {

int obman_prof_enable(int on) {
  if (on && g_pool.empty()) {
    g_pool.resize(POOL);
    for (auto& r : g_pool) {
      if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    }
  }
  g_on = on != 0;
  g_used = 0;
  return 0;
}

int obman_prof_summary(int kernel_id, double* total_ms, long* launches) {
  double tot = 0.0;
  long n = 0;
  for (size_t i = 0; i < g_used; ++i) {
    if (g_pool[i].id != kernel_id) continue;
    if (hipEventSynchronize(g_pool[i].b) != hipSuccess) return -1;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_pool[i].a, g_pool[i].b) != hipSuccess) return -2;
    tot += ms;
    ++n;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return 0;
}

}  // extern "C"
