// Library identification + optional in-library kernel timing (HIP events recorded on the launch stream immediately
// around the two dominant kernels), used by bench.py for the live roofline so that its per-launch durations are the
// same quantity rocprofv3 --kernel-trace reports.
#include <vector>
#include "common.h"
#include "bts_internal.h"

extern "C" const char* bts_version(void) { return "bts_hip 0.1 gfx950"; }

struct BtsProfRec {
  int sym;
  double flops;
  hipEvent_t e0, e1;
};
static std::vector<BtsProfRec> g_recs;
static int g_prof_on = 0;

int bts_prof_on() { return g_prof_on; }
void bts_prof_begin(int sym, double flops, hipStream_t stream) {
  BtsProfRec r;
  r.sym = sym;
  r.flops = flops;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, stream);
  g_recs.push_back(r);
}
void bts_prof_end(hipStream_t stream) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, stream);
}

// on != 0: start recording (drops earlier records); on == 0: stop recording (records stay readable)
extern "C" int bts_profile_enable(int on) {
  if (on) {
    for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_recs.clear();
  }
  g_prof_on = on ? 1 : 0;
  return BTS_OK;
}
extern "C" int bts_profile_count(void) { return (int)g_recs.size(); }
// record i -> symbol id (0..4 igemm_kernel config, +8 for the 1x1x1 staging variant; 100 = wgrad_kernel),
// algorithmic FLOPs of the launch, elapsed ms (the stream must have been synchronised)
extern "C" int bts_profile_get(int i, int* sym, double* flops, float* ms) {
  if (i < 0 || i >= (int)g_recs.size()) return BTS_ERR_SHAPE;
  *sym = g_recs[i].sym;
  *flops = g_recs[i].flops;
  hipError_t e = hipEventElapsedTime(ms, g_recs[i].e0, g_recs[i].e1);
  return e == hipSuccess ? BTS_OK : (int)e;
}
