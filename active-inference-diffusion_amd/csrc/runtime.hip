// Error string, ABI version and the event-based profiler behind aid_prof_*.
#include "aid_common.h"
#include <stdarg.h>
#include <vector>
#include <mutex>

namespace aid {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

struct ProfRec { hipEvent_t e0, e1; double flops, bytes; int id, m, n, k, layout, variant; };
struct ProfState {
    bool on = false;
    std::mutex mu;
    std::vector<ProfRec> recs[PROF_COUNT];
};
static ProfState g_prof;

ProfScope::ProfScope(int id_, hipStream_t s_, double fl, double by) : id(id_), s(s_), on(g_prof.on), flops(fl), bytes(by) {
    if (!on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e0, s);
}
void ProfScope::cancel() {
    if (!on) return;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); on = false;
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.recs[id].push_back({e0, e1, flops, bytes, id, m, n, k, layout, variant});
}

}  // namespace aid

using namespace aid;

extern "C" {
int aid_version(void) { return AID_ABI_VERSION; }
const char* aid_last_error(void) { return g_err; }
int aid_prof_enable(int on) { g_prof.on = on != 0; return AID_OK; }
int aid_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int k = 0; k < PROF_COUNT; ++k) {
        for (auto& r : g_prof.recs[k]) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        g_prof.recs[k].clear();
    }
    return AID_OK;
}
int aid_prof_dump(const char* path) {
    AID_REQUIRE(path, AID_E_NULL, "aid_prof_dump: null path");
    FILE* f = fopen(path, "w");
    AID_REQUIRE(f, AID_E_SHAPE, "aid_prof_dump: cannot open %s", path);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    fprintf(f, "kernel_id,M,N,K,layout,ms,variant\n");
    for (int k = 0; k < PROF_COUNT; ++k)
        for (auto& r : g_prof.recs[k]) {
            float t = 0.f;
            if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = -1.f;
            fprintf(f, "%d,%d,%d,%d,%d,%.6f,%d\n", r.id, r.m, r.n, r.k, r.layout, t, r.variant);
        }
    fclose(f);
    return AID_OK;
}
int aid_prof_read(int kernel_id, double out[4]) {
    AID_REQUIRE(kernel_id >= 0 && kernel_id < PROF_COUNT && out, AID_E_SHAPE, "aid_prof_read: bad kernel id");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double ms = 0, fl = 0, by = 0;
    for (auto& r : g_prof.recs[kernel_id]) {
        AID_CHECK_HIP(hipEventSynchronize(r.e1));
        float t = 0.f; AID_CHECK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        ms += t; fl += r.flops; by += r.bytes;
    }
    out[0] = (double)g_prof.recs[kernel_id].size(); out[1] = ms; out[2] = fl; out[3] = by;
    return AID_OK;
}
}
