// libn3d.so runtime plumbing: ABI version, last-error string, per-family event profiling.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

static thread_local char g_err[1024] = "";

int n3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

struct ProfRec {
    hipEvent_t start, stop;
    int family;
    double flops, bytes;
    bool closed;
};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;

N3dProfScope::N3dProfScope(int family, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.family = family; r.flops = flops; r.bytes = bytes; r.closed = false;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    (void)hipEventRecord(r.start, s);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}

N3dProfScope::~N3dProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].stop, stream);
    g_prof[slot].closed = true;
}

extern "C" {

int n3d_abi_version(void) { return N3D_ABI_VERSION; }
const char* n3d_last_error(void) { return g_err; }

int n3d_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

int n3d_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_prof.clear();
    return 0;
}

int n3d_prof_read(int family, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t cnt = 0;
    for (auto& r : g_prof) {
        if (r.family != family || !r.closed) continue;
        if (hipEventSynchronize(r.stop) != hipSuccess) return n3d_set_error("hipEventSynchronize failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) return n3d_set_error("hipEventElapsedTime failed");
        ms += t; fl += r.flops; by += r.bytes; ++cnt;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}

}  // extern "C"
