// api.hip — library-level entry points: version, error string, per-kernel HIP-event profiling.
#include "common.h"
#include <stdarg.h>
#include <vector>
#include <mutex>

static thread_local char g_err[512] = "";

void scl_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int scl_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        scl_set_error("%s: %s", what, hipGetErrorString(e));
        return SCL_ELAUNCH;
    }
    return SCL_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
namespace {
struct ProfPair { hipEvent_t a, b; };
struct ProfMeta { int32_t v[8]; };      // what the launch was: M, N, K, flags, z (batch x split-K), kernel variant, 0, 0 (scl_prof_note)
struct ProfState {
    bool on = false;
    std::vector<ProfPair> used;
    std::vector<ProfMeta> meta;      // parallel to `used`
    std::vector<ProfPair> pool;
    double flops = 0.0;
};
ProfState g_prof[SCL_KID_MAX];
std::mutex g_prof_mu;
}  // namespace

thread_local SclProfScope* scl_prof_active = nullptr;

SclProfScope::SclProfScope(int kid_, hipStream_t s_, double flops, bool dispatch_)
    : kid(kid_), s(s_), slot(nullptr), ea(nullptr), eb(nullptr), dispatch(dispatch_), taken(false) {
    if (kid < 0 || kid >= SCL_KID_MAX || !g_prof[kid].on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& st = g_prof[kid];
    ProfPair p;
    if (!st.pool.empty()) { p = st.pool.back(); st.pool.pop_back(); }
    else { if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; }
    st.used.push_back(p);
    st.meta.push_back(ProfMeta{{0, 0, 0, 0, 0, 0, 0, 0}});
    st.flops += flops;
    slot = (void*)(uintptr_t)st.used.size();  // index + 1
    ea = p.a; eb = p.b;
    if (dispatch) scl_prof_active = this;      // the launch inside the scope takes the two events itself
    else hipEventRecord(p.a, s);
}
void SclProfScope::note(int M, int N, int K, int flags, int z, int variant) {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& st = g_prof[kid];
    const size_t i = (size_t)(uintptr_t)slot - 1;
    if (i < st.meta.size()) st.meta[i] = ProfMeta{{M, N, K, flags, z, variant, 0, 0}};
}
SclProfScope::~SclProfScope() {
    if (dispatch && scl_prof_active == this) scl_prof_active = nullptr;
    if (!slot) return;
    if (dispatch) {
        if (!taken) { hipEventRecord(ea, s); hipEventRecord(eb, s); }      // no launch went through SCL_LAUNCH: an empty interval
        return;
    }
    hipEventRecord(eb, s);
}

extern "C" int scl_version(void) { return 100; }
extern "C" const char* scl_last_error(void) { return g_err; }

extern "C" int scl_prof_enable(int kid, int on) {
    SCL_REQUIRE(kid >= 0 && kid < SCL_KID_MAX, "prof: bad kernel id %d", kid);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[kid].on = on != 0;
    return SCL_OK;
}

extern "C" int scl_prof_reserve(int kid, int n_pairs) {
    SCL_REQUIRE(kid >= 0 && kid < SCL_KID_MAX && n_pairs >= 0 && n_pairs <= 65536, "prof: bad reserve (%d, %d)", kid, n_pairs);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& st = g_prof[kid];
    while ((int)st.pool.size() < n_pairs) {
        ProfPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return SCL_ELAUNCH;
        st.pool.push_back(p);
    }
    return SCL_OK;
}

// per-launch view of what scl_prof_read sums: call it BEFORE scl_prof_read (which recycles the events)
extern "C" int scl_prof_read_launches(int kid, int cap, float* ms, int32_t* meta8, int64_t* n_launches) {
    SCL_REQUIRE(kid >= 0 && kid < SCL_KID_MAX && cap >= 0 && (cap == 0 || (ms && meta8)), "prof: bad launch read (%d, %d)", kid, cap);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& st = g_prof[kid];
    const size_t n = st.used.size();
    if (n_launches) *n_launches = (int64_t)n;
    for (size_t i = 0; i < n && i < (size_t)cap; ++i) {
        if (hipEventSynchronize(st.used[i].b) != hipSuccess) { scl_set_error("prof: event sync failed"); return SCL_ELAUNCH; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, st.used[i].a, st.used[i].b) != hipSuccess) t = 0.f;
        ms[i] = t;
        for (int j = 0; j < 8; ++j) meta8[8 * i + j] = st.meta[i].v[j];
    }
    return SCL_OK;
}

extern "C" int scl_prof_read(int kid, int64_t* n_launches, double* total_ms, double* total_flops) {
    SCL_REQUIRE(kid >= 0 && kid < SCL_KID_MAX, "prof: bad kernel id %d", kid);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfState& st = g_prof[kid];
    double ms = 0.0;
    for (auto& p : st.used) {
        if (hipEventSynchronize(p.b) != hipSuccess) { scl_set_error("prof: event sync failed"); return SCL_ELAUNCH; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) ms += t;
        st.pool.push_back(p);
    }
    if (n_launches) *n_launches = (int64_t)st.used.size();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = st.flops;
    st.used.clear();
    st.meta.clear();
    st.flops = 0.0;
    return SCL_OK;
}


// ---- stream ordering ---------------------------------------------------------------------------------------------------------------
// "work submitted to `waiter` from now on starts after everything submitted to `signaler` so far": one event record + one stream
// wait.  Events come from a ring (a wait captures the record that precedes it, so a slot may be re-recorded once its wait has been
// issued).  Used by the encoder backward to run the weight-gradient GEMMs on a second stream (scl_amd/encoder.py).
#include <mutex>
extern "C" int scl_stream_wait_stream(void* waiter, void* signaler) {
    static hipEvent_t ring[256];
    static int created = 0, next = 0;
    static std::mutex mu;
    SCL_REQUIRE(waiter != signaler, "stream_wait_stream: a stream cannot wait for itself");
    std::lock_guard<std::mutex> lock(mu);
    if (!created) {
        for (int i = 0; i < 256; ++i)
            if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess) return SCL_ELAUNCH;
        created = 1;
    }
    hipEvent_t e = ring[next];
    next = (next + 1) & 255;
    if (hipEventRecord(e, (hipStream_t)signaler) != hipSuccess) return SCL_ELAUNCH;
    if (hipStreamWaitEvent((hipStream_t)waiter, e, 0) != hipSuccess) return SCL_ELAUNCH;
    return SCL_OK;
}
