// C-ABI plumbing shared by every entry point: error reporting, library identification and the
// opt-in HIP-event timer that bench.py uses for its live per-kernel roofline measurement.
#include "cfn_common.h"
#include <mutex>
#include <vector>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" const char* cfn_last_error(void) { return g_err; }

int cfn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int cfn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return CFN_OK;
}

extern "C" const char* cfn_version(void) { return "cfn_hip 0.1 (gfx950)"; }

// number of compute units / device name: lets the host side size grids and print the box
extern "C" int cfn_device_info(int* cus, int* lds_per_cu, char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "hipGetDevice failed");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "hipGetDeviceProperties failed");
    if (cus) *cus = p.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return CFN_OK;
}

// ---------------------------------------------------------------------------------------------
// per-family event timing.  Disabled by default (zero overhead: one relaxed load per launch).
// ---------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; double bytes; };
static std::mutex g_prof_mu;
static bool g_prof_on[CFN_K_COUNT] = {false};
static std::vector<ProfRec> g_prof[CFN_K_COUNT];
static std::vector<hipEvent_t> g_free_events;

static hipEvent_t prof_event() {
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

CfnProfScope::CfnProfScope(int family, hipStream_t stream, double bytes) : fam(family), s(stream), e0(nullptr), on(false) {
    if (!g_prof_on[family]) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    on = true;
    ProfRec r;
    r.e0 = prof_event();
    r.e1 = prof_event();
    r.bytes = bytes;
    hipEventRecord(r.e0, s);
    g_prof[family].push_back(r);
}

CfnProfScope::~CfnProfScope() {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_prof[fam].back().e1, s);
}

extern "C" int cfn_prof_enable(int family, int on) {
    if (family < 0 || family >= CFN_K_COUNT) return cfn_fail(CFN_ERR_ARG, "cfn_prof_enable: bad family %d", family);
    g_prof_on[family] = on != 0;
    return CFN_OK;
}

// Sums (and clears) the recorded launches of one family.  Synchronises on each end event.
extern "C" int cfn_prof_collect(int family, double* total_ms, long* launches, double* total_bytes) {
    if (family < 0 || family >= CFN_K_COUNT) return cfn_fail(CFN_ERR_ARG, "cfn_prof_collect: bad family %d", family);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, by = 0;
    long n = 0;
    for (auto& r : g_prof[family]) {
        hipEventSynchronize(r.e1);
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms += t; by += r.bytes; ++n; }
        g_free_events.push_back(r.e0);
        g_free_events.push_back(r.e1);
    }
    g_prof[family].clear();
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (total_bytes) *total_bytes = by;
    return CFN_OK;
}
