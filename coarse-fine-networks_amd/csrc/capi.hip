// C-ABI plumbing shared by every entry point: error reporting, library identification and the
// opt-in HIP-event timer that bench.py uses for its live per-kernel roofline measurement.
#include "cfn_common.h"
#include <mutex>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <hipcub/hipcub.hpp>

static thread_local char g_err[512] = "";

extern "C" const char* cfn_last_error(void) { return g_err; }

int cfn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int cfn_det_flush(const char* what);
static bool g_det_on = false;

int cfn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    if (g_det_on) return cfn_det_flush(what);
    return CFN_OK;
}

// ---------------------------------------------------------------------------------------------
// Deterministic mode (see cfn_common.h cfn_add64): record buffer, canonical-order commit.
// ---------------------------------------------------------------------------------------------
static std::vector<void (*)(const CfnDetState*)>& det_setters() {
    static std::vector<void (*)(const CfnDetState*)> v;      // (function-local: translation units register during static initialisation)
    return v;
}
int cfn_det_register(void (*set)(const CfnDetState*)) { det_setters().push_back(set); return (int)det_setters().size(); }

struct DetBuf {
    int dev = -1;
    unsigned long long *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr, *count = nullptr;
    void* temp = nullptr; size_t temp_bytes = 0; unsigned long long cap = 0;
};
static DetBuf g_det;

// one thread per record: the first record of an address adds that address's addends to it in sorted order
__global__ void cfn_det_apply_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    if (i > 0 && keys[i - 1] == k) return;
    double* p = (double*)(uintptr_t)k;
    double acc = *p;
    for (unsigned long long j = i; j < n && keys[j] == k; ++j) acc += __builtin_bit_cast(double, vals[j]);
    *p = acc;
}

static int cfn_det_flush(const char* what) {
    // any stream may have launched the recording kernels: drain the device (this mode is for parity runs, not for speed)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return cfn_fail(CFN_ERR_LAUNCH, "%s: deterministic mode cannot run inside a stream capture", what); }
    (void)cs;
    unsigned long long n = 0;
    if (hipMemcpy(&n, g_det.count, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "%s: deterministic mode: reading the record count failed", what);
    if (n == 0) return CFN_OK;
    (void)hipMemset(g_det.count, 0, sizeof(n));
    if (n > g_det.cap)
        return cfn_fail(CFN_ERR_LAUNCH, "%s: deterministic mode: %llu accumulations in one entry point exceed the record buffer (%llu; set CFN_DET_RECORDS)", what, n, g_det.cap);
    // canonical order: by addend bits, then (stable) by address
    size_t tb = g_det.temp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(g_det.temp, tb, g_det.vals, g_det.vals2, g_det.keys, g_det.keys2, (int)n, 0, 64, (hipStream_t)0) != hipSuccess ||
        hipcub::DeviceRadixSort::SortPairs(g_det.temp, tb, g_det.keys2, g_det.keys, g_det.vals2, g_det.vals, (int)n, 0, 64, (hipStream_t)0) != hipSuccess)
        return cfn_fail(CFN_ERR_LAUNCH, "%s: deterministic mode: sorting the records failed", what);
    hipLaunchKernelGGL(cfn_det_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)0, g_det.keys, g_det.vals, n);
    if (hipDeviceSynchronize() != hipSuccess) return cfn_fail(CFN_ERR_LAUNCH, "%s: deterministic mode: commit failed: %s", what, hipGetErrorString(hipGetLastError()));
    return CFN_OK;
}

// on = 1 / 0: switch; -1: query.  Returns the previous setting (or a negative error).  Per process and device: the mode belongs to the device
// that is current when it is switched on (one process per GPU, as the engine runs).  Not for use during hipGraph capture.
extern "C" int cfn_deterministic(int on) {
    const int prev = g_det_on ? 1 : 0;
    if (on < 0) return prev;
    if ((on != 0) == g_det_on) return prev;
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); cfn_fail(CFN_ERR_LAUNCH, "cfn_deterministic: device synchronisation failed (stream capture?)"); return -1; }
    CfnDetState st = {nullptr, nullptr, nullptr, 0};
    if (on) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (g_det.keys == nullptr || g_det.dev != dev) {
            const char* e = getenv("CFN_DET_RECORDS");
            const unsigned long long cap = e ? strtoull(e, nullptr, 10) : (32ull << 20);
            DetBuf b; b.dev = dev; b.cap = cap;
            size_t tb = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                     (unsigned long long*)nullptr, (int)cap, 0, 64, (hipStream_t)0);
            b.temp_bytes = tb;
            if (hipMalloc(&b.keys, cap * 8) != hipSuccess || hipMalloc(&b.vals, cap * 8) != hipSuccess || hipMalloc(&b.keys2, cap * 8) != hipSuccess ||
                hipMalloc(&b.vals2, cap * 8) != hipSuccess || hipMalloc(&b.count, 8) != hipSuccess || hipMalloc(&b.temp, tb ? tb : 8) != hipSuccess) {
                (void)hipGetLastError();
                cfn_fail(CFN_ERR_LAUNCH, "cfn_deterministic: cannot allocate the record buffers (%llu records)", cap);
                return -1;
            }
            g_det = b;                                  // (buffers of another device, if any, stay allocated)
        }
        (void)hipMemset(g_det.count, 0, 8);
        st = CfnDetState{g_det.keys, g_det.vals, g_det.count, g_det.cap};
    }
    for (auto set : det_setters()) set(&st);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); cfn_fail(CFN_ERR_LAUNCH, "cfn_deterministic: switching failed"); return -1; }
    g_det_on = on != 0;
    return prev;
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
