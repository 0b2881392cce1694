// common.cpp -- error parking, lazy device context, HIP-event kernel timers.
#include "common.hpp"

#include <mutex>

namespace sr {

std::string &last_error() {
    static thread_local std::string e;
    return e;
}

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
}

void fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(buf);
}

std::atomic<long> g_devbuf_epoch{0};

std::recursive_mutex &api_mutex() {
    static std::recursive_mutex *m = new std::recursive_mutex();   // leaked: usable during exit
    return *m;
}

static Ctx g_ctx;
static bool g_ready = false;
static std::mutex g_mu;

Ctx &ctx() { return g_ctx; }

void ensure_device() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready) return;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail("no HIP device available (%s); lib/pygmm.so has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (g_ctx.device >= n) fail("device %d requested but only %d visible", g_ctx.device, n);
    SR_HIP(hipSetDevice(g_ctx.device));
    hipDeviceProp_t prop;
    SR_HIP(hipGetDeviceProperties(&prop, g_ctx.device));
    g_ctx.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SR_HIP(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    g_ready = true;
}

// ---------------- timers ----------------
struct Pending {
    TimerKind kind;
    hipEvent_t e0, e1;
};
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static double g_ms[T_COUNT];
static long g_launches[T_COUNT];

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    SR_HIP(hipEventCreate(&e));
    return e;
}

ScopedKernelTimer::ScopedKernelTimer(TimerKind k) : kind(k) {
    if (!ctx().profiling) return;
    e0 = get_event();
    e1 = get_event();
    SR_HIP(hipEventRecord(e0, ctx().stream));
}

ScopedKernelTimer::~ScopedKernelTimer() {
    if (!e0) return;
    (void)hipEventRecord(e1, ctx().stream);
    g_pending.push_back({kind, e0, e1});
}

// The HIP runtime grows its signal pool the first time more than a handful of events have been
// recorded (measured: one ~8 ms stall in the second profiled step).  Do that growth here, once, so
// that timed regions bracketed by events see none of it.
void profile_prewarm() {
    static bool done = false;
    if (done) return;
    done = true;
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) {
        SR_HIP(hipEventCreate(&e));
        SR_HIP(hipEventRecord(e, ctx().stream));
    }
    SR_HIP(hipStreamSynchronize(ctx().stream));
    float ms = 0.f;
    for (size_t i = 1; i < ev.size(); i++) (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
    for (auto &e : ev) g_pool.push_back(e);
}

void profile_collect() {
    if (g_pending.empty()) return;
    SR_HIP(hipStreamSynchronize(ctx().stream));
    for (auto &p : g_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_ms[p.kind] += ms;
            g_launches[p.kind] += 1;
        }
        g_pool.push_back(p.e0);
        g_pool.push_back(p.e1);
    }
    g_pending.clear();
}

void profile_reset() {
    profile_collect();
    for (int i = 0; i < T_COUNT; i++) {
        g_ms[i] = 0;
        g_launches[i] = 0;
    }
}

void profile_get(int kind, double *ms, long *launches) {
    profile_collect();
    if (kind < 0 || kind >= T_COUNT) fail("bad timer kind %d", kind);
    if (ms) *ms = g_ms[kind];
    if (launches) *launches = g_launches[kind];
}

}  // namespace sr
