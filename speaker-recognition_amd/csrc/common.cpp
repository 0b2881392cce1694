// common.cpp -- error parking, lazy device context, HIP-event kernel timers.
#include "common.hpp"

#include <climits>
#include <mutex>

#include <new>

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>

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

static std::atomic<int> g_default_device{0};
static thread_local int tl_device = -1;

int current_device() { return tl_device >= 0 ? tl_device : g_default_device.load(); }

static void check_device_index(int device) {
    if (device < 0 || device >= MAX_DEVICES) fail("device index %d out of range (0..%d)", device, MAX_DEVICES - 1);
}
void set_thread_device(int device) {
    check_device_index(device);
    tl_device = device;
}
void set_default_device(int device) {
    check_device_index(device);
    g_default_device.store(device);
    tl_device = device;
}

static std::recursive_mutex *g_api_mutex = new std::recursive_mutex[MAX_DEVICES];   // leaked: usable during exit
std::recursive_mutex &api_mutex_of(int device) { return g_api_mutex[device]; }
std::recursive_mutex &api_mutex() { return g_api_mutex[current_device()]; }

static Ctx g_ctx[MAX_DEVICES];
static bool g_ready[MAX_DEVICES];
static std::mutex g_mu;
static std::mutex g_slot_mu;

void reference_rand_fork_child();       // kmeans_init.hip

// ---- fork (common.hpp) ----
static std::atomic<long> g_runtime_pid{0};       // the process that made the first HIP call (0: none yet)
static std::atomic<bool> g_runtime_lost{false};
void fork_proxy_atfork_child();       // fork_proxy.cpp
static std::atomic<bool> g_atfork_installed{false};

static void atfork_child() {
    // Only the forking thread exists here.  Locks another thread of the parent held at that instant would stay locked
    // for ever: give the child fresh ones (nothing else of the library's state is shared with a thread that is gone).
    new (&g_mu) std::mutex();
    new (&g_slot_mu) std::mutex();
    for (int i = 0; i < MAX_DEVICES; i++) new (&g_api_mutex[i]) std::recursive_mutex();
    reference_rand_fork_child();
    fork_proxy_atfork_child();
    if (g_runtime_pid.load() != 0) g_runtime_lost.store(true);
}

void note_gpu_runtime_use() {
    if (!g_atfork_installed.exchange(true)) (void)pthread_atfork(nullptr, nullptr, atfork_child);
    long expected = 0;
    (void)g_runtime_pid.compare_exchange_strong(expected, (long)getpid());
}

bool gpu_runtime_lost() {
    if (g_runtime_lost.load(std::memory_order_relaxed)) return true;
    // (a fork() that bypassed the handlers -- a raw clone -- is caught by the pid)
    const long owner = g_runtime_pid.load(std::memory_order_relaxed);
    if (owner != 0 && owner != (long)getpid()) {
        g_runtime_lost.store(true);
        return true;
    }
    return false;
}

void fail_gpu_runtime_lost(const char *what) {
    fail("%s: this process (pid %ld) was forked after its parent (pid %ld) had initialised the GPU runtime, and HIP does "
         "not survive fork().  The ten pygmm symbols, sr_score_frames_f32 and sr_train_f32 still work here (helper "
         "process); for the batched sr_* interface create the pool BEFORE the first compute call or use the 'spawn' "
         "start method",
         what, (long)getpid(), g_runtime_pid.load());
}

int visible_devices() {
    if (gpu_runtime_lost()) return 0;
    note_gpu_runtime_use();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

Ctx &ctx() { return g_ctx[current_device()]; }

void ensure_device() {
    if (gpu_runtime_lost()) fail_gpu_runtime_lost("no usable GPU runtime");
    note_gpu_runtime_use();
    const int d = current_device();
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready[d]) {
        // the HIP current device is per host thread: set it on every entry (a thread that first
        // reaches the library after another one initialised the context would otherwise allocate
        // on device 0 while the stream belongs to device d)
        SR_HIP(hipSetDevice(d));
        return;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail("no HIP device available (%s); lib/pygmm.so has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (d >= n) fail("device %d requested but only %d visible", d, n);
    SR_HIP(hipSetDevice(d));
    hipDeviceProp_t prop;
    SR_HIP(hipGetDeviceProperties(&prop, d));
    g_ctx[d].device = d;
    g_ctx[d].n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SR_HIP(hipStreamCreateWithFlags(&g_ctx[d].main, hipStreamNonBlocking));
    SR_HIP(hipStreamCreateWithFlags(&g_ctx[d].aux, hipStreamNonBlocking));
    SR_HIP(hipStreamCreateWithFlags(&g_ctx[d].copy, hipStreamNonBlocking));
    g_ctx[d].stream = g_ctx[d].main;
    g_ready[d] = true;
}

// ---------------- NUMA ----------------
static bool read_small_file(const std::string &path, char *buf, size_t cap) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}

int device_numa_node(int device) {
    // (slot threads of sr_multi ask concurrently: one atomic per device, "unknown yet" = INT_MIN; a race computes the same value twice)
    static std::atomic<int> cache[MAX_DEVICES];
    static std::once_flag init;
    std::call_once(init, [] { for (auto &c : cache) c.store(INT_MIN); });
    check_device_index(device);
    if (cache[device].load() != INT_MIN) return cache[device].load();
    int node = -1;
    if (!gpu_runtime_lost()) {
        note_gpu_runtime_use();
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess) {
            for (char *c = bus; *c; c++) *c = (char)tolower(*c);          // sysfs spells the address in lower case
            char txt[64];
            if (read_small_file(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", txt, sizeof txt)) node = atoi(txt);
        } else {
            (void)hipGetLastError();
        }
    }
    cache[device].store(node);
    return node;
}

std::atomic<int> &numa_bind_option() {      // sr_set_option("multi_numa_bind", 0 | 1)
    static std::atomic<int> v{1};
    return v;
}

int bind_thread_near_device(int device) {
    const int node = device_numa_node(device);
    if (node < 0) return -1;
    char txt[4096];
    if (!read_small_file("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", txt, sizeof txt)) return -1;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_set = 0;
    for (char *tok = strtok(txt, ",\n"); tok; tok = strtok(nullptr, ",\n")) {      // "0-63,128-191"
        int lo = 0, hi = 0;
        const int got = sscanf(tok, "%d-%d", &lo, &hi);
        if (got == 1) hi = lo;
        if (got >= 1)
            for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) {
                CPU_SET(c, &set);
                n_set++;
            }
    }
    // never WIDEN what the thread may use: a taskset, a launcher's per-rank pinning or a cgroup cpuset stays in force, and a mask
    // that shares no CPU with the node is left alone (sched_setaffinity would fail with EINVAL on an empty intersection)
    cpu_set_t have;
    CPU_ZERO(&have);
    if (n_set == 0 || sched_getaffinity(0, sizeof have, &have) != 0) return -1;
    CPU_AND(&set, &set, &have);
    if (CPU_COUNT(&set) == 0) return -1;
    if (CPU_EQUAL(&set, &have)) return node;               // already there
    if (sched_setaffinity(0, sizeof set, &set) != 0) return -1;
    return node;
}

StreamScope::StreamScope(hipStream_t s) : saved(ctx().stream) { ctx().stream = s; }
StreamScope::~StreamScope() { ctx().stream = saved; }

void *per_device_slot(void **slots, void *(*make)()) {
    const int d = current_device();
    std::lock_guard<std::mutex> lk(g_slot_mu);
    if (!slots[d]) slots[d] = make();
    return slots[d];
}

// ---------------- timers ----------------
struct Pending {
    TimerKind kind;
    hipEvent_t e0, e1;
};
struct TimerState {
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double ms[T_COUNT] = {};
    long launches[T_COUNT] = {};
    bool prewarmed = false;
};
static TimerState &ts() { return per_device<TimerState>(); }

static hipEvent_t get_event() {
    auto &t = ts();
    if (!t.pool.empty()) {
        hipEvent_t e = t.pool.back();
        t.pool.pop_back();
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
    ts().pending.push_back({kind, e0, e1});
}

// The HIP runtime grows its signal pool the first time more than a handful of events have been
// recorded (measured: one ~8 ms stall in the second profiled step).  Do that growth here, once, so
// that timed regions bracketed by events see none of it.
void profile_prewarm() {
    auto &t = ts();
    if (t.prewarmed) return;
    t.prewarmed = true;
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) {
        SR_HIP(hipEventCreate(&e));
        SR_HIP(hipEventRecord(e, ctx().stream));
    }
    SR_HIP(hipStreamSynchronize(ctx().stream));
    float ms = 0.f;
    for (size_t i = 1; i < ev.size(); i++) (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
    for (auto &e : ev) t.pool.push_back(e);
}

void profile_collect() {
    auto &t = ts();
    if (t.pending.empty()) return;
    if (gpu_runtime_lost()) {          // a forked child: the parent's events mean nothing here
        t.pending.clear();
        return;
    }
    SR_HIP(hipStreamSynchronize(ctx().main));
    SR_HIP(hipStreamSynchronize(ctx().aux));
    for (auto &p : t.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            t.ms[p.kind] += ms;
            t.launches[p.kind] += 1;
        }
        t.pool.push_back(p.e0);
        t.pool.push_back(p.e1);
    }
    t.pending.clear();
}

void profile_reset() {
    profile_collect();
    auto &t = ts();
    for (int i = 0; i < T_COUNT; i++) {
        t.ms[i] = 0;
        t.launches[i] = 0;
    }
}

void profile_get(int kind, double *ms, long *launches) {
    profile_collect();
    if (kind < 0 || kind >= T_COUNT) fail("bad timer kind %d", kind);
    auto &t = ts();
    if (ms) *ms = t.ms[kind];
    if (launches) *launches = t.launches[kind];
}

}  // namespace sr
