// common.hpp -- shared host-side plumbing for lib/pygmm.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace sr {

// ---- errors: nothing may unwind through the C ABI (the reference's `throw "literal"`,
// gmm.cc:45,59,585, would); entry points catch and park the text here. ----
std::string &last_error();
void set_error(const char *fmt, ...);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

[[noreturn]] void fail(const char *fmt, ...);

#define SR_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            ::sr::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                      \
    } while (0)

// ---- devices.  One process may drive several GPUs: every host thread has a CURRENT device
// (thread-local; the process default until sr_set_device / set_thread_device changes it), and
// everything the library caches -- stream, workspaces, timers, lock -- exists once per device.
// The reference parallelises scoring with a thread pool inside the call (gmm.cc:533-560) and
// utterances with processes (test-gmm.py:128-133); here a host thread per GPU plays both roles
// (SURVEY.md 8e), and a lock per device serialises the threads that share one.
constexpr int MAX_DEVICES = 16;
int current_device();
void set_thread_device(int device);      // this thread's current device
void set_default_device(int device);     // what threads that never chose get (and this thread)
int visible_devices();                   // hipGetDeviceCount, 0 on error

// One lock per device around every compute entry point of the C ABI: the library keeps one stream
// and cached workspaces per device, so concurrent callers on the same device (ctypes drops the
// GIL) are serialised rather than left to race; callers on different devices run in parallel.
std::recursive_mutex &api_mutex();       // of the calling thread's current device

// ---- fork().  The HIP runtime does not survive it: a process forked from one that had already used the GPU inherits
// queues, events and device pointers the kernel driver no longer honours for it (its threads that waited on them are
// gone), and a HIP call there may hang or crash.  The reference's own drivers do exactly that (src/test/test-nperson.py:
// 126-139, test-gmm.py:120-133: `gmmset.fit` in the parent, THEN multiprocessing.Pool, `predict_one` in the forked
// workers) -- harmless for its CPU library.  Here:
//   * nothing touches HIP before the first call that needs the device, so a pool created BEFORE the first compute call
//     gets one runtime per worker (lazy initialisation after the fork);
//   * a process forked AFTER that is marked (pthread_atfork child handler + a pid check): every entry point that needs
//     the device fails with an sr::Error naming the remedy instead of calling into the dead runtime, nothing of the
//     inherited device state is freed there -- and the reference's ten symbols (+ sr_score_frames_f32 / sr_train_f32)
//     keep WORKING: fork_proxy.cpp forwards them to a helper process (lib/sr_fork_helper, spawned on first use) that
//     loads this library afresh and owns a runtime of its own.
void note_gpu_runtime_use();     // call before the process's first HIP call: remembers the pid, installs the fork handler
bool gpu_runtime_lost();         // true in a process forked from one that had used the runtime
[[noreturn]] void fail_gpu_runtime_lost(const char *what);
std::recursive_mutex &api_mutex_of(int device);

// ---- per-device context: one stream, lazily created (nothing touches HIP before the first call that
// needs the device: see above). ----
struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // where launches go (normally `main`; StreamScope redirects)
    hipStream_t main = nullptr, aux = nullptr;
    hipStream_t copy = nullptr;        // host -> device transfers that run under compute (multi.cpp): one per device, so that
                                       // the transfers of the slots sharing a GPU are served in the order they were queued
    bool profiling = false;
    int n_cu = 256;
};
// NUMA: a host thread that feeds a GPU (uploads from its memory, waits on its events) belongs on the cores next to that GPU's
// PCIe root.  device_numa_node: the node of `device` from sysfs (/sys/bus/pci/devices/<bus id>/numa_node), -1 when the platform
// does not say; bind_thread_near_device: pins the CALLING thread to that node's cores (sched_setaffinity) and returns the node
// (-1: left alone).  Used by the slot threads of sr_multi_* and, through sr_bind_thread_near_device, by bench.py's ranks.
int device_numa_node(int device);
int bind_thread_near_device(int device);    // (intersected with the thread's current mask: never widens it)
std::atomic<int> &numa_bind_option();        // sr_set_option("multi_numa_bind", 0 | 1): sr_multi slot threads bind themselves (default 1)
Ctx &ctx();             // of the calling thread's current device
void ensure_device();   // hipSetDevice(current) for THIS thread + lazy stream; throws sr::Error when no usable GPU

// Redirects the calling thread's device launches to another stream of the same device for a scope
// (the feature stage of the pipelined predict path runs on `aux` under the scoring of `main`).
struct StreamScope {
    hipStream_t saved;
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
};

// Lazily constructed per-device singleton (workspaces); leaked on purpose: no hipFree at exit.
void *per_device_slot(void **slots, void *(*make)());
template <typename T>
T &per_device() {
    static void *slots[MAX_DEVICES] = {};
    return *static_cast<T *>(per_device_slot(slots, []() -> void * { return new T(); }));
}

// ---- kernel timing with HIP events on OUR stream ----
enum TimerKind { T_SCORE = 0, T_MFCC = 1, T_CMVN = 2, T_FINALIZE = 3, T_ESTEP = 4, T_SCORE_REF = 5, T_COUNT = 6 };
struct ScopedKernelTimer {
    explicit ScopedKernelTimer(TimerKind k);
    ~ScopedKernelTimer();
    TimerKind kind;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
void profile_collect();          // resolves pending event pairs (synchronises the stream)
void profile_prewarm();          // grows the runtime's event/signal pools once, outside any timed region
void profile_reset();
void profile_get(int kind, double *ms, long *launches);
// probe.hip: executed fp16 MFMA TFLOP/s and shader clock of a kernel that does nothing but v_mfma_f32_32x32x16_f16 on every SIMD
void mfma_peak_probe(double ms_target, double *tflops, double *mhz, int mode = 0);

// Bumped by every device (re)allocation and every upload through DevBuf: a captured hipGraph holds
// raw pointers into, and depends on the contents of, the library's cached workspaces, so it is
// re-captured when this moved (stream.cpp).
extern std::atomic<long> g_devbuf_epoch;

// ---- device buffer ----
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        g_devbuf_epoch++;
        if (count) SR_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void release() {
        if (p) {
            if (!gpu_runtime_lost()) (void)hipFree(p);      // (a forked child leaves its parent's allocations alone)
            g_devbuf_epoch++;
        }
        p = nullptr;
        n = 0;
    }
    void upload(const T *src, size_t count) {
        ensure(count);
        g_devbuf_epoch++;
        if (count) SR_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
    }
    void download(T *dst, size_t count) const {
        if (count) SR_HIP(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
    }
};

inline void sync_stream() { SR_HIP(hipStreamSynchronize(ctx().stream)); }

// Pinned host staging buffer.  A D2H copy straight into the caller's pageable memory makes the
// runtime pin that range on the fly: measured ~8 ms whenever the caller's buffer lands on a new
// address (a fresh numpy array); results therefore come back through this buffer + a memcpy.
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    PinnedBuf(PinnedBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    PinnedBuf &operator=(PinnedBuf &&o) noexcept {
        if (this != &o) {
            if (p && !gpu_runtime_lost()) (void)hipHostFree(p);
            p = o.p; n = o.n; o.p = nullptr; o.n = 0;
        }
        return *this;
    }
    ~PinnedBuf() { if (p && !gpu_runtime_lost()) (void)hipHostFree(p); }
    // (flags: hipHostMallocCoherent | hipHostMallocMapped for memory a kernel writes and the host polls)
    void ensure(size_t count, unsigned flags = hipHostMallocDefault) {
        if (count <= n) return;
        if (p && !gpu_runtime_lost()) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
        SR_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), flags));
        n = count;
    }
};

// An event owned by a movable object (created on first use, timing disabled).
struct EventHolder {
    hipEvent_t e = nullptr;
    EventHolder() = default;
    EventHolder(const EventHolder &) = delete;
    EventHolder &operator=(const EventHolder &) = delete;
    EventHolder(EventHolder &&o) noexcept : e(o.e) { o.e = nullptr; }
    EventHolder &operator=(EventHolder &&o) noexcept {
        if (this != &o) { drop(); e = o.e; o.e = nullptr; }
        return *this;
    }
    ~EventHolder() { drop(); }
    void drop() {
        if (e && !gpu_runtime_lost()) (void)hipEventDestroy(e);
        e = nullptr;
    }
};

// A small host -> device table of an object that is REUSED from call to call (a serving loop's batch whose utterance changes
// length, its tile tables, the feature stage's frame offsets): through a page-locked copy owned by the object, the transfer left in
// flight on the library's stream -- no host wait.  From pageable memory every such table is a blocking staging copy of the
// runtime's plus, because the source may be rewritten, a stream synchronisation: three of them per decision once the layout
// changes (0.182 against 0.127 ms for a fixed-length loop, scripts/debug/serving_varlen.py).  `done` guards the page-locked copy
// against the next refill (normally long complete: the host has seen the pass's results in between).
// Fresh objects keep the plain upload + wait: a page-locked allocation per new batch would cost more than it saves.
constexpr size_t STAGED_TABLE_MAX_BYTES = (size_t)256 << 10;
template <typename T>
struct StagedUpload {
    PinnedBuf<T> h;
    EventHolder done;
    void send(DevBuf<T> &dst, const T *src, size_t n) {
        if (n > dst.n) dst.ensure(n + n / 4);              // (a quarter of headroom: a loop whose sizes creep up re-allocates a few times, not at every new maximum)
        g_devbuf_epoch++;                                  // (contents changed: as DevBuf::upload)
        if (n == 0) return;
        if (!done.e) SR_HIP(hipEventCreateWithFlags(&done.e, hipEventDisableTiming));
        else if (hipEventQuery(done.e) != hipSuccess) SR_HIP(hipEventSynchronize(done.e));
        if (n > h.n) h.ensure(n + n / 4);
        std::memcpy(h.p, src, n * sizeof(T));
        SR_HIP(hipMemcpyAsync(dst.p, h.p, n * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
        SR_HIP(hipEventRecord(done.e, ctx().stream));
    }
};

}  // namespace sr
