// fork_proxy.cpp -- the reference's calling pattern of this ABI that a GPU library cannot serve in-process:
//
//     gmmset.fit(X_train, y_train)                      # parent: train_model -- the GPU runtime is up
//     pool = multiprocessing.Pool(concurrency)          # fork()
//     pool.map(predict_one, ...)                        # children: score_all on the pickled set
//                                                   (src/test/test-nperson.py:126-139, src/test/test-gmm.py:120-133)
//
// The reference's lib/pygmm.so is plain CPU code and does not care.  A HIP runtime does not survive fork() (common.hpp):
// in such a child every HIP call is a call into queues and events of another process.  So a child that was forked after its
// parent had used the GPU never calls HIP; instead the entry points that take plain host data -- the reference's ten
// symbols (pygmm.hh:28-41: train_model, train_model_from_ubm, score_all, score_batch, score_instance compute; the other five are
// host-only and work as they are) and their contiguous forms sr_score_frames_f32 / sr_train_f32 -- are forwarded over a
// socketpair to a HELPER PROCESS: lib/sr_fork_helper (csrc/fork_helper.c: posix_spawn, i.e. a fresh address space), which
// dlopens this same library, brings up a runtime of its own on the same device and serves one forked child until that child's
// end of the socket closes.  Every child spawns its own helper on its first forwarded call (the pattern above: one per pool
// worker, exactly the processes-per-GPU the "pool before the first compute call" variant ends up with).
//
// What crosses: requests carry the model(s) as float64 parameter arrays (once per model and helper: the helper keeps what it has
// seen, keyed by the caller's handle and a hash of the contents), the frames as the contiguous fp32 matrix the entry point has
// built anyway, the caller's Parameter block, the options set through sr_set_option so far (replayed at start-up, forwarded
// afterwards), the current device index and the state of the restated libc rand() (kmeans_init.hip), which comes back with
// the reply: training from scratch in a forked child draws what the reference's forked child would draw.  Replies carry a
// status + message (an sr::Error in the helper becomes an sr::Error here), sums / per-frame values, or the trained parameters.
//
// Handles that own device state (SRModelSet, SRBatch, SRMfcc extraction, streams, multi) are NOT forwarded: their entry
// points fail with a message naming the remedy (common.cpp: fail_gpu_runtime_lost).
#include "../../include/pygmm_hip.h"

#include "common.hpp"
#include "fork_proxy.hpp"
#include "gmm_model.hpp"

#include <atomic>
#include <cerrno>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <signal.h>
#include <spawn.h>
#include <sys/prctl.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

extern char **environ;

namespace sr {

int train_em(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param, long seed);   // em.hip
void reference_rand_state(int32_t *words36, bool set);                                                           // kmeans_init.hip
void score_one_local(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags);       // abi.cpp
void score_models_local(GMM *const *models, int n_models, const float *X, long n, int dim, double *sums_out, int flags);   // abi.cpp

namespace {

constexpr uint32_t MAGIC = 0x53524650u;      // "SRFP"
enum Op : uint32_t { OP_SCORE = 1, OP_TRAIN = 2, OP_OPTION = 3, OP_PING = 4, OP_SCORE_MODELS = 5 };

// ---- framed, blocking I/O on a stream socket ----
void send_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            fail("fork helper: send failed (%s) -- the helper process is gone", strerror(errno));
        }
        c += k;
        n -= (size_t)k;
    }
}
void recv_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t k = ::recv(fd, c, n, 0);
        if (k < 0) {
            if (errno == EINTR) continue;
            fail("fork helper: recv failed (%s)", strerror(errno));
        }
        if (k == 0) fail("fork helper: the other end closed the connection");
        c += k;
        n -= (size_t)k;
    }
}

struct Writer {
    std::vector<char> buf;
    template <typename T>
    void pod(const T &v) {
        const char *c = reinterpret_cast<const char *>(&v);
        buf.insert(buf.end(), c, c + sizeof(T));
    }
    void bytes(const void *p, size_t n) {
        const char *c = static_cast<const char *>(p);
        buf.insert(buf.end(), c, c + n);
    }
    void str(const std::string &s) {
        pod<uint64_t>(s.size());
        bytes(s.data(), s.size());
    }
    void flush(int fd) {             // [u64 length][payload]
        const uint64_t n = buf.size();
        send_all(fd, &n, sizeof n);
        send_all(fd, buf.data(), buf.size());
        buf.clear();
    }
};
struct Reader {
    std::vector<char> buf;
    size_t at = 0;
    void fill(int fd) {
        uint64_t n = 0;
        recv_all(fd, &n, sizeof n);
        if (n > ((uint64_t)1 << 40)) fail("fork helper: absurd message length");
        buf.resize((size_t)n);
        at = 0;
        if (n) recv_all(fd, buf.data(), (size_t)n);
    }
    void need(size_t n) const {
        if (at + n > buf.size()) fail("fork helper: short message");
    }
    template <typename T>
    T pod() {
        need(sizeof(T));
        T v;
        std::memcpy(&v, buf.data() + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    const char *take(size_t n) {
        need(n);
        const char *p = buf.data() + at;
        at += n;
        return p;
    }
    std::string str() {
        const uint64_t n = pod<uint64_t>();
        return std::string(take((size_t)n), (size_t)n);
    }
};

uint64_t fnv1a64(uint64_t h, const void *p, size_t bytes) {       // (FNV-1a over 8-byte words, then the tail's bytes)
    const unsigned char *b = static_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 1099511628211ull;
    }
    for (; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
uint64_t model_hash(const GMM &g) {
    uint64_t h = 1469598103934665603ull;
    const int hd[2] = {g.nr_mixtures, g.dim};
    h = fnv1a64(h, hd, sizeof hd);
    h = fnv1a64(h, g.weights.data(), g.weights.size() * sizeof(double));
    h = fnv1a64(h, g.mean.data(), g.mean.size() * sizeof(double));
    h = fnv1a64(h, g.sigma.data(), g.sigma.size() * sizeof(double));
    return h;
}

// a model on the wire: [u64 handle][u64 hash][i32 nr_mixtures][i32 dim][u8 with_params]( [w][mean][sigma] as float64 )
void put_model(Writer &w, const GMM &g, uint64_t hash, bool with_params) {
    w.pod<uint64_t>((uint64_t)(uintptr_t)&g);
    w.pod<uint64_t>(hash);
    w.pod<int32_t>(g.nr_mixtures);
    w.pod<int32_t>(g.dim);
    const bool trained = g.trained();
    w.pod<uint8_t>(trained ? (with_params ? 1 : 2) : 0);      // 0 untrained, 1 parameters follow, 2 "you have them"
    if (trained && with_params) {
        w.bytes(g.weights.data(), g.weights.size() * sizeof(double));
        w.bytes(g.mean.data(), g.mean.size() * sizeof(double));
        w.bytes(g.sigma.data(), g.sigma.size() * sizeof(double));
    }
}
void put_params(Writer &w, const GMM &g) {
    w.pod<int32_t>(g.nr_mixtures);
    w.pod<int32_t>(g.dim);
    w.bytes(g.weights.data(), g.weights.size() * sizeof(double));
    w.bytes(g.mean.data(), g.mean.size() * sizeof(double));
    w.bytes(g.sigma.data(), g.sigma.size() * sizeof(double));
}
void get_params(Reader &r, GMM &g) {
    g.nr_mixtures = r.pod<int32_t>();
    g.dim = r.pod<int32_t>();
    if (g.nr_mixtures < 0 || g.dim < 0) fail("fork helper: bad model shape");
    const size_t K = (size_t)g.nr_mixtures, KD = K * (size_t)g.dim;
    const double *p = reinterpret_cast<const double *>(r.take(K * sizeof(double)));
    g.weights.assign(p, p + K);
    p = reinterpret_cast<const double *>(r.take(KD * sizeof(double)));
    g.mean.assign(p, p + KD);
    p = reinterpret_cast<const double *>(r.take(KD * sizeof(double)));
    g.sigma.assign(p, p + KD);
    g.drop_single();
}

// what the helper answers when asked to use a model it has evicted (the conversation stays intact; the caller sends it again)
constexpr char HELPER_MISS[] = "fork helper miss";
// models (host parameters + packed device set) the helper keeps, least recently used out: by count and by bytes of parameters.
// (Through round 5: 64 models.  The reference's own logged run -- test-nperson.py: 80 speakers, every pool worker scoring each utterance
// against all of them in order -- cycled through 80 with room for 64: every call missed, sent its model again and had it packed and
// uploaded again: 0.9 ms per score_all call, 36 s for the run's 4000 utterances on 8 workers.)
constexpr size_t HELPER_MAX_MODELS = 4096;
constexpr size_t HELPER_MAX_MODEL_BYTES = (size_t)1 << 30;
std::atomic<long> g_helper_max_models{0};    // 0: HELPER_MAX_MODELS (test hook: the eviction path with a handful of models)

// ---- the options set so far: a fresh helper starts from the library's defaults, the forked child did not ----
std::vector<std::pair<std::string, long>> &option_log() {
    static auto *v = new std::vector<std::pair<std::string, long>>();
    return *v;
}

// ---- client side: one helper per (forked) process ----
struct Helper {
    long owner_pid = 0;
    pid_t pid = -1;
    int fd = -1;
    std::unordered_map<uint64_t, uint64_t> sent;      // handle -> hash of the parameters the helper holds for it
    // handle -> {uid, generation, hash}: the hash of a model's parameters is formed once per state of the parameters, not once per
    // call (byte by byte it was 17 us per 32 x 34 model -- 1.4 ms of every 80-speaker predict_one in a forked worker)
    struct Hashed { uint64_t uid, generation, hash; };
    std::unordered_map<uint64_t, Hashed> hashed;
    // one conversation at a time: the callers' api_mutex is per DEVICE, and two threads of a forked child on two devices would
    // interleave their frames on the one socket.  (A pointer: a grandchild gets a fresh one, see start_helper.)
    std::mutex *mu = new std::mutex();
};
Helper &helper() {
    static Helper *h = new Helper();
    return *h;
}

std::mutex *helper_mutex(Helper &h) { return h.mu; }

std::string helper_path() {
    if (const char *e = getenv("SR_FORK_HELPER")) return e;
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&sr_last_error), &info) || !info.dli_fname)
        fail("fork helper: cannot locate lib/pygmm.so (dladdr failed)");
    std::string p(info.dli_fname);
    const size_t slash = p.rfind('/');
    p = slash == std::string::npos ? std::string(".") : p.substr(0, slash);
    return p + "/sr_fork_helper";
}

void reply_status(Reader &r) {
    const int32_t status = r.pod<int32_t>();
    const std::string msg = r.str();
    if (status != 0) fail("%s", msg.c_str());
}

// SIGTERM, then a BOUNDED wait: a helper stuck in an uninterruptible driver call (or stopped) must not hang the worker that drops
// it -- with its helper record's mutex held -- so after ~0.5 s it is killed outright and reaped (no zombie either way).
static void reap_helper(pid_t pid) {
    (void)::kill(pid, SIGTERM);
    int st = 0;
    for (int i = 0; i < 100; i++) {
        const pid_t r = ::waitpid(pid, &st, WNOHANG);
        if (r == pid || (r < 0 && errno != EINTR)) return;
        ::usleep(5000);
    }
    (void)::kill(pid, SIGKILL);
    for (int i = 0; i < 200; i++) {                 // (SIGKILL cannot be ignored; a D-state process still takes its time)
        const pid_t r = ::waitpid(pid, &st, WNOHANG);
        if (r == pid || (r < 0 && errno != EINTR)) return;
        ::usleep(5000);
    }
}

void start_helper(Helper &h) {
    // (a grandchild inherits its parent's Helper record: the socket in it belongs to the parent's conversation)
    if (h.fd >= 0 && h.owner_pid != (long)getpid()) {
        ::close(h.fd);
        h.fd = -1;
        h.pid = -1;
        h.sent.clear();
    }
    if (h.fd >= 0) return;
    const std::string exe = helper_path();
    if (::access(exe.c_str(), X_OK) != 0)
        fail("this process was forked after its parent had initialised the GPU runtime; its calls are served by a helper "
             "process, but '%s' is missing or not executable (built by `make -C speaker-recognition_amd/csrc`; SR_FORK_HELPER overrides the path)",
             exe.c_str());
    int sv[2];
    if (::socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, sv) != 0) fail("fork helper: socketpair failed (%s)", strerror(errno));
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, sv[1], 3);      // (dup2 clears close-on-exec on the copy)
    char arg_fd[] = "3";
    char *argv[] = {const_cast<char *>(exe.c_str()), arg_fd, nullptr};
    pid_t pid = -1;
    const int rc = ::posix_spawn(&pid, exe.c_str(), &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    ::close(sv[1]);
    if (rc != 0) {
        ::close(sv[0]);
        fail("fork helper: cannot start '%s' (%s)", exe.c_str(), strerror(rc));
    }
    h.fd = sv[0];
    h.pid = pid;
    h.owner_pid = (long)getpid();
    h.sent.clear();
    try {
        // handshake + the options this process had been given before it forked
        Writer w;
        w.pod<uint32_t>(MAGIC);
        w.pod<uint32_t>(OP_PING);
        w.flush(h.fd);
        Reader r;
        r.fill(h.fd);
        reply_status(r);
        for (const auto &kv : option_log()) {
            w.pod<uint32_t>(MAGIC);
            w.pod<uint32_t>(OP_OPTION);
            w.str(kv.first);
            w.pod<int64_t>(kv.second);
            w.flush(h.fd);
            r.fill(h.fd);
            reply_status(r);
        }
    } catch (...) {
        ::close(h.fd);
        h.fd = -1;
        reap_helper(pid);
        h.pid = -1;
        throw;
    }
}

void drop_helper(Helper &h) {
    if (h.fd >= 0) ::close(h.fd);
    h.fd = -1;
    if (h.pid > 0 && h.owner_pid == (long)getpid()) {
        // the helper leaves on end-of-file; it is told to as well in case it is stuck in a request, then reaped (no zombie)
        reap_helper((pid_t)h.pid);
    }
    h.pid = -1;
    h.sent.clear();
}

void put_header(Writer &w, Op op) {
    w.pod<uint32_t>(MAGIC);
    w.pod<uint32_t>(op);
    w.pod<int32_t>(current_device());
}

void put_model_tracked(Writer &w, Helper &h, const GMM &g, std::vector<std::pair<uint64_t, uint64_t>> &pending) {
    const uint64_t key = (uint64_t)(uintptr_t)&g;
    uint64_t hash;
    {
        const auto hit = h.hashed.find(key);
        if (hit != h.hashed.end() && hit->second.uid == g.uid && hit->second.generation == g.generation) {
            hash = hit->second.hash;
        } else {
            hash = model_hash(g);
            h.hashed[key] = Helper::Hashed{g.uid, g.generation, hash};
        }
    }
    const auto it = h.sent.find(key);
    bool have = g.trained() && it != h.sent.end() && it->second == hash;
    for (const auto &kv : pending)            // the same handle earlier in THIS request (a set may name a model twice): sent once
        if (kv.first == key && kv.second == hash) have = true;
    put_model(w, g, hash, !have);
    if (g.trained() && !have) pending.emplace_back(key, hash);
}

}  // namespace

// ---------------- client entry points (abi.cpp calls these when gpu_runtime_lost()) ----------------

// pthread_atfork child handler (common.cpp): only the forking thread exists; whoever held the record's mutex does not
void fork_proxy_atfork_child() { helper().mu = new std::mutex(); }

void fork_proxy_set_max_models(long n) { g_helper_max_models.store(n); }

void fork_proxy_note_option(const char *key, long value) {
    auto &log = option_log();
    bool known = false;
    for (auto &kv : log)
        if (kv.first == key) {
            kv.second = value;
            known = true;
        }
    if (!known) log.emplace_back(key, value);
    Helper &h = helper();
    if (!gpu_runtime_lost()) return;
    std::lock_guard<std::mutex> lock(*helper_mutex(h));
    if (h.fd >= 0 && h.owner_pid == (long)getpid()) {
        // the option IS set in this process (and logged for the next helper): a helper that died must not turn that into a failure
        try {
            Writer w;
            w.pod<uint32_t>(MAGIC);
            w.pod<uint32_t>(OP_OPTION);
            w.str(key);
            w.pod<int64_t>(value);
            w.flush(h.fd);
            Reader r;
            r.fill(h.fd);
            reply_status(r);
        } catch (const Error &e) {
            if (std::strncmp(e.what(), "fork helper:", 12) == 0) drop_helper(h);      // the next call starts a fresh one from the log
            else throw;                                                             // the helper refused the value: the caller hears it
        }
    }
}

void fork_proxy_score(GMM *g, const float *X, long n, int dim, float *ll_out, double *sum_out, int flags) {
    if (!g) fail("null GMM handle");
    if (!g->trained()) fail("GMM has no parameters yet (train or load it first)");
    if (dim != g->dim) fail("nr_dim %d does not match the model's dim %d", dim, g->dim);
    Helper &h = helper();
    std::lock_guard<std::mutex> lock(*helper_mutex(h));
    start_helper(h);
    for (int attempt = 0;; attempt++)
    try {
        std::vector<std::pair<uint64_t, uint64_t>> pending;
        Writer w;
        put_header(w, OP_SCORE);
        put_model_tracked(w, h, *g, pending);
        w.pod<int64_t>(n);
        w.pod<int32_t>(dim);
        w.pod<int32_t>(flags);
        w.pod<uint8_t>(ll_out ? 1 : 0);
        w.bytes(X, (size_t)n * dim * sizeof(float));
        w.flush(h.fd);
        Reader r;
        r.fill(h.fd);
        reply_status(r);
        for (const auto &kv : pending) h.sent[kv.first] = kv.second;
        const double sum = r.pod<double>();
        if (sum_out) *sum_out = sum;
        if (ll_out && n > 0) std::memcpy(ll_out, r.take((size_t)n * sizeof(float)), (size_t)n * sizeof(float));
        return;
    } catch (const Error &e) {
        // the helper keeps a bounded number of models: one it has evicted is sent again, once
        if (attempt == 0 && std::strncmp(e.what(), HELPER_MISS, sizeof HELPER_MISS - 1) == 0) {
            h.sent.erase((uint64_t)(uintptr_t)g);
            continue;
        }
        // an error raised BY the helper leaves the conversation intact; a broken conversation does not
        if (std::strncmp(e.what(), "fork helper:", 12) == 0) drop_helper(h);
        throw;
    }
}

// All of a speaker set's models on one utterance in ONE conversation and one fused pass in the helper (sr_score_models_f32):
// the reference's pool workers call predict_one per utterance (test-nperson.py:133-146), which through score_all is a
// conversation, a launch chain and a reply per SPEAKER -- 80 of them per utterance in its logged run.
void fork_proxy_score_models(GMM *const *models, int n_models, const float *X, long n, int dim, double *sums_out, int flags) {
    if (!models || n_models <= 0) fail("empty model list");
    if (!sums_out) fail("null sums_out");
    for (int i = 0; i < n_models; i++) {
        if (!models[i]) fail("null GMM handle in model list");
        if (!models[i]->trained()) fail("GMM has no parameters yet (train or load it first)");
        if (models[i]->dim != dim) fail("nr_dim %d does not match model %d's dim %d", dim, i, models[i]->dim);
    }
    Helper &h = helper();
    std::lock_guard<std::mutex> lock(*helper_mutex(h));
    start_helper(h);
    for (int attempt = 0;; attempt++)
    try {
        std::vector<std::pair<uint64_t, uint64_t>> pending;
        Writer w;
        put_header(w, OP_SCORE_MODELS);
        w.pod<int32_t>(n_models);
        for (int i = 0; i < n_models; i++) put_model_tracked(w, h, *models[i], pending);
        w.pod<int64_t>(n);
        w.pod<int32_t>(dim);
        w.pod<int32_t>(flags);
        w.bytes(X, (size_t)n * dim * sizeof(float));
        w.flush(h.fd);
        Reader r;
        r.fill(h.fd);
        reply_status(r);
        for (const auto &kv : pending) h.sent[kv.first] = kv.second;
        std::memcpy(sums_out, r.take((size_t)n_models * sizeof(double)), (size_t)n_models * sizeof(double));
        return;
    } catch (const Error &e) {
        // a model the helper has evicted: the whole list again with its parameters, once
        if (attempt == 0 && std::strncmp(e.what(), HELPER_MISS, sizeof HELPER_MISS - 1) == 0) {
            for (int i = 0; i < n_models; i++) h.sent.erase((uint64_t)(uintptr_t)models[i]);
            continue;
        }
        if (std::strncmp(e.what(), "fork helper:", 12) == 0) drop_helper(h);
        throw;
    }
}

int fork_proxy_train(GMM &gmm, const GMM *ubm, const float *X, long n, int dim, const Parameter &param, long seed) {
    Helper &h = helper();
    std::lock_guard<std::mutex> lock(*helper_mutex(h));
    start_helper(h);
    for (int attempt = 0;; attempt++)
    try {
        std::vector<std::pair<uint64_t, uint64_t>> pending;
        Writer w;
        put_header(w, OP_TRAIN);
        put_model(w, gmm, 0, true);                           // (about to change: never cached)
        w.pod<uint8_t>(ubm ? 1 : 0);
        if (ubm) put_model_tracked(w, h, *ubm, pending);
        w.pod<int64_t>(n);
        w.pod<int32_t>(dim);
        w.pod<int64_t>(seed);
        w.bytes(&param, sizeof(Parameter));
        int32_t rs[36];
        reference_rand_state(rs, false);
        w.bytes(rs, sizeof rs);
        w.bytes(X, (size_t)n * dim * sizeof(float));
        w.flush(h.fd);
        Reader r;
        r.fill(h.fd);
        reply_status(r);
        for (const auto &kv : pending) h.sent[kv.first] = kv.second;
        const int32_t n_iter = r.pod<int32_t>();
        std::memcpy(rs, r.take(sizeof rs), sizeof rs);
        reference_rand_state(rs, true);
        get_params(r, gmm);
        h.sent.erase((uint64_t)(uintptr_t)&gmm);
        return n_iter;
    } catch (const Error &e) {
        if (attempt == 0 && ubm && std::strncmp(e.what(), HELPER_MISS, sizeof HELPER_MISS - 1) == 0) {
            h.sent.erase((uint64_t)(uintptr_t)ubm);
            continue;
        }
        if (std::strncmp(e.what(), "fork helper:", 12) == 0) drop_helper(h);
        throw;
    }
}

}  // namespace sr

// ---------------- server side: what lib/sr_fork_helper runs ----------------
using namespace sr;

extern "C" int sr_fork_helper_main(int fd) {
    ::prctl(PR_SET_PDEATHSIG, SIGTERM);      // (and EOF on the socket: whichever comes first)
    struct Held {
        uint64_t hash = 0, tick = 0;
        std::unique_ptr<GMM> g;
    };
    std::unordered_map<uint64_t, Held> models;
    Held scratch;
    uint64_t clock = 0;
    // a model off the wire -> the helper's copy of it (kept per caller handle; rebuilt when the contents changed)
    auto take_model = [&](Reader &r, bool keep) -> GMM * {
        const uint64_t key = r.pod<uint64_t>();
        const uint64_t hash = r.pod<uint64_t>();
        const int K = r.pod<int32_t>();
        const int D = r.pod<int32_t>();
        const int state = r.pod<uint8_t>();
        if (state == 2) {
            const auto it = models.find(key);
            if (it == models.end() || !it->second.g || it->second.hash != hash) fail("%s: model %llx", HELPER_MISS, (unsigned long long)key);
            it->second.tick = ++clock;
            return it->second.g.get();
        }
        Held &slot = keep ? models[key] : scratch;       // a training target is single-use: it never enters the kept set
        if (keep && state == 1 && slot.g && slot.hash == hash && slot.g->nr_mixtures == K && slot.g->dim == D) {
            // parameters the helper already holds (sent again after a miss elsewhere in the list, or twice in one request): the
            // object stays -- an earlier entry of this request may point at it -- and its packed sets with it
            const size_t k = (size_t)K, kd = k * (size_t)D;
            r.take(k * sizeof(double));
            r.take(kd * sizeof(double));
            r.take(kd * sizeof(double));
            slot.tick = ++clock;
            return slot.g.get();
        }
        slot.tick = ++clock;
        slot.g = std::make_unique<GMM>();
        slot.hash = hash;
        slot.g->nr_mixtures = K;
        slot.g->dim = state == 1 ? D : 0;
        if (state == 1) {
            const size_t k = (size_t)K, kd = k * (size_t)D;
            const double *p = reinterpret_cast<const double *>(r.take(k * sizeof(double)));
            slot.g->weights.assign(p, p + k);
            p = reinterpret_cast<const double *>(r.take(kd * sizeof(double)));
            slot.g->mean.assign(p, p + kd);
            p = reinterpret_cast<const double *>(r.take(kd * sizeof(double)));
            slot.g->sigma.assign(p, p + kd);
        }
        return slot.g.get();
    };
    for (;;) {
        Reader r;
        try {
            r.fill(fd);
        } catch (const std::exception &) {
            return 0;                        // the forked child is gone
        }
        Writer w;
        try {
            if (r.pod<uint32_t>() != MAGIC) fail("fork helper: bad magic");
            const uint32_t op = r.pod<uint32_t>();
            if (op == OP_PING) {
                w.pod<int32_t>(0);
                w.str("");
            } else if (op == OP_OPTION) {
                const std::string key = r.str();
                const long value = (long)r.pod<int64_t>();
                if (sr_set_option(key.c_str(), value) != 0) fail("%s", sr_last_error());
                w.pod<int32_t>(0);
                w.str("");
            } else if (op == OP_SCORE) {
                set_default_device(r.pod<int32_t>());
                GMM *g = take_model(r, true);
                const long n = (long)r.pod<int64_t>();
                const int dim = r.pod<int32_t>();
                const int flags = r.pod<int32_t>();
                const bool want_ll = r.pod<uint8_t>() != 0;
                if (n < 0 || dim <= 0) fail("fork helper: bad frame matrix shape");
                const float *X = reinterpret_cast<const float *>(r.take((size_t)n * dim * sizeof(float)));
                std::vector<float> ll(want_ll ? (size_t)n : 0);
                double sum = 0.0;
                {
                    std::lock_guard<std::recursive_mutex> lock(api_mutex());
                    score_one_local(g, X, n, dim, want_ll ? ll.data() : nullptr, &sum, flags);
                }
                w.pod<int32_t>(0);
                w.str("");
                w.pod<double>(sum);
                if (want_ll) w.bytes(ll.data(), ll.size() * sizeof(float));
            } else if (op == OP_SCORE_MODELS) {
                set_default_device(r.pod<int32_t>());
                const int n_models = r.pod<int32_t>();
                if (n_models <= 0 || n_models > (1 << 20)) fail("fork helper: bad model count");
                std::vector<GMM *> ms((size_t)n_models);
                for (int i = 0; i < n_models; i++) ms[(size_t)i] = take_model(r, true);
                const long n = (long)r.pod<int64_t>();
                const int dim = r.pod<int32_t>();
                const int flags = r.pod<int32_t>();
                if (n < 0 || dim <= 0) fail("fork helper: bad frame matrix shape");
                const float *X = reinterpret_cast<const float *>(r.take((size_t)n * dim * sizeof(float)));
                std::vector<double> sums((size_t)n_models, 0.0);
                {
                    std::lock_guard<std::recursive_mutex> lock(api_mutex());
                    score_models_local(ms.data(), n_models, X, n, dim, sums.data(), flags);
                }
                w.pod<int32_t>(0);
                w.str("");
                w.bytes(sums.data(), sums.size() * sizeof(double));
            } else if (op == OP_TRAIN) {
                set_default_device(r.pod<int32_t>());
                GMM *g = take_model(r, false);
                const bool has_ubm = r.pod<uint8_t>() != 0;
                GMM *ubm = has_ubm ? take_model(r, true) : nullptr;
                const long n = (long)r.pod<int64_t>();
                const int dim = r.pod<int32_t>();
                const long seed = (long)r.pod<int64_t>();
                Parameter param;
                std::memcpy(&param, r.take(sizeof(Parameter)), sizeof(Parameter));
                int32_t rs[36];
                std::memcpy(rs, r.take(sizeof rs), sizeof rs);
                reference_rand_state(rs, true);
                if (n < 0 || dim <= 0) fail("fork helper: bad frame matrix shape");
                const float *X = reinterpret_cast<const float *>(r.take((size_t)n * dim * sizeof(float)));
                int n_iter;
                {
                    std::lock_guard<std::recursive_mutex> lock(api_mutex());
                    n_iter = train_em(*g, ubm, X, n, dim, param, seed);
                    if (n_iter < 0) fail("%s", last_error().c_str());
                }
                reference_rand_state(rs, false);
                w.pod<int32_t>(0);
                w.str("");
                w.pod<int32_t>(n_iter);
                w.bytes(rs, sizeof rs);
                put_params(w, *g);
            } else {
                fail("fork helper: unknown request %u", op);
            }
        } catch (const std::exception &e) {
            w.buf.clear();
            w.pod<int32_t>(-1);
            w.str(e.what());
        }
        try {
            w.flush(fd);
        } catch (const std::exception &) {
            return 0;
        }
        // training targets are single-use (scratch); the rest is a bounded least-recently-used set, so that a long-lived pool worker
        // enrolling speaker after speaker does not grow this process (and its packed device sets) without bound
        scratch = Held();
        auto kept_bytes = [&]() {
            size_t b = 0;
            for (const auto &kv : models)
                if (kv.second.g) b += (kv.second.g->weights.size() + kv.second.g->mean.size() + kv.second.g->sigma.size()) * sizeof(double);
            return b;
        };
        const size_t max_models = g_helper_max_models.load() > 0 ? (size_t)g_helper_max_models.load() : HELPER_MAX_MODELS;
        while (models.size() > max_models || (models.size() > 1 && kept_bytes() > HELPER_MAX_MODEL_BYTES)) {
            auto oldest = models.begin();
            for (auto it = models.begin(); it != models.end(); ++it)
                if (it->second.tick < oldest->second.tick) oldest = it;
            models.erase(oldest);
        }
    }
}
