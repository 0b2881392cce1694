// multi.cpp -- one host process, every GPU of the node: utterance-sharded prediction without torch,
// MPI or any collective ("one host thread + one hipStream pair per device").
//
// The reference's parallelism on this path is a thread pool inside the scoring call
// (Threadpool pool(concurrency), src/gmm/src/gmm.cc:533-560) and a process pool over utterances
// (multiprocessing.Pool, src/test/test-gmm.py:128-133).  Here the unit of sharding is the
// utterance (CMVN, deltas and the per-utterance sums need whole utterances and nothing else):
// utterances are dealt to the slots greedily by length, every slot owns a replica of the models
// and of the extractor tables on its GPU, a host thread per slot runs PCM upload -> MFCC -> CMVN /
// deltas -> all models -> sums + argmax on that GPU's stream, and the host concatenates the
// per-utterance rows.  Bytes that cross between GPUs: none.
//
// A slot is bound to device `slot % visible devices`, so asking for more slots than GPUs is legal:
// the surplus slots share a GPU (serialised by that device's lock) -- how the threading is tested
// on a single-GPU box.
#include "../../include/pygmm_hip.h"

#include "batch.hpp"
#include "common.hpp"
#include "gmm_model.hpp"
#include "mfcc.hpp"
#include "score.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <thread>

using namespace sr;

struct SRMulti {
    struct Slot {
        int device = 0;
        std::unique_ptr<SRModelSet> set;
        std::unique_ptr<SRBatch> pcm, feat;
        std::vector<int16_t> host_pcm;      // this slot's utterances, concatenated
        std::vector<int64_t> offsets;
        std::vector<int> utts;              // global utterance indices, in slot order
        std::vector<double> sums;
        std::vector<int> argmax;
        std::string error;
        double seconds = 0.0;               // wall time of the slot's last pass
    };
    std::unique_ptr<SRMfcc> mfcc;           // host tables shared; device tables per GPU inside
    std::vector<Slot> slots;
    int n_models = 0;
};

namespace {

// Longest-first greedy assignment by sample count (what shard.partition_utterances does in Python).
void partition(const int64_t *off, int n_utt, std::vector<SRMulti::Slot> &slots) {
    std::vector<int> order(n_utt);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    std::vector<int64_t> load(slots.size(), 0);
    for (auto &s : slots) s.utts.clear();
    for (int u : order) {
        const size_t k = std::min_element(load.begin(), load.end()) - load.begin();
        slots[k].utts.push_back(u);
        load[k] += off[u + 1] - off[u];
    }
    for (auto &s : slots) std::sort(s.utts.begin(), s.utts.end());
}

void run_slot(SRMulti *m, SRMulti::Slot &s, const int16_t *pcm, const int64_t *off, int nd, int flags) {
    try {
        set_thread_device(s.device);
        std::lock_guard<std::recursive_mutex> lock(api_mutex());
        ensure_device();
        const auto t0 = std::chrono::steady_clock::now();
        const int U = (int)s.utts.size();
        s.offsets.assign(U + 1, 0);
        for (int i = 0; i < U; i++) s.offsets[i + 1] = s.offsets[i] + (off[s.utts[i] + 1] - off[s.utts[i]]);
        s.host_pcm.resize((size_t)s.offsets[U]);
        for (int i = 0; i < U; i++)
            std::memcpy(s.host_pcm.data() + s.offsets[i], pcm + off[s.utts[i]],
                        sizeof(int16_t) * (size_t)(s.offsets[i + 1] - s.offsets[i]));
        if (!s.pcm) s.pcm = std::make_unique<SRBatch>();
        if (!s.feat) s.feat = std::make_unique<SRBatch>();
        SRBatch &b = *s.pcm;
        b.bind_device();
        b.kind = SRBatch::PCM16;
        b.n_utt = U;
        b.offsets = s.offsets;
        b.n_rows = s.offsets[U];
        b.tile_tables.clear();
        b.pcm16.upload(s.host_pcm.data(), s.host_pcm.size());
        b.d_offsets.upload(b.offsets.data(), b.offsets.size());
        sync_stream();
        s.sums.assign((size_t)U * m->n_models, 0.0);
        s.argmax.assign((size_t)U, -1);
        if (U > 0) predict_pcm(m->mfcc.get(), s.set.get(), &b, nd, s.sums.data(), s.argmax.data(), flags);
        s.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (const std::exception &e) {
        s.error = e.what();
    } catch (...) {
        s.error = "unknown C++ exception";
    }
}

}  // namespace

extern "C" {

SRMulti *sr_multi_create(GMM *const *models, int n_models, double fs, double win_length_ms,
                         double win_shift_ms, int fft_size, int n_filters, int n_ceps,
                         double pre_emphasis, int n_slots) {
    try {
        if (!models || n_models <= 0) fail("empty model list");
        const int visible = visible_devices();
        if (visible <= 0) fail("no HIP device available; lib/pygmm.so has no CPU path");
        if (n_slots <= 0) n_slots = visible;
        if (n_slots > 64) fail("at most 64 slots");
        std::vector<const GMM *> v(models, models + n_models);
        for (auto *g : v)
            if (!g) fail("null GMM handle in model list");
        auto m = std::make_unique<SRMulti>();
        m->mfcc = std::make_unique<SRMfcc>(fs, win_length_ms, win_shift_ms, fft_size, n_filters, n_ceps, pre_emphasis);
        m->n_models = n_models;
        m->slots.resize((size_t)n_slots);
        const int prev = current_device();
        // replicate the models: packed once per slot on that slot's GPU (threads: packing a
        // 1000-speaker set takes seconds)
        std::vector<std::thread> th;
        for (int i = 0; i < n_slots; i++) {
            m->slots[i].device = i % visible;
            th.emplace_back([&, i]() {
                auto &s = m->slots[i];
                try {
                    set_thread_device(s.device);
                    std::lock_guard<std::recursive_mutex> lock(api_mutex());
                    s.set = std::make_unique<SRModelSet>();
                    pack_model_set(*s.set, v);
                    upload_model_set(*s.set);
                } catch (const std::exception &e) {
                    s.error = e.what();
                }
            });
        }
        for (auto &t : th) t.join();
        set_thread_device(prev);
        for (auto &s : m->slots)
            if (!s.error.empty()) fail("device %d: %s", s.device, s.error.c_str());
        return m.release();
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return nullptr;
    }
}

void sr_multi_free(SRMulti *m) {
    if (!m) return;
    const int prev = current_device();
    for (auto &s : m->slots) {
        try {
            set_thread_device(s.device);
            std::lock_guard<std::recursive_mutex> lock(api_mutex());
            (void)hipSetDevice(s.device);
            s.set.reset();
            s.pcm.reset();
            s.feat.reset();
        } catch (...) {
        }
    }
    try { set_thread_device(prev); } catch (...) {}
    delete m;
}

int sr_multi_slots(SRMulti *m) { return m ? (int)m->slots.size() : 0; }
int sr_multi_slot_device(SRMulti *m, int slot) {
    return (m && slot >= 0 && slot < (int)m->slots.size()) ? m->slots[slot].device : -1;
}

int sr_multi_predict_pcm(SRMulti *m, const int16_t *pcm, const int64_t *sample_offsets, int n_utt,
                         int nd, double *sums_out, int *argmax_out, double *slot_seconds_out, int flags) {
    try {
        if (!m || !sample_offsets || n_utt < 0) fail("bad arguments to sr_multi_predict_pcm");
        if (sample_offsets[0] != 0) fail("sample_offsets[0] must be 0");
        for (int u = 0; u < n_utt; u++)
            if (sample_offsets[u + 1] < sample_offsets[u]) fail("sample_offsets must be non-decreasing");
        if (sample_offsets[n_utt] > 0 && !pcm) fail("null PCM pointer");
        partition(sample_offsets, n_utt, m->slots);
        std::vector<std::thread> th;
        for (auto &s : m->slots) {
            s.error.clear();
            th.emplace_back(run_slot, m, std::ref(s), pcm, sample_offsets, nd, flags);
        }
        for (auto &t : th) t.join();
        for (auto &s : m->slots)
            if (!s.error.empty()) fail("device %d: %s", s.device, s.error.c_str());
        const int S = m->n_models;
        for (size_t k = 0; k < m->slots.size(); k++) {
            const auto &s = m->slots[k];
            for (size_t i = 0; i < s.utts.size(); i++) {
                const int u = s.utts[i];
                if (sums_out) std::memcpy(sums_out + (size_t)u * S, s.sums.data() + i * S, sizeof(double) * S);
                if (argmax_out) argmax_out[u] = s.argmax[i];
            }
            if (slot_seconds_out) slot_seconds_out[k] = s.seconds;
        }
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

}  // extern "C"
