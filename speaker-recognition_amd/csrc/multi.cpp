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
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>
#include <numeric>
#include <system_error>
#include <thread>

using namespace sr;

constexpr int MULTI_CHUNKS = 8;         // a slot's utterances are uploaded and scored in up to this many pieces
constexpr int MULTI_DEFAULT_PIECES = 8; // ... and in this many when copy and kernels are about as long: eight, each 1.1 x the one before
constexpr double MULTI_MILD_GROWTH = 1.1;   // (round 6, configs[1] from page-locked PCM, 12 calls each in one session: 6 equal pieces 7.4-7.8 ms,
                                        // 8 equal 7.3-7.5, 8 x 1.1 7.13-7.30, 8 x 1.2 7.26-7.43, 8 x 1.3 7.5-7.6, 5 x 1.25 7.8-7.9, 4 x 1.5 8.4-8.6:
                                        // the kernels of a piece -- 1.15 ms against its 0.95 ms of link time -- are the longer leg by a little)
constexpr double MULTI_GROWTH = 3.0;    // kernel-bound slots (run_slot): every piece this many times everything before it
std::atomic<int> &multi_merge_option() {     // sr_set_option("multi_merge_same_device", 0 | 1)
    static std::atomic<int> v{1};
    return v;
}

struct SRMulti {
    // One piece of a slot's utterances: its PCM on the device, the feature stage's workspace and output, page-locked result
    // buffers.  Everything a piece needs is its own, so that pieces of different shapes -- and of different slots on one device --
    // never re-upload a cached table (that would synchronise the stream in the middle of the pipeline).
    struct Chunk {
        std::unique_ptr<SRBatch> pcm;
        PinnedBuf<int16_t> staging;         // only used when the caller's PCM is not page-locked
        hipEvent_t uploaded = nullptr, done = nullptr;
        int u0 = 0, u1 = 0;                 // range of the slot's utterances
        SRBatch feat;
        MfccScratch *scratch = nullptr;
        PinnedBuf<double> h_sums;           // (+ room for the argmax values right behind the sums: they come in one copy, as they lie in the workspace)
        int *h_arg = nullptr;               // where this pass's argmax values landed (behind the sums, or h_argmax)
        PinnedBuf<int> h_argmax, h_flags;   // h_flags: {a frame saturated the fp16 engine, (tile, model) pairs in the partial-product band}
        // the piece's list of (tile, model) pairs in the band, set aside on the device (the scoring workspace it was produced in
        // belongs to the next piece by then): what gmm_flush.hip re-evaluates when it is not empty
        DevBuf<int2> d_list;
        const TileTable *tiles = nullptr;
        int flush_cap = 0;
    };
    struct Slot {
        int device = 0;
        std::unique_ptr<SRModelSet> set;
        Chunk chunk[MULTI_CHUNKS];
        std::vector<int64_t> offsets;
        std::vector<int> utts;              // global utterance indices, in slot order
        std::vector<double> sums;
        std::vector<int> argmax;
        std::string error;
        double seconds = 0.0;               // wall time of the slot's last pass
        int numa_node = -1;                 // where the slot's host thread was pinned (-1: nowhere)
        // what the last passes told about this slot's work: device time per PCM byte against the link's time per byte (rho >= 1:
        // the kernels are the longer leg) on a batch of rho_samples samples -- two passes in a row that agree change the shape of
        // the next pass's pieces (run_slot): 0 = equal pieces, 1 = growing pieces
        int schedule = 0, votes = 0;
        int64_t rho_samples = 0;
    };
    std::unique_ptr<SRMfcc> mfcc;           // host tables shared; device tables per GPU inside
    std::deque<Slot> slots;                // (a slot owns page-locked buffers and events: not movable)
    int n_models = 0;
};

namespace {

// Utterances -> slots.  Many utterances that are small against a slot's share: contiguous ranges of about equal sample
// counts, in the caller's order -- a slot's PCM is then ONE run of the caller's buffer and travels as a few large copies (dealt
// round-robin, 1000 equal utterances over 2 slots were 1000 copies of 320 KB: 15 ms of copy calls for 6 ms of PCIe time).
// Few or very uneven utterances: longest-first greedy by sample count (what shard.partition_utterances does in Python).
void partition(const int64_t *off, int n_utt, std::vector<SRMulti::Slot *> &slots) {
    if (n_utt == 0 || slots.empty()) return;
    const int64_t total = off[n_utt];
    int64_t longest = 0;
    for (int u = 0; u < n_utt; u++) longest = std::max(longest, off[u + 1] - off[u]);
    const size_t ns = slots.size();
    if (longest * 8 * (int64_t)ns <= total) {
        int u = 0;
        for (size_t k = 0; k < ns; k++) {
            const int64_t hi = total * (int64_t)(k + 1) / (int64_t)ns;
            while (u < n_utt && (k + 1 == ns || off[u + 1] <= hi)) slots[k]->utts.push_back(u++);
        }
        return;
    }
    std::vector<int> order(n_utt);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    std::vector<int64_t> load(ns, 0);
    for (int u : order) {
        const size_t k = std::min_element(load.begin(), load.end()) - load.begin();
        slots[k]->utts.push_back(u);
        load[k] += off[u + 1] - off[u];
    }
    for (auto *s : slots) std::sort(s->utts.begin(), s->utts.end());
}

// Pageable caller memory -> the page-locked staging buffer.  One thread's memcpy (~25 GB/s) is slower than the link it feeds
// (55 GB/s): copies of more than a few MB are cut over MULTI_STAGING_THREADS short-lived threads (they inherit the slot
// thread's placement next to its GPU).
constexpr int MULTI_STAGING_THREADS = 4;
void staging_copy(void *dst, const void *src, size_t bytes) {
    constexpr size_t PART_MIN = (size_t)4 << 20;
    const int parts = (int)std::min<size_t>(MULTI_STAGING_THREADS, bytes / PART_MIN);
    if (parts <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
    struct Joiner {                          // (a thread that could not be started leaves its part to this one)
        std::vector<std::thread> th;
        ~Joiner() {
            for (auto &t : th) t.join();
        }
    } helpers;
    size_t mine_hi = std::min(bytes, per);   // this thread copies [0, mine_hi) and whatever nobody else took
    std::vector<std::pair<size_t, size_t>> left;
    for (int p = 1; p < parts; p++) {
        const size_t lo = std::min(bytes, per * p), hi = p + 1 == parts ? bytes : std::min(bytes, per * (p + 1));
        try {
            helpers.th.emplace_back([=] { std::memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
        } catch (const std::system_error &) {
            left.emplace_back(lo, hi);
        }
    }
    std::memcpy(dst, src, mine_hi);
    for (const auto &r : left) std::memcpy((char *)dst + r.first, (const char *)src + r.first, r.second - r.first);
}

// true when [p, p + bytes) is page-locked host memory the copy engines can read directly (hipHostMalloc / hipHostRegister
// / sr_host_register): then the slots DMA straight out of the caller's buffer
bool host_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();           // (an ordinary malloc pointer is "invalid value" to the runtime: not an error here)
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// One slot: its utterances cut into up to MULTI_CHUNKS pieces of whole utterances.  Every piece goes host -> device on the
// device's copy stream (queued in order, an event behind it) -- from the caller's own memory when that is page-locked, else
// through a page-locked staging buffer this thread fills one piece ahead of the copy engine -- and its kernels (MFCC, CMVN /
// deltas, all models, finalize) plus the copies of its results into page-locked buffers are ENQUEUED on the main stream behind
// that event, piece after piece, without a host synchronisation in between: the copy of piece i + 1 runs under the kernels of
// piece i, and the host waits once, at the end.  (Round 3 scored the pieces one synchronous call each: four waits per slot and a
// quarter of the PCM uploaded before the first kernel -- 8.8 ms on configs[1] where copy and kernels are 5.8 and 5.7 ms.)  What
// needs the host in the loop -- a frame that saturated the fp16 engine, frames in the band of the reference's partial-product
// flushes (lse.hpp) -- is noticed in the piece's flags afterwards and that piece is scored again, synchronously, from its
// features, which are still on the device (as csrc/stream.cpp does for a serving tick).
// The device's lock is taken piece by piece, so slots that share a GPU interleave on its stream.
void run_slot(SRMulti *m, SRMulti::Slot &s, const int16_t *pcm, const int64_t *off, int nd, int flags, bool pinned,
              double *sums_out, int *argmax_out) {
    auto drain = [&]() {                   // nothing of this call may still be reading the caller's buffer when it returns
        try {
            (void)hipStreamSynchronize(ctx().copy);
            (void)hipStreamSynchronize(ctx().main);
        } catch (...) {
        }
    };
    try {
        set_thread_device(s.device);
        ensure_device();
        // this thread fills staging buffers and waits on this GPU's events: next to its PCIe root unless told not to
        s.numa_node = numa_bind_option().load() ? bind_thread_near_device(s.device) : device_numa_node(s.device);
        const auto t0 = std::chrono::steady_clock::now();
        const int U = (int)s.utts.size();
        const int S = m->n_models;
        s.offsets.assign(U + 1, 0);
        for (int i = 0; i < U; i++) s.offsets[i + 1] = s.offsets[i] + (off[s.utts[i] + 1] - off[s.utts[i]]);
        if (!sums_out) s.sums.assign((size_t)U * S, 0.0);
        if (!argmax_out) s.argmax.assign((size_t)U, -1);
        // piece boundaries: whole utterances, about equal sample counts; pieces of at least ~2 MB of PCM (smaller ones are
        // all launch overhead and kernel tails)
        const int64_t total = s.offsets[U];
        const int want = MULTI_DEFAULT_PIECES;
        // Copy and kernels take about the same time on this path (configs[1]: 5.6 and 5.3 ms), so the call ends at about
        // copy(everything) + kernels(last piece): equal pieces, enough of them that the last one is short and few enough that
        // the per-piece launches do not add up (round 4's sweep, HISTORY.md section 5; page-locked PCM: 1 piece 11.3 ms, 2 8.7,
        // 4 7.5, 6 7.2, 8 7.25; a small-first / small-last shape, round 4's first attempt, 7.7)
        //
        // Round 6: (nearly) equal pieces -- eight now, each MULTI_MILD_GROWTH x the one before: with the float64 feature stage a piece's
        // kernels are the longer leg by a fifth -- are right when copy and kernels are about as long.  When the kernels are the longer leg by a
        // factor rho (configs[2]: 3.2 GB = 58 ms of link time under 280 ms of kernels, rho ~ 4.8) the only exposed copy is the
        // FIRST piece's, and a piece may be rho times everything before it without the device ever waiting for its bytes:
        // cumulative shares S_k = rho S_{k-1} + s_0, S_{n-1} = 1  =>  s_0 = (rho - 1) / (rho^n - 1).  Two shapes only (a new
        // shape means new buffers and tables for every piece): the balanced one above, and four pieces growing by MULTI_GROWTH = 3
        // (2.5 / 7.5 / 22.5 / 67.5 %) once two passes in a row on a batch of about this size measured rho >= 3.5; back to the
        // balanced one when two in a row measure < 2.5.  Pieces are whole utterances and an utterance's results do not depend on the batch
        // around it: the bits are the same for any cut.
        // The FIRST pass on a batch of this size starts from an estimate (the measured adaptation took four calls to settle --
        // two equal-piece passes, one that allocated the new pieces' buffers, one more -- 350 / 337 / 454 / 335 ms before 276 on
        // configs[2]): device seconds per frame from the set's arithmetic at the rate its engine class sustains, against the
        // link's seconds per frame.  configs[2]: 16.7 Mflop per frame / 700 TFLOP/s + MFCC 2.7 ns = 26.5 ns against 5.8 ns of
        // link: 4.6; configs[1]: 0.93.  The votes below correct a wrong guess.
        if (s.rho_samples == 0 || !(total > s.rho_samples / 2 && total < s.rho_samples * 2)) {
            const SRModelSet &set = *s.set;
            double mixtures = 0.0;                         // of all models together (padded to whole records of KB)
            for (const ChunkDesc &cd : set.host.chunks) mixtures += (double)cd.n_records * KB;
            const double flops = mixtures * (4.0 * set.host.dim + 6.0);          // per frame
            const double rate = !set.h2s.params.empty() ? 700e12 : (!set.h2.params.empty() || !set.bx3.params.empty() || !set.shared.params.empty()) ? 350e12 : 60e12;
            const double dev_s = flops / rate + 2.7e-9;
            const double link_s = (double)m->mfcc->frame_shift * sizeof(int16_t) / 55e9;
            s.schedule = dev_s / link_s >= 3.5 ? 1 : 0;
            s.votes = 0;
        }
        const double rho = s.schedule ? MULTI_GROWTH : MULTI_MILD_GROWTH;
        const int want_n = s.schedule ? 4 : want;
        const int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want_n, U), total / ((int64_t)1 << 20)));
        double cum[MULTI_CHUNKS + 1];                    // cumulative shares S_k, cum[n_chunks] = 1
        {
            const double s0 = rho > 1.0 + 1e-9 ? (rho - 1.0) / (std::pow(rho, n_chunks) - 1.0) : 1.0 / n_chunks;
            cum[0] = 0.0;
            for (int c = 1; c <= n_chunks; c++) cum[c] = rho * cum[c - 1] + s0;
            for (int c = 1; c <= n_chunks; c++) cum[c] = std::min(1.0, cum[c] / cum[n_chunks]);
        }
        for (int c = 0; c < MULTI_CHUNKS; c++) {
            auto &ch = s.chunk[c];
            ch.u0 = ch.u1 = 0;
            if (c >= n_chunks) continue;
            const int64_t lo = (int64_t)((double)total * cum[c]), hi = (int64_t)((double)total * cum[c + 1]);
            ch.u0 = c == 0 ? 0 : (int)(std::lower_bound(s.offsets.begin(), s.offsets.end(), lo) - s.offsets.begin());
            ch.u1 = c == n_chunks - 1 ? U : (int)(std::lower_bound(s.offsets.begin(), s.offsets.end(), hi) - s.offsets.begin());
            ch.u0 = std::min(ch.u0, U);
            ch.u1 = std::max(ch.u0, std::min(ch.u1, U));
        }
        for (int c = 1; c < n_chunks; c++) s.chunk[c].u0 = s.chunk[c - 1].u1;      // contiguous cover
        // ---- upload: every piece queued on the copy stream, an event behind it
        for (int c = 0; c < n_chunks; c++) {
            auto &ch = s.chunk[c];
            if (!ch.pcm) ch.pcm = std::make_unique<SRBatch>();
            if (!ch.uploaded) SR_HIP(hipEventCreateWithFlags(&ch.uploaded, hipEventDisableTiming));
            if (!ch.done) SR_HIP(hipEventCreateWithFlags(&ch.done, hipEventDisableTiming));
            if (!ch.scratch) ch.scratch = mfcc_scratch_new();
            SRBatch &b = *ch.pcm;
            const int nu = ch.u1 - ch.u0;
            const int64_t base = s.offsets[ch.u0], n_samp = s.offsets[ch.u1] - base;
            {
                std::lock_guard<std::recursive_mutex> lock(api_mutex());   // (the batch's buffers may be reallocated: not under a kernel)
                b.bind_device();
                std::vector<int64_t> po((size_t)nu + 1, 0);
                for (int i = 0; i <= nu; i++) po[i] = s.offsets[ch.u0 + i] - base;
                if (b.kind != SRBatch::PCM16 || b.offsets != po || !b.d_offsets.p) {   // a serving loop repeats its shape: nothing to redo
                    b.kind = SRBatch::PCM16;
                    b.n_utt = nu;
                    b.offsets = po;
                    b.n_rows = n_samp;
                    b.invalidate_tiles();
                    b.pcm16.ensure((size_t)std::max<int64_t>(1, n_samp));
                    b.d_offsets.upload(b.offsets.data(), b.offsets.size());
                    sync_stream();
                }
                ch.h_sums.ensure((size_t)std::max(1, nu) * S + ((size_t)std::max(1, nu) + 1) / 2);
                ch.h_argmax.ensure((size_t)std::max(1, nu));
                ch.h_flags.ensure(2);
                ch.h_flags.p[0] = ch.h_flags.p[1] = 0;         // (nothing of this piece is in flight: the previous call waited for it)
            }
            if (!pinned) ch.staging.ensure((size_t)std::max<int64_t>(1, n_samp));
            // runs of utterances that are neighbours in the caller's buffer travel as one copy
            int i = ch.u0;
            while (i < ch.u1) {
                int j = i;
                while (j + 1 < ch.u1 && s.utts[j + 1] == s.utts[j] + 1) j++;
                const int64_t src0 = off[s.utts[i]], n = off[s.utts[j] + 1] - src0, dst0 = s.offsets[i] - base;
                if (n > 0) {
                    const int16_t *src = pcm + src0;
                    if (!pinned) {
                        staging_copy(ch.staging.p + dst0, src, sizeof(int16_t) * (size_t)n);
                        src = ch.staging.p + dst0;
                    }
                    SR_HIP(hipMemcpyAsync(b.pcm16.p + dst0, src, sizeof(int16_t) * (size_t)n, hipMemcpyHostToDevice, ctx().copy));
                }
                i = j + 1;
            }
            SR_HIP(hipEventRecord(ch.uploaded, ctx().copy));
            // ---- its kernels and result copies behind the event: launches only
            if (nu == 0) continue;
            std::lock_guard<std::recursive_mutex> lock(api_mutex());
            SR_HIP(hipStreamWaitEvent(ctx().stream, ch.uploaded, 0));
            mfcc_extract_with(*m->mfcc, b, nd, 1, ch.feat, ch.scratch);
            const ScoreResult r = score_device(*s.set, ch.feat, false, flags);
            // (the pass's two counters and its sums + argmax lie side by side in the workspace: one copy each instead of two -- a copy
            // is ~8 us on the stream, and eight pieces' small operations are what keeps the call above max(copy, kernels))
            const bool flags_together = r.d_oor && r.d_flush_count == r.d_oor + 1;
            if (flags_together) SR_HIP(hipMemcpyAsync(ch.h_flags.p, r.d_oor, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
            else if (r.d_oor) SR_HIP(hipMemcpyAsync(ch.h_flags.p, r.d_oor, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
            ch.tiles = r.tiles;
            ch.flush_cap = 0;
            if (r.d_flush_count) {
                // frames in the band of the reference's partial-product flushes are the NORMAL case on some workloads (synthetic
                // speech against random models: ~2 k pairs per 10 M frames): keep what resolving them needs, a few MB device to device
                if (!flags_together) SR_HIP(hipMemcpyAsync(ch.h_flags.p + 1, r.d_flush_count, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
                ch.d_list.ensure((size_t)std::max(1, r.flush_cap));
                ch.flush_cap = r.flush_cap;
                SR_HIP(hipMemcpyAsync(ch.d_list.p, r.d_flush_list, (size_t)r.flush_cap * sizeof(int2), hipMemcpyDeviceToDevice, ctx().stream));
            }
            if ((const void *)r.d_argmax == (const void *)(r.d_sums + (size_t)nu * S)) {
                SR_HIP(hipMemcpyAsync(ch.h_sums.p, r.d_sums, (size_t)nu * S * sizeof(double) + (size_t)nu * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
                ch.h_arg = reinterpret_cast<int *>(ch.h_sums.p + (size_t)nu * S);
            } else {
                SR_HIP(hipMemcpyAsync(ch.h_sums.p, r.d_sums, (size_t)nu * S * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
                SR_HIP(hipMemcpyAsync(ch.h_argmax.p, r.d_argmax, (size_t)nu * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
                ch.h_arg = ch.h_argmax.p;
            }
            SR_HIP(hipEventRecord(ch.done, ctx().stream));
        }
        // ---- collect: one wait per piece, in order; the rare piece that needs the host is redone from its features
        for (int c = 0; c < n_chunks; c++) {
            auto &ch = s.chunk[c];
            const int nu = ch.u1 - ch.u0;
            if (nu == 0) continue;
            SR_HIP(hipEventSynchronize(ch.done));
            if (ch.h_flags.p[0] == 0 && ch.h_flags.p[1] > 0 && ch.h_flags.p[1] <= ch.flush_cap) {
                // pairs in the band, nothing else: re-evaluate exactly those with the reference's arithmetic and complete the
                // piece's results where they are, in host memory -- on the device's SECOND stream (what it reads -- the piece's
                // features, the models, the list -- is nobody else's), so that its one host wait does not wait for the later
                // pieces' kernels
                std::lock_guard<std::recursive_mutex> lock(api_mutex());
                StreamScope side(ctx().aux);
                flush_resolve_host(*s.set, ch.feat, *ch.tiles, ch.d_list.p, ch.h_flags.p[1], ch.h_sums.p, ch.h_arg);
            } else if (ch.h_flags.p[0] != 0 || ch.h_flags.p[1] != 0) {
                // a frame saturated the fp16 engine, or the list overflowed: this piece again, synchronously, from its features
                std::lock_guard<std::recursive_mutex> lock(api_mutex());
                const int fl = flags | (ch.h_flags.p[0] != 0 ? SCORE_PRECISE : 0);
                ScoreResult r = score_device(*s.set, ch.feat, false, fl);
                ch.h_arg = ch.h_argmax.p;
                if (!fetch_results(*s.set, ch.feat, fl, r, ch.h_sums.p, ch.h_argmax.p, nullptr)) {
                    r = score_device(*s.set, ch.feat, false, fl | SCORE_PRECISE);
                    fetch_results(*s.set, ch.feat, fl | SCORE_PRECISE, r, ch.h_sums.p, ch.h_argmax.p, nullptr);
                }
            }
            // the piece's rows go straight to the caller's arrays (runs of neighbouring utterances as one copy)
            for (int i = ch.u0; i < ch.u1;) {
                int j = i;
                while (j + 1 < ch.u1 && s.utts[j + 1] == s.utts[j] + 1) j++;
                const size_t n = (size_t)(j + 1 - i), at = (size_t)(i - ch.u0);
                if (sums_out) std::memcpy(sums_out + (size_t)s.utts[i] * S, ch.h_sums.p + at * S, n * S * sizeof(double));
                else std::memcpy(s.sums.data() + (size_t)i * S, ch.h_sums.p + at * S, n * S * sizeof(double));
                if (argmax_out) std::memcpy(argmax_out + s.utts[i], ch.h_arg + at, n * sizeof(int));
                else std::memcpy(s.argmax.data() + i, ch.h_arg + at, n * sizeof(int));
                i = j + 1;
            }
        }
        SR_HIP(hipStreamSynchronize(ctx().copy));              // (pieces without utterances still queued their empty copies)
        s.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // this pass's device time per byte against the link's (55 GB/s, what page-locked copies reach on this platform): everything
        // but the first piece's upload is kernels when rho >= 1, and when it is not the estimate only has to stay below 1
        if (total > 0 && n_chunks > 1) {
            const double link_s = (double)total * sizeof(int16_t) / 55e9;
            const double first = link_s * (double)(s.offsets[s.chunk[0].u1]) / (double)total;
            const double rho_seen = std::max(0.0, s.seconds - first) / link_s;
            const bool change = s.schedule ? rho_seen < 2.5 : rho_seen >= 3.5;
            s.votes = change ? s.votes + 1 : 0;
            if (s.votes >= 2) {
                s.schedule ^= 1;
                s.votes = 0;
            }
            s.rho_samples = total;
        }
    } catch (const std::exception &e) {
        s.error = e.what();
        drain();
    } catch (...) {
        s.error = "unknown C++ exception";
        drain();
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
    if (!m || gpu_runtime_lost()) return;     // (a forked child leaves its parent's device state alone: common.hpp)
    const int prev = current_device();
    for (auto &s : m->slots) {
        try {
            set_thread_device(s.device);
            std::lock_guard<std::recursive_mutex> lock(api_mutex());
            (void)hipSetDevice(s.device);
            s.set.reset();
            for (auto &ch : s.chunk) {
                ch.pcm.reset();
                if (ch.uploaded) (void)hipEventDestroy(ch.uploaded);
                if (ch.done) (void)hipEventDestroy(ch.done);
                ch.uploaded = ch.done = nullptr;
                if (ch.scratch) mfcc_scratch_delete(ch.scratch);
                ch.scratch = nullptr;
                ch.feat = SRBatch();
                ch.d_list.release();
            }
        } catch (...) {
        }
    }
    try { set_thread_device(prev); } catch (...) {}
    delete m;
}

// Page-locks caller memory (a serving loop's PCM ring, say) so that sr_multi_predict_pcm's copy engines read it in place.
int sr_host_register(void *p, size_t bytes) {
    try {
        ensure_device();
        if (!p || bytes == 0) fail("bad arguments to sr_host_register");
        SR_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}
int sr_host_unregister(void *p) {
    try {
        ensure_device();
        SR_HIP(hipHostUnregister(p));
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

int sr_multi_slots(SRMulti *m) { return m ? (int)m->slots.size() : 0; }
int sr_multi_slot_numa_node(SRMulti *m, int slot) {
    return (m && slot >= 0 && slot < (int)m->slots.size()) ? m->slots[slot].numa_node : -1;
}
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
        if (gpu_runtime_lost()) fail_gpu_runtime_lost("sr_multi_predict_pcm");
        // Slots that share a device are ONE queue on it (round 4): the device's lock would serialise their pieces anyway, in
        // an order nobody chose, with both slots' tails at the end.  The first slot of a device takes the work of all of them
        // (sr_set_option("multi_merge_same_device", 0): every slot its own share and thread -- what the tests of the threaded
        // path on a one-GPU box use).
        std::vector<SRMulti::Slot *> active;
        for (auto &s : m->slots) {
            s.utts.clear();
            s.seconds = 0.0;
            bool first = true;
            if (multi_merge_option().load())
                for (auto *a : active) first = first && a->device != s.device;
            if (first) active.push_back(&s);
        }
        partition(sample_offsets, n_utt, active);
        const bool pinned = sample_offsets[n_utt] > 0 && host_pinned(pcm) &&
                            host_pinned(pcm + sample_offsets[n_utt] - 1);
        std::vector<std::thread> th;
        for (auto &s : m->slots) s.error.clear();
        // (every slot writes its utterances' rows into the caller's arrays itself: disjoint rows)
        for (auto *s : active) th.emplace_back(run_slot, m, std::ref(*s), pcm, sample_offsets, nd, flags, pinned, sums_out, argmax_out);
        for (auto &t : th) t.join();
        for (auto &s : m->slots)
            if (!s.error.empty()) fail("device %d: %s", s.device, s.error.c_str());
        for (size_t k = 0; k < m->slots.size(); k++)
            if (slot_seconds_out) slot_seconds_out[k] = m->slots[k].seconds;
        return 0;
    } catch (const std::exception &e) {
        set_error("%s", e.what());
        return -1;
    }
}

}  // extern "C"
