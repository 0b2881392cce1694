// Host-side checks of csrc/gmm_model.cpp (model packers, text format, tail packing, the MFCC kernels' mel sweep starts), built by tests/test_host_sanitizers.py with -fsanitize=address,undefined (and once
// with -fsanitize=thread): the model packers (every layout, threaded and not), the text parser on mutated model texts, and
// the printf / strtod-free number conversions against libc.  Test infrastructure: not part of lib/pygmm.so.
#include "gmm_model.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>

namespace sr {
void fail(const char *fmt, ...) {                      // the library's throws sr::Error; any exception type serves here
    char b[256];
    va_list a;
    va_start(a, fmt);
    vsnprintf(b, sizeof b, fmt, a);
    va_end(a);
    throw std::runtime_error(b);
}
}  // namespace sr
using namespace sr;

static std::vector<GMM> make_set(int S, int K, int D, bool shared, unsigned seed) {
    std::mt19937 r(seed);
    std::normal_distribution<double> nd;
    std::vector<GMM> ms(S);
    for (int s = 0; s < S; s++) {
        GMM &g = ms[s];
        g.nr_mixtures = K;
        g.dim = D;
        if (s == 0 || !shared) {
            g.weights.assign(K, 1.0 / K);
            g.mean.resize((size_t)K * D);
            g.sigma.resize((size_t)K * D);
            for (auto &v : g.mean) v = nd(r) * 3;
            for (auto &v : g.sigma) v = 0.5 + std::abs(nd(r));
        } else {
            g = ms[0];
            for (size_t i = 0; i < g.mean.size(); i += 3) g.mean[i] += 0.1 * nd(r);
        }
    }
    return ms;
}

static int check_packers(int S, int K, int D) {
    for (int shared = 0; shared < 2; shared++) {
        auto ms = make_set(S, K, D, shared != 0, 7 + shared);
        ms[S / 2].weights[K / 2] = 0.0;                 // a dead mixture
        std::vector<const GMM *> v;
        for (auto &g : ms) v.push_back(&g);
        auto host = pack_models(v);
        auto b3 = pack_models_split(v, SPLIT_BF16X3);
        auto h2 = pack_models_split(v, SPLIT_F16X2);
        if (host.params.empty() || b3.params.empty() || h2.params.empty()) return 1;
        if (models_share_sigma_and_weights(v) != (shared != 0 && false)) {
            // (the dead mixture breaks weight sharing on purpose: both answers are exercised below)
        }
        ms[S / 2].weights[K / 2] = 1.0 / K;
        if (shared) {
            if (!models_share_sigma_and_weights(v)) return 2;
            auto sh = pack_models_bx3_shared(v);
            auto hs = pack_models_h2_shared(v);
            if (sh.params.empty() || hs.params.empty() || !(hs.amp > 0)) return 3;
        }
    }
    return 0;
}

static int check_parser(int iters) {
    std::mt19937 r(3);
    auto ms = make_set(1, 5, 7, false, 11);
    const std::string base = gmm_format_text(ms[0]);
    GMM back;
    gmm_parse_text(base, back);
    if (gmm_format_text(back) != base) return 1;
    const char junk[] = "0123456789.eE+-xX nanif\n\t,;:pP";
    long parsed = 0;
    for (int it = 0; it < iters; it++) {
        std::string t = base;
        const int nmut = 1 + (int)(r() % 4);
        for (int m = 0; m < nmut; m++) {
            switch (r() % 4) {
            case 0: t.resize(r() % (t.size() + 1)); break;
            case 1: if (!t.empty()) t[r() % t.size()] = junk[r() % (sizeof junk - 1)]; break;
            case 2: if (!t.empty()) t.insert(r() % t.size(), 1, junk[r() % (sizeof junk - 1)]); break;
            default: if (!t.empty()) t.erase(r() % t.size(), 1 + r() % 3);
            }
        }
        try {
            GMM h;
            gmm_parse_text(std::string(t.data(), t.size()), h);
            parsed++;
            if (h.trained() && h.dim <= 64) (void)pack_models({&h});
        } catch (const std::exception &) {
        }
    }
    return parsed > 0 ? 0 : 2;
}

// the text format's numbers against printf("%g") / strtod, through the only doors the file has: format and parse
static int check_numbers(int n) {
    std::mt19937_64 r(7);
    std::uniform_real_distribution<double> u(0, 1);
    GMM g;
    g.nr_mixtures = 1;
    g.dim = n;
    g.weights = {1.0};
    g.mean.resize(n);
    g.sigma.assign(n, 1.0);
    for (int i = 0; i < n; i++) {
        double v;
        switch (i % 5) {
        case 0: v = (u(r) - 0.5) * 20; break;
        case 1: v = std::pow(10.0, (u(r) - 0.5) * 40) * (u(r) < 0.5 ? -1 : 1); break;
        case 2: v = (std::floor(u(r) * 1e6) + 0.5) * std::pow(10.0, (int)(u(r) * 24) - 12); break;   // ties at the 7th digit
        case 3: v = (double)(long)(u(r) * 2000000) * std::pow(10.0, (int)(u(r) * 30) - 15); break;
        default: { uint64_t b = r(); std::memcpy(&v, &b, 8); if (!(std::fabs(v) < 1e300)) v = 1.0; }
        }
        g.mean[i] = v;
    }
    const std::string text = gmm_format_text(g);
    // line 0: "1", line 1: weights, line 2: "dim cov", line 3: means
    size_t pos = 0;
    for (int l = 0; l < 3; l++) pos = text.find('\n', pos) + 1;
    const char *p = text.c_str() + pos;
    for (int i = 0; i < n; i++) {
        char want[64];
        const int len = snprintf(want, sizeof want, "%g", g.mean[i]);
        if (std::strncmp(p, want, (size_t)len) != 0 || p[len] != ' ') {
            fprintf(stderr, "format of %.17g: got '%.24s', libc '%s'\n", g.mean[i], p, want);
            return 1;
        }
        p += len + 1;
    }
    GMM back;
    gmm_parse_text(text, back);
    for (int i = 0; i < n; i++) {
        char w[64];
        snprintf(w, sizeof w, "%g", g.mean[i]);
        const double ref = std::strtod(w, nullptr);
        if (std::memcmp(&ref, &back.mean[i], 8) != 0) {
            fprintf(stderr, "parse of '%s': %.17g, libc %.17g\n", w, back.mean[i], ref);
            return 2;
        }
    }
    return 0;
}

// pack_tail_tiles: every tile exactly once; full tiles alone and in order; packs of <= 4 tiles and <= 32 columns, all behind the full ones
static int check_tail_packing() {
    std::mt19937 rng(5);
    for (int round = 0; round < 200; round++) {
        const int n = (int)(rng() % 400);
        std::vector<int> counts(n);
        for (int &c : counts) c = (rng() % 3 == 0) ? 1 + (int)(rng() % 31) : 32;
        if (round == 0) counts.assign(64, 8);                          // only tails: 16 packs of 4
        if (round == 1) counts.assign(10, 32);                         // no tails
        for (int pack = 0; pack < 2; pack++) {
            const auto items = pack_tail_tiles(counts, 32, pack != 0);
            std::vector<int> seen(counts.size(), 0);
            bool in_packs = false;
            int last_full = -1;
            for (const auto &it : items) {
                int cols = 0, members = 0;
                for (int p = 0; p < 4; p++) {
                    if (it.t[p] < 0) continue;
                    if (p > 0 && it.t[p - 1] < 0) return 1;               // members are packed to the front
                    if (it.t[p] >= (int)counts.size()) return 2;
                    seen[it.t[p]]++;
                    cols += counts[it.t[p]];
                    members++;
                }
                if (members == 0 || cols > 32) return 3;
                const bool is_pack = pack && counts[it.t[0]] < 32;
                if (!is_pack) {
                    if (members != 1 || in_packs) return 4;               // a full tile after a pack, or sharing its wave
                    if (it.t[0] <= last_full) return 5;
                    last_full = it.t[0];
                } else {
                    in_packs = true;
                    for (int p = 0; p < members; p++)
                        if (counts[it.t[p]] >= 32) return 6;
                }
            }
            for (int v : seen)
                if (v != 1) return 7;
            if (!pack && items.size() != counts.size()) return 8;
        }
        if (round == 0 && pack_tail_tiles(counts, 32, true).size() != 16) return 9;
    }
    return 0;
}

// mel_sweep_starts: starts are multiples of 4, never negative, never above the band's first column, the padded run fits the pass
// wherever the unshifted one did, and no group of bands served together conflicts more than with the unshifted starts; on the
// reference's own 16 kHz / 50-filter bank shape (runs growing with the band index) the first three passes come down to 2 extra
// cycles per read (round 5's choice: 4 / 6 / 4).
static int check_mel_starts() {
    std::mt19937 rng(11);
    for (int round = 0; round < 300; round++) {
        const int B = 1 + (int)(rng() % 64);
        int col0[64] = {0}, cnt[64] = {0}, start[64] = {0}, naive[64] = {0}, pass_len[4] = {0, 0, 0, 0};
        int c = (int)(rng() % 8);
        for (int b = 0; b < B; b++) {
            cnt[b] = (rng() % 20 == 0) ? 0 : 1 + (int)(rng() % (round % 3 == 0 ? 12 : 4 + 3 * b));
            col0[b] = c;
            c += 1 + (int)(rng() % (2 + cnt[b]));
            naive[b] = col0[b] & ~3;
            pass_len[b / 16] = std::max(pass_len[b / 16], ((cnt[b] + 15) / 16) * 16);
        }
        mel_sweep_starts(col0, cnt, B, pass_len, start);
        for (int b = 0; b < B; b++) {
            if (cnt[b] == 0) continue;
            if (start[b] < 0 || (start[b] & 3) || start[b] > col0[b]) return 1;
            const bool fitted = col0[b] - naive[b] + cnt[b] <= pass_len[b / 16];
            if (fitted && col0[b] - start[b] + cnt[b] > pass_len[b / 16]) return 2;
            if (!fitted && start[b] != naive[b]) return 3;
        }
        for (int ps = 0; ps < 4; ps++)
            if (mel_sweep_extra_cycles(start, cnt, B, ps) > mel_sweep_extra_cycles(naive, cnt, B, ps)) return 4;
    }
    // the 16 kHz bank of MFCC.py's defaults: first columns and run lengths as tests/golden's melbank has them
    const int col16k[50] = {1, 5, 10, 15, 20, 26, 31, 38, 44, 51, 58, 65, 73, 81, 90, 99, 108, 118, 129, 140, 152, 164, 177, 190, 204, 219, 235, 251, 268,
                            286, 305, 325, 346, 368, 392, 416, 442, 468, 497, 526, 558, 590, 625, 661, 699, 739, 781, 825, 871, 920};
    const int cnt16k[50] = {9, 10, 10, 11, 11, 12, 13, 13, 14, 14, 15, 16, 17, 18, 18, 19, 21, 22, 23, 24, 25, 26, 27, 29, 31, 32, 33, 35, 37, 39, 41, 43,
                            46, 48, 50, 52, 55, 58, 61, 64, 67, 71, 74, 78, 82, 86, 90, 95, 100, 104};
    const int len16k[4] = {32, 48, 96, 112};
    int st[64] = {0};
    mel_sweep_starts(col16k, cnt16k, 50, len16k, st);
    for (int ps = 0; ps < 3; ps++)
        if (mel_sweep_extra_cycles(st, cnt16k, 50, ps) > 2) return 5;
    if (mel_sweep_extra_cycles(st, cnt16k, 50, 3) != 0) return 6;
    for (int b = 0; b < 50; b++)
        if (col16k[b] - st[b] + cnt16k[b] > len16k[b / 16]) return 7;          // the kernels' unrolled sweep lengths (mel_preset_steps) stand
    return 0;
}

int main(int argc, char **argv) {
    const bool big = argc > 1 && std::strcmp(argv[1], "threads") == 0;     // sizes at which the packers go multi-threaded
    int rc = big ? check_packers(150, 1024, 39) : (check_packers(17, 37, 13) | check_packers(31, 64, 39));
    if (rc) return printf("packers: %d\n", rc), 10 + rc;
    if (!big) {
        if ((rc = check_tail_packing())) return printf("tail packing: %d\n", rc), 40 + rc;
        if ((rc = check_mel_starts())) return printf("mel starts: %d\n", rc), 60 + rc;
        if ((rc = check_parser(20000))) return printf("parser: %d\n", rc), 20 + rc;
        if ((rc = check_numbers(300000))) return printf("numbers: %d\n", rc), 30 + rc;
    }
    printf("host checks ok\n");
    return 0;
}
