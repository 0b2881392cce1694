// gmm_model.cpp -- text model format + parameter packing (host, float64 -> fp32 tables).
#include "gmm_model.hpp"

#include <atomic>
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace sr {
uint64_t next_gmm_uid() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}
}  // namespace sr

namespace sr {

namespace {

// Packing a configs[3]-sized set (1001 models x 2048 mixtures) is ~10^8 log/divide/split steps: the independent pieces
// (models, blocks of 15 models) are shared by a few host threads.  `work(i)` must only write what belongs to piece i.
template <class F>
void host_parallel_for(int n, size_t cost_per_piece, F work) {
    const size_t total = (size_t)n * cost_per_piece;
    const int n_threads = total < ((size_t)1 << 22) ? 1 : std::max(1, std::min({n, (int)std::thread::hardware_concurrency(), 32}));
    auto run = [&](int t) { for (int i = t; i < n; i += n_threads) work(i, t); };
    if (n_threads == 1) return run(0);
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(run, t);
    run(0);
    for (auto &x : th) x.join();
}

// ---- decimal <-> double for the text model format.  A configs[3] enrolment is 1001 models x 160 k numbers: strtod and
// printf would cost more than scoring the rank's shard.  Both directions stay EXACT: the short cuts cover the cases in
// which one correctly rounded IEEE operation gives the correctly rounded result, everything else goes to libc.
const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                        1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// <= 15 significant digits and |decimal exponent| <= 22: mantissa and power of ten are exact doubles, their product or
// quotient is rounded once (Clinger's fast path) -- all that "%g" ever writes.  false = not taken, nothing consumed.
inline bool fast_decimal(const char *p, const char *end, const char *&next, double &out) {
    const char *s = p;
    bool neg = false;
    if (s < end && (*s == '-' || *s == '+')) neg = *s++ == '-';
    uint64_t m = 0;
    int nd = 0, e10 = 0;
    bool any = false;
    for (; s < end && *s >= '0' && *s <= '9'; s++) {
        any = true;
        if (m == 0 && *s == '0') continue;
        if (++nd > 15) return false;
        m = m * 10 + (uint64_t)(*s - '0');
    }
    if (s < end && *s == '.') {
        for (s++; s < end && *s >= '0' && *s <= '9'; s++) {
            any = true;
            e10--;
            if (m == 0 && *s == '0') continue;
            if (++nd > 15) return false;
            m = m * 10 + (uint64_t)(*s - '0');
        }
    }
    if (!any) return false;                                   // nan, inf, garbage: libc decides
    if (s < end && (*s == 'e' || *s == 'E')) {
        const char *t = s + 1;
        bool eneg = false;
        if (t < end && (*t == '-' || *t == '+')) eneg = *t++ == '-';
        if (t < end && *t >= '0' && *t <= '9') {
            int ev = 0;
            for (; t < end && *t >= '0' && *t <= '9'; t++)
                if ((ev = ev * 10 + (*t - '0')) > 9999) return false;
            e10 += eneg ? -ev : ev;
            s = t;
        }
    }
    if (s < end && (*s == '.' || *s == 'x' || *s == 'X' || *s == 'p' || *s == 'P')) return false;   // hex floats and the like
    if (m == 0) e10 = 0;
    if (e10 < -22 || e10 > 22) return false;
    const double v = e10 < 0 ? (double)m / P10[-e10] : (double)m * P10[e10];
    out = neg ? -v : v;
    next = s;
    return true;
}

// printf("%g") of a finite non-zero double, without printf: the 6 significant digits come from ONE correctly rounded
// scaling by a power of ten (exact up to 10^22); a scaled value within 1e-8 of a rounding tie (the scaling's own error is
// < 2e-10) is left to libc.  Returns the length written, 0 = not taken.
inline int fast_g6(char *buf, double v) {
    if (!(v == v) || v == 0.0) return 0;
    const double a = std::fabs(v);
    if (!(a >= 1e-12 && a < 1e17)) return 0;
    int e2;
    std::frexp(a, &e2);
    int x = (int)std::floor((e2 - 1) * 0.30102999566398120);  // floor(log10 a) or one less: the loop settles it
    double scaled = 0.0;
    for (int tries = 0; tries < 3; tries++) {
        const int n = 5 - x;
        scaled = n >= 0 ? a * P10[n] : a / P10[-n];
        if (scaled < 1e5) x--;
        else if (scaled >= 1e6) x++;
        else break;
    }
    if (!(scaled >= 1e5 && scaled < 1e6)) return 0;
    const double fl = std::floor(scaled), fr = scaled - fl;
    if (std::fabs(fr - 0.5) < 1e-8) return 0;
    uint32_t dig = (uint32_t)fl + (fr > 0.5 ? 1u : 0u);
    if (dig == 1000000u) {
        dig = 100000u;
        x++;
    }
    char d[6];
    for (int i = 5; i >= 0; i--) {
        d[i] = (char)('0' + dig % 10);
        dig /= 10;
    }
    int last = 5;                                             // trailing zeros go (no '#' flag)
    while (last > 0 && d[last] == '0') last--;
    char *o = buf;
    if (v < 0) *o++ = '-';
    if (x < -4 || x >= 6) {                                   // scientific
        *o++ = d[0];
        if (last > 0) {
            *o++ = '.';
            for (int i = 1; i <= last; i++) *o++ = d[i];
        }
        *o++ = 'e';
        int ex = x;
        if (ex < 0) {
            *o++ = '-';
            ex = -ex;
        } else {
            *o++ = '+';
        }
        if (ex >= 100) *o++ = (char)('0' + ex / 100);
        *o++ = (char)('0' + ex / 10 % 10);
        *o++ = (char)('0' + ex % 10);
    } else if (x >= 0) {
        for (int i = 0; i <= x; i++) *o++ = d[i];
        if (last > x) {
            *o++ = '.';
            for (int i = x + 1; i <= last; i++) *o++ = d[i];
        }
    } else {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -x - 1; i++) *o++ = '0';
        for (int i = 0; i <= last; i++) *o++ = d[i];
    }
    return (int)(o - buf);
}

struct Tok {
    const char *p;
    const char *end;
    void skip() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool more() {
        skip();
        return p < end;
    }
    double num(const char *what) {
        skip();
        if (p >= end) fail("model text truncated while reading %s", what);
        double v;
        if (fast_decimal(p, end, p, v)) return v;
        char *q = nullptr;
        v = std::strtod(p, &q);
        if (q == p) fail("model text: cannot parse %s near '%.16s'", what, p);
        p = q;
        return v;
    }
};

}  // namespace

// Token-stream parse with the semantics of `in >> x` (gmm.cc:664-682, :125-150).
void gmm_parse_text(const std::string &text, GMM &out) {
    Tok t{text.data(), text.data() + text.size()};
    double kd = t.num("nr_mixtures");
    int K = (int)kd;
    if (K <= 0 || K > (1 << 20)) fail("model text: bad mixture count %g", kd);
    std::vector<double> w(K);
    for (auto &v : w) v = t.num("weight");
    std::vector<double> mean, sigma;
    int dim = 0;
    for (int k = 0; k < K; k++) {
        int d = (int)t.num("dim");
        int cov = (int)t.num("covariance_type");
        if (cov != 1) fail("model text: only diagonal covariance (type 1) exists, got %d", cov);
        if (d <= 0) fail("model text: bad dim %d", d);
        if (k == 0) {
            dim = d;
            mean.resize((size_t)K * dim);
            sigma.resize((size_t)K * dim);
        } else if (d != dim) {
            fail("model text: mixture %d has dim %d, expected %d", k, d, dim);
        }
        for (int i = 0; i < dim; i++) mean[(size_t)k * dim + i] = t.num("mean");
        for (int i = 0; i < dim; i++) sigma[(size_t)k * dim + i] = t.num("sigma");
    }
    out.nr_mixtures = K;
    out.covariance_type = 1;
    out.dim = dim;
    out.weights = std::move(w);
    out.mean = std::move(mean);
    out.sigma = std::move(sigma);
    out.drop_single();
}

static void put(std::string &s, double v) {   // `out << v << ' '` at default precision = "%g "
    char buf[64];
    int n = fast_g6(buf, v);
    if (n == 0) n = snprintf(buf, sizeof buf, "%g", v);
    buf[n++] = ' ';
    s.append(buf, (size_t)n);
}

std::string gmm_format_text(const GMM &g) {
    std::string s;
    s.reserve((size_t)g.nr_mixtures * (g.dim * 2 + 2) * 12 + 64);
    s += std::to_string(g.nr_mixtures) + "\n";
    for (double w : g.weights) put(s, w);
    s += "\n";
    for (int k = 0; k < g.nr_mixtures; k++) {
        s += std::to_string(g.dim) + " " + std::to_string(g.covariance_type) + "\n";
        for (int i = 0; i < g.dim; i++) put(s, g.mean[(size_t)k * g.dim + i]);
        s += "\n";
        for (int i = 0; i < g.dim; i++) put(s, g.sigma[(size_t)k * g.dim + i]);
        s += "\n";
    }
    return s;
}

// 8..64: every engine; 80, 96: the vector-ALU engine with one frame per lane (the matrix-core layouts stop at 64);
// wider rows: whole slices of WIDE_DC dimensions for the D-chunked kernels (gmm_score_wide_kernel, em_stats_wide_kernel) --
// the reference has no limit (gmm.cc:40-51), MAX_DIM only keeps a corrupt model file from asking for terabytes
static const int kDims[] = {8, 13, 16, 24, 26, 32, 34, 39, 40, 48, 56, 64, 80, 96};

int pick_padded_dim(int dim) {
    for (int d : kDims)
        if (d >= dim) return d;
    if (dim <= MAX_DIM) return (dim + WIDE_DC - 1) / WIDE_DC * WIDE_DC;
    fail("feature dim %d > %d", dim, MAX_DIM);
}

PackedModels pack_models(const std::vector<const GMM *> &models) {
    if (models.empty()) fail("empty model set");
    PackedModels pm;
    pm.n_models = (int)models.size();
    pm.dim = models[0]->dim;
    pm.dp = pick_padded_dim(pm.dim);
    const int DP = pm.dp;
    const size_t rec_f4 = (size_t)2 * DP + 1;
    const double LOG2E = 1.4426950408889634073599;
    const double SQRT_2_PI = 2.5066282746310002;  // gmm.cc:22
    pm.center.assign(DP, 0.0f);
    {
        std::vector<double> acc(pm.dim, 0.0);
        size_t cnt = 0;
        for (const GMM *g : models) {
            if (!g->trained() || g->dim != pm.dim) continue;        // reported below
            for (int k = 0; k < g->nr_mixtures; k++)
                for (int d = 0; d < pm.dim; d++) acc[d] += g->mean[(size_t)k * pm.dim + d];
            cnt += (size_t)g->nr_mixtures;
        }
        for (int d = 0; d < pm.dim && cnt; d++) pm.center[d] = (float)(acc[d] / (double)cnt);
    }
    pm.model_chunk_begin.push_back(0);
    std::vector<size_t> base(pm.n_models);
    size_t total_f4 = 0;
    for (int s = 0; s < pm.n_models; s++) {
        const GMM &g = *models[s];
        if (!g.trained()) fail("model %d of the set is untrained/empty", s);
        if (g.dim != pm.dim) fail("model %d has dim %d, set has %d", s, g.dim, pm.dim);
        const int n_rec = (g.nr_mixtures + KB - 1) / KB;
        base[s] = total_f4;
        for (int r0 = 0; r0 < n_rec; r0 += CB) {
            ChunkDesc cd;
            cd.offset_f4 = (uint32_t)(total_f4 + (size_t)r0 * rec_f4);
            cd.n_records = std::min(CB, n_rec - r0);
            cd.model_done = (r0 + CB >= n_rec) ? s : -1;
            cd.pad = 0;
            pm.chunks.push_back(cd);
        }
        pm.model_chunk_begin.push_back((int)pm.chunks.size());
        total_f4 += (size_t)n_rec * rec_f4;
    }
    if (total_f4 >= ((size_t)1 << 32)) fail("model set of %zu parameter records of 16 bytes: chunk offsets are 32-bit", total_f4);
    pm.params.assign(total_f4 * 4, 0.0f);
    std::vector<double> lift(pm.n_models, 0.0);             // per model: max_k sum_d max(0, -ln sigma_kd)
    host_parallel_for(pm.n_models, (size_t)models[0]->nr_mixtures * pm.dim, [&](int s, int) {
        const GMM &g = *models[s];
        const int K = g.nr_mixtures;
        const int n_rec = (K + KB - 1) / KB;
        for (int r = 0; r < n_rec; r++) {
            float *rec = pm.params.data() + (base[s] + (size_t)r * rec_f4) * 4;
            for (int j = 0; j < KB; j++) {
                const int k = r * KB + j;
                float *cslot = rec + (size_t)2 * DP * 4 + j;
                if (k >= K) {
                    *cslot = NEG_BIG;
                    continue;
                }
                // c = log2e * (ln w - sum ln(sqrt(2pi) sigma)); ln of a non-positive weight is
                // treated as -inf -> NEG_BIG (the reference's linear-domain sum just adds 0).
                double c = g.weights[k] > 0 ? std::log(g.weights[k]) : -INFINITY;
                double up = 0.0;
                for (int d = 0; d < pm.dim; d++) {
                    const double sg = g.sigma[(size_t)k * pm.dim + d];
                    up += std::max(0.0, -std::log(sg));
                    const double mu = g.mean[(size_t)k * pm.dim + d] - (double)pm.center[d];
                    const double sc = std::sqrt(LOG2E * 0.5) / sg;
                    // record layout: dim d -> two float4: {s0,m0,s1,m1} {s2,m2,s3,m3}
                    float *pair = rec + (size_t)d * 8 + (size_t)j * 2;
                    pair[0] = (float)sc;
                    pair[1] = (float)(-mu * sc);
                    c -= std::log(SQRT_2_PI * sg);
                }
                c *= LOG2E;
                *cslot = (std::isfinite(c) && c > (double)NEG_BIG) ? (float)c : NEG_BIG;
                // (a degenerate mixture -- sigma 0 or negative: -ln sigma is inf or NaN -- must not lower the band: with an infinite
                // band every frame of the set goes through the exact partial-product path instead of being silently skipped)
                lift[s] = std::isfinite(up) ? std::max(lift[s], up) : std::numeric_limits<double>::infinity();
            }
        }
    });
    int kmax = 1;
    for (const GMM *g : models) kmax = std::max(kmax, g->nr_mixtures);
    pm.flush_band = *std::max_element(lift.begin(), lift.end()) + std::log((double)kmax) + 17.5;
    return pm;
}

static inline uint32_t f32_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return u;
}
static inline float bits_f32(uint32_t u) {
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}
static inline uint32_t bf16_rne_bits(float v) {   // fp32 bit pattern of v rounded to bf16
    const uint32_t u = f32_bits(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

void split_bf16x3(float v, uint16_t out[3]) {
    const uint32_t h = bf16_rne_bits(v);
    const float r1 = v - bits_f32(h);            // exact in fp32
    const uint32_t m = bf16_rne_bits(r1);
    const float r2 = r1 - bits_f32(m);           // exact
    const uint32_t l = bf16_rne_bits(r2);
    out[0] = (uint16_t)(h >> 16);
    out[1] = (uint16_t)(m >> 16);
    out[2] = (uint16_t)(l >> 16);
}

// IEEE binary16 <-> binary32 on the host (round-to-nearest-even, gradual underflow, saturating to
// +-inf like v_cvt_f16_f32); the packers must produce exactly what the device conversion would.
uint16_t f32_to_f16_rne(float v) {
    const uint32_t u = f32_bits(v);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((a > 0x7F800000u) ? 0x200u : 0u));
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);           // rounds to >= 65520 -> inf
    if (a < 0x33000001u) return (uint16_t)sign;                         // <= 2^-25 -> 0 (ties to even)
    int e = (int)(a >> 23) - 127;
    uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;                        // 24 bits
    int shift = (e >= -14) ? 13 : (13 + (-14 - e));                     // bits dropped
    uint32_t q = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out;
    if (e >= -14)
        out = ((uint32_t)(e + 15) << 10) + (q - 0x400u);                // a carry out of q bumps the exponent
    else
        out = q;                                                        // subnormal (q == 0x400 becomes the smallest normal)
    return (uint16_t)(sign | out);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0x1F) return bits_f32(sign | 0x7F800000u | (m << 13));
    if (e == 0) {
        const float v = (float)m * 5.9604644775390625e-08f;            // m * 2^-24
        return sign ? -v : v;
    }
    return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

void split_f16x2(float v, uint16_t out[2]) {
    out[0] = f32_to_f16_rne(v);
    const float r = v - f16_to_f32(out[0]);      // exact in fp32
    out[1] = f32_to_f16_rne(r);
}

PackedSplit pack_models_split(const std::vector<const GMM *> &models, int scheme) {
    PackedSplit pm;
    pm.scheme = scheme;
    pm.parts = scheme == SPLIT_F16X2 ? 2 : 3;
    const int P = pm.parts;
    const int dim = models[0]->dim;
    pm.ks = (dim + 1 + 7) / 8;
    const int KS = pm.ks;
    const size_t tile_u16 = (size_t)KS * P * 64 * 8;
    const double LOG2E = 1.4426950408889634073599;
    const double SQRT_2_PI = 2.5066282746310002;
    const float dead = scheme == SPLIT_F16X2 ? F16_NEG_BIG : NEG_BIG;
    pm.center.assign(dim, 0.0f);
    pm.scale.assign(dim, 1.0f);
    {
        std::vector<double> acc(dim, 0.0), lsum(dim, 0.0), smin(dim, INFINITY), smax(dim, 0.0);
        size_t cnt = 0;
        for (const GMM *g : models) {
            for (int k = 0; k < g->nr_mixtures; k++)
                for (int d = 0; d < dim; d++) {
                    const double sg = g->sigma[(size_t)k * dim + d];
                    acc[d] += g->mean[(size_t)k * dim + d];
                    lsum[d] += std::log2(sg);
                    smin[d] = std::min(smin[d], sg);
                    smax[d] = std::max(smax[d], sg);
                }
            cnt += (size_t)g->nr_mixtures;
        }
        for (int d = 0; d < dim; d++) {
            pm.center[d] = (float)(acc[d] / (double)cnt);
            pm.sigma_ratio = std::max(pm.sigma_ratio, smax[d] / smin[d]);
            if (scheme == SPLIT_F16X2) {
                int e = (int)std::lround(lsum[d] / (double)cnt);          // log2 of the geometric-mean sigma
                e = std::max(-40, std::min(40, e));
                pm.scale[d] = (float)std::ldexp(1.0, -e);
            }
        }
    }
    size_t live = 0, padded = 0;
    pm.model_chunk_begin.push_back(0);
    // slots (ks, hh, 0..7) of contraction step ks: features f0..f3 = 8 ks + 4 hh + 0..3 as  A2 f0, A2 f1, A1 f0, A1 f1, A2 f2,
    // A2 f3, A1 f2, A1 f3  (A2 pairs with x'^2, A1 with x'): a lane of the frame side owns BOTH powers of four features, in the
    // pairs its packed instructions produce (split_prologue.hpp).  The last slot (A1 of d = 8 KS - 1 >= dim) carries C (pairs
    // with the constant 1).
    std::vector<float> sq((size_t)KS * 8), lin((size_t)KS * 8);
    for (size_t s = 0; s < models.size(); s++) {
        const GMM &g = *models[s];
        const int K = g.nr_mixtures;
        const int n_tiles = (K + MT - 1) / MT;
        const size_t base = pm.params.size();
        pm.params.resize(base + (size_t)n_tiles * tile_u16, 0);
        live += (size_t)K;
        padded += (size_t)n_tiles * MT;
        for (int t = 0; t < n_tiles; t++) {
            uint16_t *tile = pm.params.data() + base + (size_t)t * tile_u16;
            for (int i = 0; i < MT; i++) {
                const int k = t * MT + i;
                std::fill(sq.begin(), sq.end(), 0.0f);
                std::fill(lin.begin(), lin.end(), 0.0f);
                float cst_f = dead;
                // fp16 layouts cannot say "minus infinity" (-60000 is as low as a constant goes, and a frame far from every
                // mixture scores BELOW that): a padding mixture (k >= K) or one of weight 0 is a copy of a real mixture
                // pushed 60000 log2 units down -- always 2^-60000 of something that is in the sum already
                const bool f16_dead = scheme == SPLIT_F16X2 && (k >= K || !(g.weights[k < K ? k : K - 1] > 0));
                const int ksrc = f16_dead ? (k < K ? k : K - 1) : k;
                if (k < K || f16_dead) {
                    const int k = ksrc;                              // (shadows: the coefficients of the source mixture)
                    double cst = f16_dead ? 0.0 : g.weights[k] > 0 ? std::log(g.weights[k]) : -INFINITY;
                    double a = 0.0;
                    for (int d = 0; d < dim; d++) {
                        const double sg = g.sigma[(size_t)k * dim + d];
                        const double mu = g.mean[(size_t)k * dim + d] - (double)pm.center[d];
                        const double iv = 1.0 / (sg * sg);
                        const double us = 1.0 / (double)pm.scale[d];      // x' = z / scale: powers of two, exact
                        sq[d] = (float)(-0.5 * LOG2E * iv * us * us);
                        lin[d] = (float)(LOG2E * mu * iv * us);
                        cst -= std::log(SQRT_2_PI * sg) + 0.5 * mu * mu * iv;
                        a += mu * mu * iv;
                        pm.coef_max = std::max(pm.coef_max, std::max((double)std::fabs(sq[d]), (double)std::fabs(lin[d])));
                    }
                    pm.amp = std::max(pm.amp, a);
                    cst *= LOG2E;
                    if (f16_dead)
                        cst_f = (float)std::max(cst + (double)F16_NEG_BIG, -65000.0);
                    else if (std::isfinite(cst) && cst > (double)dead)
                        cst_f = (float)cst;
                    if (cst_f != dead && !f16_dead) pm.coef_max = std::max(pm.coef_max, (double)std::fabs(cst_f));
                }
                lin[(size_t)KS * 8 - 1] = cst_f;
                for (int d = 0; d < KS * 8; d++) {
                    const int ks = d >> 3, hh = (d >> 2) & 1, jp = d & 3;
                    for (int pw = 0; pw < 2; pw++) {                 // 0: A2 (against x'^2), 1: A1 (against x')
                        uint16_t parts[3];
                        if (scheme == SPLIT_F16X2)
                            split_f16x2(pw ? lin[d] : sq[d], parts);
                        else
                            split_bf16x3(pw ? lin[d] : sq[d], parts);
                        const int lane = i + 32 * hh, j = 4 * (jp >> 1) + 2 * pw + (jp & 1);
                        for (int p = 0; p < P; p++)
                            tile[(((size_t)ks * P + p) * 64 + lane) * 8 + j] = parts[p];
                    }
                }
            }
            ChunkDesc cd;
            cd.offset_f4 = (uint32_t)((base + (size_t)t * tile_u16) / 8);
            cd.n_records = 1;
            cd.model_done = (t + 1 == n_tiles) ? (int)s : -1;
            cd.pad = 0;
            pm.chunks.push_back(cd);
        }
        pm.model_chunk_begin.push_back((int)pm.chunks.size());
    }
    pm.pad_waste = 1.0 - (double)live / (double)padded;
    // the kernels index both tables with compile-time slot numbers 0 .. 8 KS - 1 (split_prologue.hpp): slots past dim are padded
    pm.center.resize((size_t)8 * KS, 0.0f);
    pm.scale.resize((size_t)8 * KS, 0.0f);
    return pm;
}

bool models_share_sigma_and_weights(const std::vector<const GMM *> &models) {
    if (models.size() < 2) return false;
    const GMM &a = *models[0];
    for (size_t s = 1; s < models.size(); s++) {
        const GMM &b = *models[s];
        if (b.nr_mixtures != a.nr_mixtures || b.dim != a.dim) return false;
        if (std::memcmp(a.weights.data(), b.weights.data(), sizeof(double) * a.weights.size()) != 0) return false;
        if (std::memcmp(a.sigma.data(), b.sigma.data(), sizeof(double) * a.sigma.size()) != 0) return false;
    }
    return true;
}

// one tile image [ks][part][lane][8]: coefficient of mixture row i at slot c = 16 ks + 8 hh + j
static void put_tile_row(uint16_t *tile, int ksteps, int i, const std::vector<float> &coef) {
    for (int c = 0; c < ksteps * 16; c++) {
        uint16_t parts[3];
        split_bf16x3(coef[c], parts);
        const int ks = c >> 4, hh = (c >> 3) & 1, j = c & 7;
        const int lane = i + 32 * hh;
        for (int p = 0; p < 3; p++) tile[(((size_t)ks * 3 + p) * 64 + lane) * 8 + j] = parts[p];
    }
}

PackedBx3Shared pack_models_bx3_shared(const std::vector<const GMM *> &models) {
    PackedBx3Shared pm;
    const GMM &g0 = *models[0];
    const int dim = g0.dim, K = g0.nr_mixtures;
    const int S = (int)models.size();
    pm.kq = (dim + 15) / 16;
    pm.kl = (dim + 1 + 15) / 16;
    pm.n_tiles = (K + MT - 1) / MT;
    const size_t q_u16 = (size_t)pm.kq * 3 * 64 * 8, l_u16 = (size_t)pm.kl * 3 * 64 * 8;
    const double LOG2E = 1.4426950408889634073599;
    const double SQRT_2_PI = 2.5066282746310002;
    pm.center.assign(dim, 0.0f);
    {
        std::vector<double> acc(dim, 0.0);
        for (const GMM *g : models)
            for (int k = 0; k < K; k++)
                for (int d = 0; d < dim; d++) acc[d] += g->mean[(size_t)k * dim + d];
        for (int d = 0; d < dim; d++) pm.center[d] = (float)(acc[d] / ((double)K * S));
    }
    const int n_blocks = (S + SHARED_SB - 1) / SHARED_SB;
    const size_t block_u16 = (size_t)pm.n_tiles * (q_u16 + (size_t)SHARED_SB * l_u16);
    pm.params.assign((size_t)n_blocks * block_u16, 0);
    // the quadratic tiles are the same in every block
    std::vector<uint16_t> qimg((size_t)pm.n_tiles * q_u16, 0);
    {
        std::vector<float> coef((size_t)pm.kq * 16);
        for (int t = 0; t < pm.n_tiles; t++)
            for (int i = 0; i < MT; i++) {
                const int k = t * MT + i;
                std::fill(coef.begin(), coef.end(), 0.0f);
                if (k < K)
                    for (int d = 0; d < dim; d++) {
                        const double sg = g0.sigma[(size_t)k * dim + d];
                        coef[d] = (float)(-0.5 * LOG2E / (sg * sg));
                    }
                put_tile_row(qimg.data() + (size_t)t * q_u16, pm.kq, i, coef);
            }
    }
    std::vector<float> coef((size_t)pm.kl * 16);
    for (int b = 0; b < n_blocks; b++) {
        uint16_t *bp = pm.params.data() + (size_t)b * block_u16;
        SharedBlock sb;
        sb.offset_u4 = (uint32_t)(((size_t)b * block_u16) / 8);
        sb.first_model = b * SHARED_SB;
        sb.n_models = std::min(SHARED_SB, S - b * SHARED_SB);
        sb.pad = 0;
        pm.blocks.push_back(sb);
        for (int t = 0; t < pm.n_tiles; t++) {
            uint16_t *tp = bp + (size_t)t * (q_u16 + (size_t)SHARED_SB * l_u16);
            std::memcpy(tp, qimg.data() + (size_t)t * q_u16, q_u16 * sizeof(uint16_t));
            for (int si = 0; si < SHARED_SB; si++) {
                uint16_t *lt = tp + q_u16 + (size_t)si * l_u16;
                const int s = b * SHARED_SB + si;
                for (int i = 0; i < MT; i++) {
                    const int k = t * MT + i;
                    std::fill(coef.begin(), coef.end(), 0.0f);
                    float cst_f = NEG_BIG;
                    if (s < S && k < K) {
                        const GMM &g = *models[s];
                        double cst = g.weights[k] > 0 ? std::log(g.weights[k]) : -INFINITY;
                        double a = 0.0;
                        for (int d = 0; d < dim; d++) {
                            const double sg = g.sigma[(size_t)k * dim + d];
                            const double mu = g.mean[(size_t)k * dim + d] - (double)pm.center[d];
                            const double iv = 1.0 / (sg * sg);
                            coef[d] = (float)(LOG2E * mu * iv);
                            cst -= std::log(SQRT_2_PI * sg) + 0.5 * mu * mu * iv;
                            a += mu * mu * iv;
                        }
                        pm.amp = std::max(pm.amp, a);
                        cst *= LOG2E;
                        if (std::isfinite(cst) && cst > (double)NEG_BIG) cst_f = (float)cst;
                    }
                    coef[(size_t)pm.kl * 16 - 1] = cst_f;
                    put_tile_row(lt, pm.kl, i, coef);
                }
            }
        }
    }
    pm.pad_waste = 1.0 - ((double)K * S) / ((double)pm.n_tiles * MT * n_blocks * SHARED_SB);
    return pm;
}

// ---- shared-sigma layout on two fp16 parts (gmm_score_h2_shared.hip) ----
namespace {

// flat image [ks][lane][8]: row i of the tile, slot c = 16 ks + 8 hh + j -> lane i + 32 hh, element j
void put_flat_row(uint16_t *img, int ksteps, int i, const std::vector<uint16_t> &slots) {
    for (int c = 0; c < ksteps * 16; c++) {
        const int ks = c >> 4, hh = (c >> 3) & 1, j = c & 7;
        img[((size_t)ks * 64 + i + 32 * hh) * 8 + j] = slots[c];
    }
}

}  // namespace

std::vector<WorkItem> pack_tail_tiles(const std::vector<int> &counts, int frames_per_tile, bool pack_tails) {
    std::vector<WorkItem> work, packs;
    work.reserve(counts.size());
    int open_n = 4, open_cols = 0;                       // (no pack open)
    for (int t = 0; t < (int)counts.size(); t++) {
        const int c = counts[t];
        if (!pack_tails || c >= frames_per_tile) {
            work.push_back(WorkItem{{t, -1, -1, -1}});
            continue;
        }
        if (open_n == 4 || open_cols + c > frames_per_tile) {
            open_n = 0;
            open_cols = 0;
            packs.push_back(WorkItem{{-1, -1, -1, -1}});
        }
        packs.back().t[open_n++] = t;
        open_cols += c;
    }
    work.insert(work.end(), packs.begin(), packs.end());
    return work;
}

PackedH2Shared pack_models_h2_shared(const std::vector<const GMM *> &models) {
    PackedH2Shared pm;
    const GMM &g0 = *models[0];
    const int dim = g0.dim, K = g0.nr_mixtures;
    const int S = (int)models.size();
    pm.kqf = (3 * dim + 15) / 16;
    pm.klf = (3 * dim + 2 + 15) / 16;
    pm.n_tiles = (K + MT - 1) / MT;
    const int kimg = std::max(pm.kqf, pm.klf);
    const size_t img_u16 = (size_t)kimg * 64 * 8;
    const double LOG2E = 1.4426950408889634073599;
    const double SQRT_2_PI = 2.5066282746310002;
    // the reference model alone, generic two-part layout: fixes center and scale for the whole set
    // from the SET's statistics (pack_models_split over all models would be wasteful for 1000
    // speakers, and sigma is common anyway: take the means of all models for the centre)
    pm.center.assign(dim, 0.0f);
    pm.scale.assign(dim, 1.0f);
    {
        std::vector<double> acc(dim, 0.0), lsum(dim, 0.0), smin(dim, INFINITY), smax(dim, 0.0);
        for (const GMM *g : models)
            for (int k = 0; k < K; k++)
                for (int d = 0; d < dim; d++) acc[d] += g->mean[(size_t)k * dim + d];
        for (int k = 0; k < K; k++)
            for (int d = 0; d < dim; d++) {
                const double sg = g0.sigma[(size_t)k * dim + d];
                lsum[d] += std::log2(sg);
                smin[d] = std::min(smin[d], sg);
                smax[d] = std::max(smax[d], sg);
            }
        for (int d = 0; d < dim; d++) {
            pm.center[d] = (float)(acc[d] / ((double)K * S));
            pm.sigma_ratio = std::max(pm.sigma_ratio, smax[d] / smin[d]);
            int e = (int)std::lround(lsum[d] / (double)K);
            e = std::max(-40, std::min(40, e));
            pm.scale[d] = (float)std::ldexp(1.0, -e);
        }
    }
    // slot tables (frame side)
    pm.q_desc.assign((size_t)pm.kqf * 16, 0);
    pm.l_desc.assign((size_t)pm.klf * 16, 0);
    // slot of part product `part` (0 lo x hi, 1 hi x lo, 2 hi x hi) of dimension d; the constant's two slots
    // (three runs; interleaving the parts per dimension instead changes nothing: profiles/r05_slot_order_ab.txt)
    auto qpos = [&](int part, int d) { return part * dim + d; };
    auto lpos = [&](int part, int d) { return part == 0 ? d : part == 1 ? (dim + 1) + d : (2 * dim + 1) + d; };
    const int c_lo = dim, c_hi = 3 * dim + 1;
    for (int d = 0; d < dim; d++) {
        pm.q_desc[qpos(0, d)] = (uint16_t)(d | (1 << 8));           // lo(A2) x hi(z^2)
        pm.q_desc[qpos(1, d)] = (uint16_t)(d | (2 << 8));           // hi(A2) x lo(z^2)
        pm.q_desc[qpos(2, d)] = (uint16_t)(d | (1 << 8));           // hi(A2) x hi(z^2)
        pm.l_desc[lpos(0, d)] = (uint16_t)(d | (1 << 8));           // lo(A1) x hi(z)
        pm.l_desc[lpos(1, d)] = (uint16_t)(d | (2 << 8));           // hi(A1) x lo(z)
        pm.l_desc[lpos(2, d)] = (uint16_t)(d | (1 << 8));           // hi(A1) x hi(z)
    }
    pm.l_desc[c_lo] = (uint16_t)(3 << 8);                           // lo(C) x 1
    pm.l_desc[c_hi] = (uint16_t)(3 << 8);                           // hi(C) x 1
    const int n_blocks = (S + SHARED_SB - 1) / SHARED_SB;
    const size_t block_u16 = (size_t)pm.n_tiles * (1 + SHARED_SB) * img_u16;
    pm.params.assign((size_t)n_blocks * block_u16, 0);
    // the quadratic images are the same in every block
    std::vector<uint16_t> qimg((size_t)pm.n_tiles * img_u16, 0);
    {
        std::vector<uint16_t> slots((size_t)pm.kqf * 16);
        for (int t = 0; t < pm.n_tiles; t++)
            for (int i = 0; i < MT; i++) {
                const int k = std::min(t * MT + i, K - 1);           // padding mixtures copy the last real one (see the linear images)
                std::fill(slots.begin(), slots.end(), 0);
                {
                    for (int d = 0; d < dim; d++) {
                        const double sg = g0.sigma[(size_t)k * dim + d];
                        const double us = 1.0 / (double)pm.scale[d];
                        const float a2 = (float)(-0.5 * LOG2E / (sg * sg) * us * us);
                        pm.coef_max = std::max(pm.coef_max, (double)std::fabs(a2));
                        uint16_t parts[2];
                        split_f16x2(a2, parts);
                        slots[qpos(0, d)] = parts[1];
                        slots[qpos(1, d)] = parts[0];
                        slots[qpos(2, d)] = parts[0];
                    }
                }
                put_flat_row(qimg.data() + (size_t)t * img_u16, pm.kqf, i, slots);
            }
    }
    for (int b = 0; b < n_blocks; b++) {
        SharedBlock sb;
        sb.offset_u4 = (uint32_t)(((size_t)b * block_u16) / 8);
        sb.first_model = b * SHARED_SB;
        sb.n_models = std::min(SHARED_SB, S - b * SHARED_SB);
        sb.pad = 0;
        pm.blocks.push_back(sb);
    }
    // sigma is common to the set: its logarithms and reciprocals once, not once per model
    std::vector<double> lsg((size_t)K * dim), ivs((size_t)K * dim);
    for (size_t e = 0; e < (size_t)K * dim; e++) {
        const double sg = g0.sigma[e];
        lsg[e] = std::log(SQRT_2_PI * sg);
        ivs[e] = 1.0 / (sg * sg);
    }
    constexpr int MAX_T = 32;
    double amp_t[MAX_T] = {0}, coef_t[MAX_T] = {0};
    host_parallel_for(n_blocks, (size_t)SHARED_SB * K * dim, [&](int b, int thr) {
        std::vector<uint16_t> slots((size_t)pm.klf * 16);
        double amp_l = amp_t[thr], coef_l = coef_t[thr];
        uint16_t *bp = pm.params.data() + (size_t)b * block_u16;
        for (int t = 0; t < pm.n_tiles; t++) {
            uint16_t *tp = bp + (size_t)t * (1 + SHARED_SB) * img_u16;
            std::memcpy(tp, qimg.data() + (size_t)t * img_u16, img_u16 * sizeof(uint16_t));
            for (int si = 0; si < SHARED_SB; si++) {
                uint16_t *lt = tp + (size_t)(1 + si) * img_u16;
                const int s = b * SHARED_SB + si;
                for (int i = 0; i < MT; i++) {
                    const int kk = t * MT + i;
                    std::fill(slots.begin(), slots.end(), 0);
                    float cst_f = F16_NEG_BIG;
                    // padding mixtures and mixtures of weight 0: a copy of a real mixture 60000 log2 units down (a bare
                    // constant of -60000 would OUTSCORE the real mixtures on a frame far from all of them)
                    const int k = std::min(kk, K - 1);
                    const bool f16_dead = s < S && (kk >= K || !(models[s]->weights[k] > 0));
                    if (s < S) {
                        const GMM &g = *models[s];
                        double cst = f16_dead ? 0.0 : std::log(g.weights[k]);
                        double a = 0.0;
                        for (int d = 0; d < dim; d++) {
                            const double mu = g.mean[(size_t)k * dim + d] - (double)pm.center[d];
                            const double iv = ivs[(size_t)k * dim + d];
                            const double us = 1.0 / (double)pm.scale[d];
                            const float a1 = (float)(LOG2E * mu * iv * us);
                            coef_l = std::max(coef_l, (double)std::fabs(a1));
                            uint16_t parts[2];
                            split_f16x2(a1, parts);
                            slots[lpos(0, d)] = parts[1];
                            slots[lpos(1, d)] = parts[0];
                            slots[lpos(2, d)] = parts[0];
                            cst -= lsg[(size_t)k * dim + d] + 0.5 * mu * mu * iv;
                            a += mu * mu * iv;
                        }
                        amp_l = std::max(amp_l, a);
                        cst *= LOG2E;
                        if (f16_dead) {
                            cst_f = (float)std::max(cst + (double)F16_NEG_BIG, -65000.0);
                        } else if (std::isfinite(cst) && cst > (double)F16_NEG_BIG) {
                            cst_f = (float)cst;
                            coef_l = std::max(coef_l, (double)std::fabs(cst_f));
                        }
                    }
                    uint16_t parts[2];
                    split_f16x2(cst_f, parts);
                    slots[c_lo] = parts[1];
                    slots[c_hi] = parts[0];
                    put_flat_row(lt, pm.klf, i, slots);
                }
            }
        }
        amp_t[thr] = amp_l;
        coef_t[thr] = coef_l;
    });
    for (int t = 0; t < MAX_T; t++) {
        pm.amp = std::max(pm.amp, amp_t[t]);
        pm.coef_max = std::max(pm.coef_max, coef_t[t]);
    }
    pm.pad_waste = 1.0 - ((double)K * S) / ((double)pm.n_tiles * MT * n_blocks * SHARED_SB);
    pm.ref = pack_models_split({models[0]}, SPLIT_F16X2);
    return pm;
}

// ---- mel gather starts (see gmm_model.hpp) ----
static const int kMelGroupBands[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

int mel_sweep_extra_cycles(const int *start, const int *cnt, int n_bands, int pass) {
    int extra = 0;
    for (int g = 0; g < 4; g++) {
        int slots[16] = {0}, worst = 1;
        for (int i = 0; i < 4; i++) {
            const int b = 16 * pass + kMelGroupBands[g][i];
            if (b >= n_bands || cnt[b] <= 0) continue;
            for (int q4 = 0; q4 < 4; q4++) worst = std::max(worst, ++slots[((start[b] >> 2) + q4) & 15]);
        }
        extra += worst - 1;
    }
    return extra;
}

void mel_sweep_starts(const int *col0, const int *cnt, int n_bands, const int pass_len[4], int *start) {
    for (int b = 0; b < n_bands; b++) start[b] = col0[b] & ~3;
    for (int ps = 0; ps < 4; ps++) {
        const int Lp = pass_len[ps];
        for (int g = 0; g < 4; g++) {
            int bands[4], nb = 0;
            for (int i = 0; i < 4; i++) {
                const int b = 16 * ps + kMelGroupBands[g][i];
                if (b < n_bands && cnt[b] > 0) bands[nb++] = b;
            }
            if (nb < 2) continue;
            int kmax[4] = {0, 0, 0, 0};
            for (int i = 0; i < nb; i++) {
                const int b = bands[i], st0 = col0[b] & ~3;
                while (kmax[i] < 15 && st0 - 4 * (kmax[i] + 1) >= 0 && col0[b] - (st0 - 4 * (kmax[i] + 1)) + cnt[b] <= Lp) kmax[i]++;
            }
            int best_cost = 1 << 30, best_k[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0};
            for (;;) {
                int slots[16] = {0}, worst = 0, pad = 0;
                for (int i = 0; i < nb; i++) {
                    const int sl = (((col0[bands[i]] & ~3) - 4 * k[i]) >> 2) & 15;
                    for (int q4 = 0; q4 < 4; q4++) worst = std::max(worst, ++slots[(sl + q4) & 15]);
                    pad += k[i];
                }
                const int cost = (worst - 1) * 1024 + pad;
                if (cost < best_cost) {
                    best_cost = cost;
                    for (int i = 0; i < nb; i++) best_k[i] = k[i];
                }
                int i = 0;
                while (i < nb && ++k[i] > kmax[i]) k[i++] = 0;
                if (i == nb) break;
            }
            for (int i = 0; i < nb; i++) start[bands[i]] = (col0[bands[i]] & ~3) - 4 * best_k[i];
        }
    }
}

}  // namespace sr
