// ccsx_host.cpp — host-side pieces of the C ABI that need no device: model / option defaults,
// result layout, the deterministic synthetic subread generator.
#include "ccsx.h"
#include "ccsx_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_last_error;
void ccsx_set_error(const std::string &s) { g_last_error = s; }

extern "C" {

int ccsx_abi_version(void) { return CCSX_ABI_VERSION; }
const char *ccsx_last_error(void) { return g_last_error.c_str(); }

// Synthetic parameter set "SYN-1".  The trained PacBio tables (docs/faq/chemistry.md:27-56,
// $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/*.json) are not in the mount; this set has the documented shape:
// dinucleotide context, pulse-width dependent emissions, SNR dependent transitions
// (docs/how-does-ccs-work.md:90-94) and is matched to the synthetic error channel below.
void ccsx_model_default(ccsx_model *m)
{
    std::memset(m, 0, sizeof(*m));
    std::strncpy(m->name, "SYN-1", sizeof(m->name) - 1);
    m->snr_lo = 4.0f;
    m->snr_hi = 20.0f;
    static const float pw_match[3] = {0.30f, 0.30f, 0.40f};
    static const float pw_extra[3] = {0.60f, 0.25f, 0.15f};
    for (int k = 0; k < CCSX_NCTX; ++k) {
        const int prev = k >> 2, cur = k & 3;
        // mild context dependence (homopolymer contexts slightly more indel-prone; every context distinct)
        const float hp = ((prev == cur) ? 1.10f : 1.0f) * (1.0f + 0.004f * (float)k);
        // branch (cognate extra), stick (non-cognate extra), deletion: w = c0 + c1*snr
        const float c0[3] = {0.050f * hp, 0.036f, 0.058f * hp};
        const float c1[3] = {-0.0012f * hp, -0.0004f, -0.0012f * hp};
        for (int mv = 0; mv < 3; ++mv) {
            m->trans_poly[k][mv][0] = c0[mv];
            m->trans_poly[k][mv][1] = c1[mv];
            m->trans_poly[k][mv][2] = 0.0f;
            m->trans_poly[k][mv][3] = 0.0f;
        }
        for (int o = 0; o < CCSX_NOBS; ++o) {
            const int b = o / 3, pwb = o % 3;
            m->em_match[k][o] = (b == cur ? 0.985f : 0.005f) * pw_match[pwb];
        }
        for (int p = 0; p < 3; ++p) {
            m->em_branch[k][p] = pw_extra[p];
            m->em_stick[k][p] = pw_extra[p];
        }
    }
}

void ccsx_opts_default(ccsx_opts *o)
{
    std::memset(o, 0, sizeof(*o));
    o->max_poa_cov = 5;      // "an approximate draft consensus from a few subreads" (docs/how-does-ccs-work.md:15); DESIGN.md §4 sweep
    o->min_passes = 3;
    o->top_passes = 60;      // docs/faq/accuracy-vs-passes.md:49-52
    o->min_length = 10;
    o->max_length = 50000;
    o->min_rq = 0.99f;       // docs/how-does-ccs-work.md:111, docs/faq/reads-bam.md:38
    o->poa_slots = 0;
    o->min_zscore = -3.4f;   // [RECALL] unanimity's MinZScore; DESIGN.md §2 "z-score gate"
}

int64_t ccsx_result_layout(const ccsx_batch *b, int64_t *seq_off)
{
    int64_t off = 0;
    for (int z = 0; z < b->n_zmw; ++z) {
        seq_off[z] = off;
        int64_t maxL = 0;
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) {
            const int64_t L = b->base_off[r + 1] - b->base_off[r];
            if (L > maxL) maxL = L;
        }
        off += ccsx_draft_cap(maxL);
    }
    seq_off[b->n_zmw] = off;
    return off;
}

// ---------------------------------------------------------------------------------------------
// synthetic generator: splitmix64 -> xoshiro256**, never std::*_distribution (not portable).
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int below(int n) { return (int)(uni() * n); }
    double gauss() { double a = 0; for (int i = 0; i < 12; ++i) a += uni(); return a - 6.0; }
};

static int draw_pw(Rng &g, const double *pmf)
{
    const double u = g.uni();
    return u < pmf[0] ? 1 : (u < pmf[0] + pmf[1] ? 2 : 3);
}

int ccsx_synth_generate(int32_t n_zmw, int32_t first_zmw_id, int32_t passes_lo, int32_t passes_hi, int32_t len_lo,
                        int32_t len_hi, uint64_t seed, ccsx_synth **out)
{
    if (n_zmw <= 0 || passes_lo < 1 || passes_hi < passes_lo || len_lo < 1 || len_hi < len_lo) {
        ccsx_set_error("ccsx_synth_generate: bad arguments");
        return -1;
    }
    static const double PM[3] = {0.30, 0.30, 0.40}, PX[3] = {0.60, 0.25, 0.15};
    static const double P_DEL = 0.04, P_SUB = 0.01, P_INS = 0.06;
    std::vector<int32_t> zmw_id(n_zmw), read_off(n_zmw + 1, 0);
    std::vector<float> snr((size_t)n_zmw * 4);
    std::vector<int64_t> base_off(1, 0), tpl_off(n_zmw + 1, 0);
    std::vector<uint8_t> bases, pw, ipd, flags, tpl;
    for (int z = 0; z < n_zmw; ++z) {
        const int id = first_zmw_id + z;
        zmw_id[z] = id;
        Rng g(seed ^ 0xCC5ull ^ ((uint64_t)id * 0x9E3779B97F4A7C15ull));
        const int P = passes_lo + g.below(passes_hi - passes_lo + 1);
        int L = len_lo;
        if (len_hi > len_lo) L = (int)std::floor(std::exp(std::log((double)len_lo) + g.uni() * (std::log((double)len_hi) - std::log((double)len_lo))) + 0.5);
        static const float base_snr[4] = {9.0f, 16.0f, 8.0f, 13.0f};
        for (int c = 0; c < 4; ++c) {
            float s = base_snr[c] * (float)(1.0 + 0.1 * g.gauss());
            snr[(size_t)z * 4 + c] = s < 4.0f ? 4.0f : s;
        }
        std::vector<uint8_t> t(L), tr(L);
        for (int j = 0; j < L; ++j) t[j] = (uint8_t)g.below(4);
        for (int j = 0; j < L; ++j) tr[j] = (uint8_t)(3 - t[L - 1 - j]);
        tpl_off[z] = (int64_t)tpl.size();
        tpl.insert(tpl.end(), t.begin(), t.end());
        for (int k = 0; k < P; ++k) {
            const int rev = k & 1;
            const std::vector<uint8_t> &src = rev ? tr : t;
            for (int j = 0; j < L; ++j) {
                while (g.uni() < P_INS) {   // extra base before consuming src[j]: half cognate (branch), half not (stick)
                    uint8_t b = src[j];
                    if (g.uni() >= 0.5) b = (uint8_t)((b + 1 + g.below(3)) & 3);
                    bases.push_back(b); pw.push_back((uint8_t)draw_pw(g, PX)); ipd.push_back((uint8_t)(1 + g.below(60)));
                }
                if (g.uni() < P_DEL) continue;
                uint8_t b = src[j];
                if (g.uni() < P_SUB) b = (uint8_t)((b + 1 + g.below(3)) & 3);
                bases.push_back(b); pw.push_back((uint8_t)draw_pw(g, PM)); ipd.push_back((uint8_t)(1 + g.below(60)));
            }
            flags.push_back((uint8_t)rev);
            base_off.push_back((int64_t)bases.size());
        }
        read_off[z + 1] = (int32_t)flags.size();
    }
    tpl_off[n_zmw] = (int64_t)tpl.size();

    auto dup = [](const void *p, size_t n) { void *q = std::malloc(n ? n : 1); std::memcpy(q, p, n); return q; };
    ccsx_synth *s = (ccsx_synth *)std::calloc(1, sizeof(ccsx_synth));
    s->batch.n_zmw = n_zmw;
    s->batch.n_reads = (int32_t)flags.size();
    s->batch.n_bases = (int64_t)bases.size();
    s->batch.zmw_id = (const int32_t *)dup(zmw_id.data(), zmw_id.size() * 4);
    s->batch.snr = (const float *)dup(snr.data(), snr.size() * 4);
    s->batch.read_off = (const int32_t *)dup(read_off.data(), read_off.size() * 4);
    s->batch.base_off = (const int64_t *)dup(base_off.data(), base_off.size() * 8);
    s->batch.bases = (const uint8_t *)dup(bases.data(), bases.size());
    s->batch.pw = (const uint8_t *)dup(pw.data(), pw.size());
    s->batch.ipd = (const uint8_t *)dup(ipd.data(), ipd.size());
    s->batch.flags = (const uint8_t *)dup(flags.data(), flags.size());
    s->tpl_off = (int64_t *)dup(tpl_off.data(), tpl_off.size() * 8);
    s->tpl = (uint8_t *)dup(tpl.data(), tpl.size());
    *out = s;
    return 0;
}

void ccsx_synth_free(ccsx_synth *s)
{
    if (!s) return;
    std::free((void *)s->batch.zmw_id); std::free((void *)s->batch.snr); std::free((void *)s->batch.read_off);
    std::free((void *)s->batch.base_off); std::free((void *)s->batch.bases); std::free((void *)s->batch.pw);
    std::free((void *)s->batch.ipd); std::free((void *)s->batch.flags); std::free(s->tpl_off); std::free(s->tpl);
    std::free(s);
}

}  // extern "C"
