// ccsx_host.cpp — host-side pieces of the C ABI that need no device: model / option defaults,
// result layout, the deterministic synthetic subread generator.
#include "ccsx.h"
#include "ccsx_internal.h"

int ccsx_kernel_is_experiment();          // ccsx_kernels.hip
const char *ccsx_kernel_build_flags();

#include <atomic>
#include <algorithm>
#include <cmath>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <cstdio>
#include <sched.h>
#include <string>
#include <vector>

static thread_local std::string g_last_error;
void ccsx_set_error(const std::string &s) { g_last_error = s; }

extern "C" {

int ccsx_abi_version(void) { return CCSX_ABI_VERSION; }
// (a library built with timing-only experiment switches — wrong results — reports the NEGATIVE version: no golden-vector or parity check accepts it)
int ccsx_spec_version(void) { return ccsx_kernel_is_experiment() ? -CCSX_SPEC_VERSION : CCSX_SPEC_VERSION; }
const char *ccsx_build_flags(void) { return ccsx_kernel_build_flags(); }
const char *ccsx_last_error(void) { return g_last_error.c_str(); }
// every environment variable the library reads for scheduling / debugging (ccsx_api.cpp, ccsx_kernels.hip ccsx_launch_all): results never depend on them
const char *ccsx_runtime_switches(void)
{
    static const char *names[] = {"CCSX_STAGE_PRIO", "CCSX_POA_SPLIT", "CCSX_SERIAL_STAGES", "CCSX_A16_ONE_REGION", "CCSX_ALIGN16_MAX_SLOTS", "CCSX_TB_ASIDE",
                                  "CCSX_POLISH_MAX_BLOCKS", "CCSX_EPOCH_REBASE_MS", "CCSX_TRACE", "CCSX_NUMA"};
    static thread_local std::string out;
    out.clear();
    for (const char *n : names)
        if (const char *v = std::getenv(n)) { if (!out.empty()) out += ' '; out += n; out += '='; out += v; }
    return out.c_str();
}

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
    o->max_insertion_size = 30;   // docs/how-does-ccs-work.md:74-78
    o->max_qv = 50;               // SPEC v7 "honest QVs"; 93 = the reference's documented range (docs/faq/qv-binning.md:31)
}

// ---- NUMA placement of a device's host threads (docs/faq/parallelize.md:8-29: one node, several GPUs).  Sysfs only: no libnuma in the image.
// The PCI address names the device's NUMA node; binding a thread to that node's CPUs BEFORE it allocates page-locked staging makes the staging node-local by
// first touch (hipHostMalloc populates and pins the pages on the calling thread), so that the H2D copies of eight GPUs do not all cross the socket link.
static bool read_small_file(const std::string &path, std::string &out)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
    std::fclose(f);
    buf[n] = 0; out = buf;
    return true;
}
static const char *sysfs_root() { const char *r = std::getenv("CCSX_SYSFS_ROOT"); return r ? r : ""; }   // (tests point this at a fake tree)

int ccsx_pci_numa_node(const char *pci_bus_id)
{
    if (!pci_bus_id || !*pci_bus_id) return -1;
    std::string id(pci_bus_id), txt;
    for (char &c : id) c = (char)std::tolower((unsigned char)c);
    if (!read_small_file(std::string(sysfs_root()) + "/sys/bus/pci/devices/" + id + "/numa_node", txt)) return -1;
    char *end = nullptr;
    const long v = std::strtol(txt.c_str(), &end, 10);
    return (end == txt.c_str() || v < 0) ? -1 : (int)v;      // (-1 in the file: the platform reports no affinity)
}

// "0-15,64-79" -> CPU set; returns the number of CPUs
static int parse_cpulist(const std::string &s, cpu_set_t &set)
{
    CPU_ZERO(&set);
    int n = 0;
    const char *p = s.c_str();
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
        if (!*p) break;
        char *e = nullptr;
        long a = std::strtol(p, &e, 10), b = a;
        if (e == p) break;
        p = e;
        if (*p == '-') { b = std::strtol(p + 1, &e, 10); if (e == p + 1) break; p = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0) { CPU_SET((int)c, &set); ++n; }
    }
    return n;
}

int ccsx_bind_thread_to_node(int node)
{
    if (node < 0) return -1;
    if (const char *e = std::getenv("CCSX_NUMA")) if (e[0] == '0') return -1;      // switch: leave the threads where the OS puts them
    std::string txt;
    if (!read_small_file(std::string(sysfs_root()) + "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", txt)) return -1;
    cpu_set_t want, have, both;
    if (parse_cpulist(txt, want) <= 0) return -1;
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return -1;
    CPU_AND(&both, &want, &have);                             // never outside what the process may use (cgroup cpusets, taskset)
    if (CPU_COUNT(&both) <= 0) return -1;
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return -1;
    return node;
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

// layout of a ccsx_drafts object: the draft slots are those of the results (capacity layout), a ZMW's window bounds take dcap / 19 + 4 words
// (cores are 19..25 columns: SPEC windows) — the same arithmetic as the engine's own slots (ccsx_api.cpp stage())
void ccsx_draft_layout(const ccsx_batch *b, int64_t *seq_off, int64_t *win_off, int64_t *seq_capacity, int64_t *win_capacity)
{
    int64_t off = 0, woff = 0;
    for (int z = 0; z < b->n_zmw; ++z) {
        if (seq_off) seq_off[z] = off;
        if (win_off) win_off[z] = woff;
        int64_t maxL = 0;
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) {
            const int64_t L = b->base_off[r + 1] - b->base_off[r];
            if (L > maxL) maxL = L;
        }
        const int64_t dcap = ccsx_draft_cap(maxL);
        off += dcap;
        woff += dcap / (CCSX_WIN_CORE - 3) + 4;
    }
    if (seq_off) seq_off[b->n_zmw] = off;
    if (win_off) win_off[b->n_zmw] = woff;
    if (seq_capacity) *seq_capacity = off;
    if (win_capacity) *win_capacity = woff;
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

// one ZMW's arrays; the stream of ZMW id depends only on (seed, id), so ZMWs are generated independently (in parallel)
struct SynthZmw {
    float snr[4];
    std::vector<uint8_t> bases, pw, ipd, flags, tpl;
    std::vector<int64_t> read_end;      // exclusive end of every read within `bases`
};

static void synth_one(int id, int32_t passes_lo, int32_t passes_hi, int32_t len_lo, int32_t len_hi, uint64_t seed, SynthZmw &o)
{
    static const double PM[3] = {0.30, 0.30, 0.40}, PX[3] = {0.60, 0.25, 0.15};
    static const double P_DEL = 0.04, P_SUB = 0.01, P_INS = 0.06;
    Rng g(seed ^ 0xCC5ull ^ ((uint64_t)id * 0x9E3779B97F4A7C15ull));
    const int P = passes_lo + g.below(passes_hi - passes_lo + 1);
    int L = len_lo;
    if (len_hi > len_lo) L = (int)std::floor(std::exp(std::log((double)len_lo) + g.uni() * (std::log((double)len_hi) - std::log((double)len_lo))) + 0.5);
    static const float base_snr[4] = {9.0f, 16.0f, 8.0f, 13.0f};
    for (int c = 0; c < 4; ++c) {
        float s = base_snr[c] * (float)(1.0 + 0.1 * g.gauss());
        o.snr[c] = s < 4.0f ? 4.0f : s;
    }
    std::vector<uint8_t> &t = o.tpl, tr(L);
    t.resize(L);
    for (int j = 0; j < L; ++j) t[j] = (uint8_t)g.below(4);
    for (int j = 0; j < L; ++j) tr[j] = (uint8_t)(3 - t[L - 1 - j]);
    o.bases.reserve((size_t)P * (L + L / 16)); o.pw.reserve(o.bases.capacity()); o.ipd.reserve(o.bases.capacity());
    for (int k = 0; k < P; ++k) {
        const int rev = k & 1;
        const std::vector<uint8_t> &src = rev ? tr : t;
        for (int j = 0; j < L; ++j) {
            while (g.uni() < P_INS) {   // extra base before consuming src[j]: half cognate (branch), half not (stick)
                uint8_t b = src[j];
                if (g.uni() >= 0.5) b = (uint8_t)((b + 1 + g.below(3)) & 3);
                o.bases.push_back(b); o.pw.push_back((uint8_t)draw_pw(g, PX)); o.ipd.push_back((uint8_t)(1 + g.below(60)));
            }
            if (g.uni() < P_DEL) continue;
            uint8_t b = src[j];
            if (g.uni() < P_SUB) b = (uint8_t)((b + 1 + g.below(3)) & 3);
            o.bases.push_back(b); o.pw.push_back((uint8_t)draw_pw(g, PM)); o.ipd.push_back((uint8_t)(1 + g.below(60)));
        }
        o.flags.push_back((uint8_t)rev);
        o.read_end.push_back((int64_t)o.bases.size());
    }
}

int ccsx_synth_generate(int32_t n_zmw, int32_t first_zmw_id, int32_t passes_lo, int32_t passes_hi, int32_t len_lo,
                        int32_t len_hi, uint64_t seed, ccsx_synth **out)
{
    if (n_zmw <= 0 || passes_lo < 1 || passes_hi < passes_lo || len_lo < 1 || len_hi < len_lo) {
        ccsx_set_error("ccsx_synth_generate: bad arguments");
        return -1;
    }
    std::vector<SynthZmw> zs(n_zmw);
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (const char *e = std::getenv("CCSX_SYNTH_THREADS")) nt = (unsigned)std::atoi(e);
        nt = std::max(1u, std::min(nt, 32u));
        if ((int64_t)n_zmw * passes_hi * len_hi < (int64_t)1 << 20) nt = 1;     // small batches: not worth the threads
        std::atomic<int> next(0);
        auto work = [&]() { for (int z; (z = next.fetch_add(1)) < n_zmw;) synth_one(first_zmw_id + z, passes_lo, passes_hi, len_lo, len_hi, seed, zs[z]); };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    int64_t NB = 0, NT = 0; int64_t R = 0;
    for (auto &z : zs) { NB += (int64_t)z.bases.size(); NT += (int64_t)z.tpl.size(); R += (int64_t)z.flags.size(); }
    ccsx_synth *s = (ccsx_synth *)std::calloc(1, sizeof(ccsx_synth));
    auto alloc = [](size_t n) { return std::malloc(n ? n : 1); };
    int32_t *zmw_id = (int32_t *)alloc((size_t)n_zmw * 4), *read_off = (int32_t *)alloc((size_t)(n_zmw + 1) * 4);
    float *snr = (float *)alloc((size_t)n_zmw * 16);
    int64_t *base_off = (int64_t *)alloc((size_t)(R + 1) * 8), *tpl_off = (int64_t *)alloc((size_t)(n_zmw + 1) * 8);
    uint8_t *bases = (uint8_t *)alloc(NB), *pw = (uint8_t *)alloc(NB), *ipd = (uint8_t *)alloc(NB), *flags = (uint8_t *)alloc(R), *tpl = (uint8_t *)alloc(NT);
    int64_t bo = 0, to = 0; int32_t ro = 0;
    base_off[0] = 0; read_off[0] = 0;
    for (int z = 0; z < n_zmw; ++z) {
        SynthZmw &q = zs[z];
        zmw_id[z] = first_zmw_id + z;
        std::memcpy(snr + 4 * (size_t)z, q.snr, 16);
        std::memcpy(bases + bo, q.bases.data(), q.bases.size()); std::memcpy(pw + bo, q.pw.data(), q.pw.size()); std::memcpy(ipd + bo, q.ipd.data(), q.ipd.size());
        for (size_t k = 0; k < q.flags.size(); ++k) { flags[ro] = q.flags[k]; base_off[++ro] = bo + q.read_end[k]; }
        read_off[z + 1] = ro;
        tpl_off[z] = to;
        std::memcpy(tpl + to, q.tpl.data(), q.tpl.size());
        bo += (int64_t)q.bases.size(); to += (int64_t)q.tpl.size();
        SynthZmw().bases.swap(q.bases); SynthZmw().pw.swap(q.pw); SynthZmw().ipd.swap(q.ipd);
    }
    tpl_off[n_zmw] = to;
    s->batch.n_zmw = n_zmw; s->batch.n_reads = ro; s->batch.n_bases = bo;
    s->batch.zmw_id = zmw_id; s->batch.snr = snr; s->batch.read_off = read_off; s->batch.base_off = base_off;
    s->batch.bases = bases; s->batch.pw = pw; s->batch.ipd = ipd; s->batch.flags = flags;
    s->tpl_off = tpl_off; s->tpl = tpl;
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
