// ccs — `ccs [options] IN.subreads.bam OUT.bam` on MI355X (reference CLI: docs/index.md:52-64,
// docs/faq/parallelize.md:8-20; flag spellings from SURVEY.md App. C).
//
// Pipeline (the reference's docs/img/ccs-impl.png with the GPU consumers only):
//   reader thread   BGZF inflate on the -j pool -> subread records -> ZMW grouping -> step-1 filters
//                   (docs/how-does-ccs-work.md:19-32) -> SoA batches of --batch-size ZMWs
//   pack workers    SoA packing of a batch straight into page-locked staging (off the reader thread)
//   GPU workers     one per device, each owns ONE ccsx_handle and keeps up to three batches in flight through
//                   ccsx_submit / ccsx_wait: upload, kernels and download of consecutive batches overlap
//   writer          restores input order, writes hifi BAM records with tags rq np ec sn zm RG
//                   (docs/faq/bam-output.md:9-30), BGZF deflate on the pool, ccs_report.txt
//                   (docs/faq/reports-aux-files.md:16-72)
// There is no CPU consensus fallback: without a gfx950 device the program exits with an error.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "bam_io.h"
#include "ccsx.h"

using namespace bamio;

namespace {

// default cost budget of a batch per ZMW of --batch-size (round 4 used 400 kb: 8192 mixed ZMWs became TWO tickets, nothing overlapped, 2.1 k ZMWs/s)
constexpr long long kDefaultBasesPerZmw = 110000;

// ZMWs of the k-th ticket when --batch-size is not given.  The engine wants LARGE batches (one wave per POA graph: a draft stage of 1000 graphs takes nearly as
// long as one of 8000, the kernels being latency-bound), a short run wants several tickets in flight from the start: 1024, 1024, 2048, 2048, 4096, 4096, 8192, ...
// — a file of 8192 ZMWs still becomes five tickets, a long run works in tickets of 8192.
inline int auto_batch_zmws(long long k) { return k >= 6 ? 8192 : 1024 << (int)(k / 2); }

struct Options {
    std::string in, out, report;
    int threads = 0;
    double min_snr = 2.5;
    int batch = 0;                 // --batch-size; 0 = automatic: tickets ramp up from 1024 to 8192 ZMWs (see batch_limit)
    long long batch_bases = 0;     // cost budget of a batch in subread bases (0 = 110 kb x --batch-size): SURVEY.md 8e "batches of ~ constant cost"
    int chunk_i = 1, chunk_n = 1;
    int workers_per_gpu = 3;      // packing threads per device (one engine handle per device keeps three batches in flight)
    std::string model_file;       // --model-file: Arrow parameter json (else: chemistry of the BAM header -> bundle dir / built-in)
    std::vector<int> gpus;
    ccsx_opts o;
    bool all_gpus = false;
    std::string write_synth;      // "N,P,L[,seed]" -> write a synthetic subreads.bam to `out`
    bool dump = false;            // print one line per ZMW after the step-1 filters, no GPU
    bool host_only = false;       // reader -> filters -> packing into staging, no engine and no output: what the host side alone sustains
    bool by_strand = false;       // --by-strand: one consensus per strand (docs/faq/mode-by-strand.md:8-23)
    bool no_partial = false;      // --no-partial-passes: drop the subreads that are not flanked by adapters on both sides (as round 2 did)
    bool qv_binning = false;      // --qv-binning: 7-bin per-base QVs after rq is computed (docs/faq/qv-binning.md:19-31)
    bool suppress_reports = false; // --suppress-reports: no report / metrics files "per default" — a file that is NAMED is still written (docs/faq/sqiie.md:36-46)
    bool report_named = false, metrics_named = false;
    std::string metrics;          // --metrics-json (default <prefix>.zmw_metrics.json.gz)
    std::string report_json;      // --report-json: the ccs_report counts as JSON (docs/changelog.md:72, docs/faq/sqiie.md:42); written only when named
    std::string hifi_summary;     // --hifi-summary-json: the HiFi statistics block as JSON (docs/faq/sqiie.md:45); written only when named
    std::string log_file;         // --log-file: log lines go there instead of stderr (docs/faq/sqiie.md:40)
    double refresh_rate = 5.0;    // --refresh-rate: seconds between progress lines at --log-level INFO (docs/faq/reports-aux-files.md:176-177)
    int log_level = 1;
};

enum HostStatus { HS_OK = 0, HS_POOR_SNR = 100, HS_NO_SUBREADS = 101, HS_TOO_FEW = 102, HS_TOO_LONG = 103 };

struct ZmwIn {
    int32_t zm = 0;
    float snr[4] = {0, 0, 0, 0};
    std::vector<Subread> reads;   // after filters: full-length passes only
    int host_status = HS_OK;
    int64_t order = 0;
    int strand_tag = 0;           // --by-strand: 1 = /fwd, 2 = /rev
    int64_t polymerase_len = 0;   // zmw_metrics: bases of all subreads of the ZMW
    int32_t median_len = 0, n_full = 0;
};

// page-locked staging of one GPU worker (bases / pw / ipd of the batch in flight), grown geometrically and reused
static bool g_plain_arenas = false;        // --host-only: ordinary memory (there may be no device to pin for)
struct Arena {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool plain = false;
    void release() { if (plain) std::free(p); else ccsx_free_pinned(p); p = nullptr; }
    uint8_t *reserve(size_t bytes)
    {
        if (bytes > cap) {
            release();
            cap = bytes + bytes / 4 + (1u << 20);
            plain = g_plain_arenas;
            p = plain ? (uint8_t *)std::malloc(cap) : (uint8_t *)ccsx_alloc_pinned(cap);
            if (!p) { cap = 0; throw std::runtime_error(plain ? std::string("staging: out of memory") : std::string("pinned staging: ") + ccsx_last_error()); }
        }
        return p;
    }
    ~Arena() { release(); }
};

struct Batch {
    int64_t index = 0;
    std::vector<ZmwIn> zmws;      // including host-failed ones (they are only counted)
    // SoA of the ZMWs with host_status == OK
    std::vector<int32_t> zmw_id, read_off;
    std::vector<float> snr;
    std::vector<int64_t> base_off;
    std::vector<uint8_t> flags;
    const uint8_t *bases = nullptr, *pw = nullptr, *ipd = nullptr;   // in the packing worker's pinned arena, valid during its engine call
    int64_t n_bases = 0;
    std::vector<int> slot;        // per zmws[]: index into the SoA or -1
    // results: views into the batch's page-locked output arena (asynchronous D2H), valid until the writer has emitted the batch
    std::vector<int64_t> seq_off;
    int32_t n = 0; int64_t cap = 0;
    int32_t *status = nullptr, *seq_len = nullptr, *np = nullptr, *iters = nullptr, *n_windows = nullptr, *fn = nullptr, *rn = nullptr;
    uint8_t *seq = nullptr, *qual = nullptr, *kin = nullptr;   // kin: 4 planes (fi, fp, ri, rp) of cap bytes each (--hifi-kinetics)
    float *rq = nullptr, *ec = nullptr;
    int dev = 0;                  // which device's packers / worker / staging pools handle the batch (index into the handle list)
    bool have_results = false;    // false: packing or the engine failed for this batch
    std::unique_ptr<struct Arena> in_arena, out_arena;
    ccsx_batch cb; ccsx_results cr; // the structs handed to ccsx_submit live as long as the ticket
};

template <class T> class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T v)
    {
        std::unique_lock<std::mutex> l(m_);
        cv_full_.wait(l, [this] { return q_.size() < cap_; });
        q_.push(std::move(v));
        cv_empty_.notify_one();
    }
    bool pop(T &v)
    {
        std::unique_lock<std::mutex> l(m_);
        cv_empty_.wait(l, [this] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front()); q_.pop();
        cv_full_.notify_one();
        return true;
    }
    void close() { std::lock_guard<std::mutex> l(m_); closed_ = true; cv_empty_.notify_all(); }

private:
    std::queue<T> q_;
    size_t cap_;
    bool closed_ = false;
    std::mutex m_;
    std::condition_variable cv_full_, cv_empty_;
};

void usage()
{
    std::fprintf(stderr,
                 "ccs (MI355X) - generate HiFi reads from PacBio subreads\n"
                 "usage: ccs [options] IN.subreads.bam OUT.{bam,fastq.gz}\n"
                 "  -j, --num-threads N       host threads for BAM (de)compression [all usable: affinity / cgroup quota]\n"
                 "      --min-passes N        minimum full-length passes [3]\n"
                 "      --top-passes N        use at most the N passes closest to the median length [60]; 0 = unlimited (the engine takes up\n"
                 "                            to 255 passes per ZMW)\n"
                 "      --model-file F        Arrow model parameters (json); default: chosen by the chemistry in the BAM header from\n"
                 "                            $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/*.json, then the built-in set\n"
                 "      --disable-heuristics  polish every position (no candidate filter)\n"
                 "      --max-insertion-size N  trim subread insertions longer than N bases relative to the draft [30]; 0 = never\n"
                 "      --min-snr F           minimum SNR of a ZMW [2.5]\n"
                 "      --min-length N        minimum draft length [10]\n"
                 "      --max-length N        maximum draft length [50000]\n"
                 "      --min-rq F            minimum predicted accuracy [0.99]\n"
                 "      --maxPoaCoverage N    subreads used for the draft [5]\n"
                 "      --by-strand           one consensus per strand, read names end in /fwd or /rev\n"
                 "      --no-partial-passes   do not use the first / last (one-adapter) subread of a ZMW in the polish\n"
                 "      --qv-binning          write 7-bin per-base QVs (Q3 Q10 Q17 Q22 Q27 Q35 Q40)\n"
                 "      --hifi-kinetics       averaged per-strand kinetics: tags fi fp fn ri rp rn (ip pw with --by-strand)\n"
                 "      --metrics-json F      per-ZMW metrics [<OUT prefix>.zmw_metrics.json.gz]\n"
                 "      --suppress-reports    do not write the default ccs_report.txt / zmw_metrics.json.gz (files named explicitly are still written)\n"
                 "      --chunk i/N           process only the i-th of N ZMW chunks\n"
                 "      --batch-size N        ZMWs per GPU batch, at most [automatic: 1024, 1024, 2048, 2048, 4096, 4096, then 8192 per ticket]\n"
                 "      --batch-bases N       ... and at most N subread bases (estimated cost, SURVEY.md 8e: batches of about constant cost for mixed\n"
                 "                            pass counts / insert lengths) [110000 x batch-size: a 10-pass x 10 kb ZMW costs ~ 102 kb, so such a batch still\n"
                 "                            closes by count, while a Sequel-II-like mix (~ 200 kb per ZMW, 400 x spread) is cut into >= 6 tickets per 8192 ZMWs]\n"
                 "      --gpus a,b,..         device ordinals [0] ('all' = every visible device)\n"
                 "      --workers-per-gpu N   packing threads per device [3] (one engine handle per device, three batches in flight)\n"
                 "      --report-file F       ccs_report.txt path [<OUT prefix>.ccs_report.txt]\n"
                 "      --report-json F       the same counts as JSON (only when named)\n"
                 "      --hifi-summary-json F HiFi yield / length / quality statistics as JSON (only when named)\n"
                 "      --log-level L         ERROR|WARN|INFO [WARN]\n"
                 "      --log-file F          write log lines to F instead of stderr\n"
                 "      --refresh-rate S      seconds between progress lines at --log-level INFO [5]\n"
                 "      --max-qv Q            largest per-base QV reported; rq follows [50 = this build's calibrated cap; 93 = the reference's range]\n"
                 "  test helpers (not in the reference):\n"
                 "      --write-synthetic N,P,L[,seed]  write a synthetic subreads.bam to OUT (no IN)\n"
                 "      --dump-zmws                     list ZMWs after the step-1 filters (no GPU, no OUT)\n"
                 "      --host-only                     read, filter and pack IN into batch staging, no engine (no GPU, no OUT): host throughput\n");
}

bool parse(int argc, char **argv, Options &o)
{
    ccsx_opts_default(&o.o);
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *n) -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", n); std::exit(2); } return argv[++i]; };
        if (a == "-h" || a == "--help") { usage(); std::exit(0); }
        else if (a == "--version") { std::printf("ccs (ccs_amd, MI355X-native consensus path) abi %d spec %d\n", ccsx_abi_version(), ccsx_spec_version()); std::exit(0); }
        else if (a == "-j" || a == "--num-threads") o.threads = std::atoi(need("-j").c_str());
        else if (a == "--min-passes") o.o.min_passes = std::atoi(need(a.c_str()).c_str());
        else if (a == "--top-passes") o.o.top_passes = std::atoi(need(a.c_str()).c_str());
        else if (a == "--min-snr") o.min_snr = std::atof(need(a.c_str()).c_str());
        else if (a == "--min-length") o.o.min_length = std::atoi(need(a.c_str()).c_str());
        else if (a == "--max-length") o.o.max_length = std::atoi(need(a.c_str()).c_str());
        else if (a == "--min-rq") o.o.min_rq = (float)std::atof(need(a.c_str()).c_str());
        else if (a == "--maxPoaCoverage") o.o.max_poa_cov = std::atoi(need(a.c_str()).c_str());
        else if (a == "--max-insertion-size") { const int v = std::atoi(need(a.c_str()).c_str()); o.o.max_insertion_size = v > 0 ? v : -1; }
        else if (a == "--max-qv") { const int v = std::atoi(need(a.c_str()).c_str()); o.o.max_qv = v < 1 ? 50 : (v > 93 ? 93 : v); }
        else if (a == "--batch-size") o.batch = std::atoi(need(a.c_str()).c_str());
        else if (a == "--batch-bases") o.batch_bases = std::atoll(need(a.c_str()).c_str());
        else if (a == "--report-file") { o.report = need(a.c_str()); o.report_named = true; }
        else if (a == "--workers-per-gpu") o.workers_per_gpu = std::max(1, std::atoi(need(a.c_str()).c_str()));
        else if (a == "--chunk") { if (std::sscanf(need(a.c_str()).c_str(), "%d/%d", &o.chunk_i, &o.chunk_n) != 2 || o.chunk_i < 1 || o.chunk_i > o.chunk_n) { std::fprintf(stderr, "bad --chunk\n"); return false; } }
        else if (a == "--gpus") { std::string v = need(a.c_str()); if (v == "all") o.all_gpus = true; else { size_t p = 0; while (p < v.size()) { o.gpus.push_back(std::atoi(v.c_str() + p)); p = v.find(',', p); if (p == std::string::npos) break; ++p; } } }
        else if (a == "--log-level") { std::string v = need(a.c_str()); o.log_level = v == "INFO" ? 2 : (v == "ERROR" ? 0 : 1); }
        else if (a == "--write-synthetic") o.write_synth = need(a.c_str());
        else if (a == "--dump-zmws") o.dump = true;
        else if (a == "--host-only") o.host_only = true;
        else if (a == "--by-strand") o.by_strand = true;
        else if (a == "--no-partial-passes") o.no_partial = true;
        else if (a == "--qv-binning") o.qv_binning = true;
        else if (a == "--hifi-kinetics") o.o.hifi_kinetics = 1;
        else if (a == "--suppress-reports") o.suppress_reports = true;
        else if (a == "--model-file") o.model_file = need(a.c_str());
        else if (a == "--disable-heuristics") o.o.disable_heuristics = 1;
        else if (a == "--metrics-json") { o.metrics = need(a.c_str()); o.metrics_named = true; }
        else if (a == "--report-json") o.report_json = need(a.c_str());
        else if (a == "--hifi-summary-json") o.hifi_summary = need(a.c_str());
        else if (a == "--log-file") o.log_file = need(a.c_str());
        else if (a == "--refresh-rate") { o.refresh_rate = std::atof(need(a.c_str()).c_str()); if (!(o.refresh_rate >= 0.0)) o.refresh_rate = 0.0; }
        else if (a == "--all" || a == "--all-kinetics" || a == "--subread-fallback" || a == "--split-heteroduplexes" || a == "--hd-finder" || a == "--streamed") {
            // modes of the reference outside this path (OUT OF SCOPE rows): refused by name rather than as a typo
            std::fprintf(stderr, "ccs: %s is not supported by this build (the MI355X consensus path covers the default HiFi mode, --by-strand and --hifi-kinetics)\n", a.c_str());
            std::exit(2);
        }
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return false; }
        else pos.push_back(a);
    }
    // --top-passes 0 = unlimited (docs/faq/accuracy-vs-passes.md:49-52); the engine takes up to CCSX_MAX_PASSES = 255 per ZMW (SPEC v5)
    if (o.o.top_passes <= 0) o.o.top_passes = CCSX_MAX_PASSES;
    if (o.o.top_passes > CCSX_MAX_PASSES) {
        std::fprintf(stderr, "ccs: warning: --top-passes %d: this engine uses at most %d passes per ZMW (those closest to the median length)\n", o.o.top_passes, CCSX_MAX_PASSES);
        o.o.top_passes = CCSX_MAX_PASSES;
    }
    if (!o.write_synth.empty()) { if (pos.size() != 1) return false; o.out = pos[0]; return true; }
    if (o.dump || o.host_only) { if (pos.size() != 1) return false; o.in = pos[0]; if (o.batch < 0) o.batch = 0; return true; }
    if (pos.size() != 2) return false;
    o.in = pos[0]; o.out = pos[1];
    {
        std::string p = o.out; size_t d = p.rfind(".bam"); if (d == std::string::npos) d = p.rfind(".fastq.gz"); if (d != std::string::npos) p = p.substr(0, d);
        if (o.report.empty()) o.report = p + ".ccs_report.txt";
        if (o.metrics.empty()) o.metrics = p + ".zmw_metrics.json.gz";
    }
    if (o.batch < 0) o.batch = 0;
    return true;
}

// host threads the process may actually use: affinity mask and cgroup CPU quota (a container can expose 256 logical CPUs
// with a 16-CPU quota; over-subscribing the quota makes the BGZF pool slower, not faster)
int effective_cores()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    auto quota = [](const char *path, bool v2) -> double {
        FILE *f = std::fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = {0}; long long per = 100000, q = -1;
        double r = 0.0;
        if (v2) { if (std::fscanf(f, "%63s %lld", a, &per) >= 1 && std::strcmp(a, "max") != 0) r = std::atof(a) / (double)(per > 0 ? per : 100000); }
        else if (std::fscanf(f, "%lld", &q) == 1 && q > 0) {
            FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (g) { if (std::fscanf(g, "%lld", &per) != 1) per = 100000; std::fclose(g); }
            r = (double)q / (double)(per > 0 ? per : 100000);
        }
        std::fclose(f);
        return r;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0.0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    if (q > 0.0) { const int c = (int)std::ceil(q); if (c < n) n = c; }
    return n < 1 ? 1 : n;
}

std::string movie_of(const std::string &qname) { const size_t p = qname.find('/'); return p == std::string::npos ? qname : qname.substr(0, p); }

// ---- synthetic subreads.bam (test helper) ----------------------------------------------------------
int write_synthetic(const Options &o, ThreadPool &pool)
{
    int n = 0, P = 0, P2 = 0, L = 0, L2 = 0, partial = 0; unsigned long long seed = 1;   // n,passes,length[,seed[,1]]: a fifth field 1 = the first and the last
    // subread of every ZMW are PARTIAL passes (one adapter only, truncated at the polymerase read's start / end); passes and length may be
    // ranges LO-HI (uniform / log-uniform: the BASELINE configs[4] mix is 3-50,1000-25000)
    {
        std::vector<std::string> f;
        size_t a = 0;
        for (;;) { const size_t c = o.write_synth.find(',', a); f.push_back(o.write_synth.substr(a, c == std::string::npos ? c : c - a)); if (c == std::string::npos) break; a = c + 1; }
        auto span = [](const std::string &t, int &lo, int &hi) { const size_t d = t.find('-'); lo = std::atoi(t.c_str()); hi = d == std::string::npos ? lo : std::atoi(t.c_str() + d + 1); };
        if (f.size() < 3) { std::fprintf(stderr, "bad --write-synthetic\n"); return 2; }
        n = std::atoi(f[0].c_str()); span(f[1], P, P2); span(f[2], L, L2);
        if (f.size() > 3) seed = std::strtoull(f[3].c_str(), nullptr, 10);
        if (f.size() > 4) partial = std::atoi(f[4].c_str());
        if (n < 1 || P < 1 || P2 < P || L < 1 || L2 < L) { std::fprintf(stderr, "bad --write-synthetic\n"); return 2; }
    }
    ccsx_synth *s = nullptr;
    if (ccsx_synth_generate(n, 1000, P, P2, L, L2, seed, &s)) { std::fprintf(stderr, "%s\n", ccsx_last_error()); return 1; }
    BgzfWriter out(o.out, pool);
    const std::string movie = "m64000_synth";
    write_header(out, "@HD\tVN:1.6\tSO:unknown\tpb:5.0.0\n@RG\tID:synth001\tPL:PACBIO\tDS:READTYPE=SUBREAD;Ipd:CodecV1=ip;PulseWidth:CodecV1=pw;"
                      "BINDINGKIT=101-789-500;SEQUENCINGKIT=101-826-100;BASECALLERVERSION=5.0.0;FRAMERATEHZ=100.000000\tPU:" + movie + "\tPM:SEQUELII\n");
    const ccsx_batch &b = s->batch;
    RecordBuilder rb;
    PbiIndex pbi; std::vector<uint64_t> marks;
    for (int z = 0; z < b.n_zmw; ++z) {
        int64_t q = 0;
        for (int r = b.read_off[z]; r < b.read_off[z + 1]; ++r) {
            int64_t a = b.base_off[r], len = b.base_off[r + 1] - a;
            int cxa = 3;                                                          // ADAPTER_BEFORE | ADAPTER_AFTER
            if (partial && b.read_off[z + 1] - b.read_off[z] >= 3) {
                if (r == b.read_off[z]) { a += len / 2; len -= len / 2; cxa = 2; }               // sequencing started inside the insert
                else if (r == b.read_off[z + 1] - 1) { len = (6 * len) / 10; cxa = 1; }          // ... and ended inside it
            }
            const std::string name = movie + "/" + std::to_string(b.zmw_id[z]) + "/" + std::to_string(q) + "_" + std::to_string(q + len);
            rb.begin(name, b.bases + a, nullptr, (uint32_t)len);
            rb.tagZ("RG", "synth001");
            rb.tagi("zm", b.zmw_id[z]);
            rb.tagi("qs", (int32_t)q); rb.tagi("qe", (int32_t)(q + len));
            rb.tagf("rq", 0.8f);
            rb.tagBf("sn", b.snr + 4 * z, 4);
            rb.tagBC("ip", b.ipd + a, (uint32_t)len);
            rb.tagBC("pw", b.pw + a, (uint32_t)len);
            rb.tagC("cx", (uint8_t)(cxa | ((b.flags[r] & 1) ? 32 : 16)));    // adapters + FORWARD/REVERSE_PASS
            marks.push_back(out.mark());
            pbi.rg_id.push_back(0); pbi.q_start.push_back((int32_t)q); pbi.q_end.push_back((int32_t)(q + len)); pbi.hole.push_back(b.zmw_id[z]);
            pbi.read_qual.push_back(0.8f); pbi.ctxt.push_back((uint8_t)cxa);
            rb.finish(out);
            q += len + 45;
        }
    }
    out.close();
    for (uint64_t m : marks) pbi.file_offset.push_back((int64_t)out.virtual_offset(m));
    write_pbi(o.out + ".pbi", pool, pbi);
    ccsx_synth_free(s);
    return 0;
}

// ---- step-1 filters (docs/how-does-ccs-work.md:19-32) ----------------------------------------------
void finish_zmw(ZmwIn &z, const Options &o)
{
    for (auto &r : z.reads) z.polymerase_len += (int64_t)r.size() + 45;   // + adapter between consecutive subreads
    if (z.reads.empty()) { z.host_status = HS_NO_SUBREADS; return; }
    {
        std::vector<size_t> s0;
        for (auto &r : z.reads) s0.push_back(r.size());
        std::nth_element(s0.begin(), s0.begin() + s0.size() / 2, s0.end());
        z.median_len = (int32_t)s0[s0.size() / 2];
        for (auto &r : z.reads) z.n_full += ((r.cx < 0) || ((r.cx & 3) == 3)) ? 1 : 0;
    }
    float mn = z.snr[0];
    for (int c = 1; c < 4; ++c) mn = std::min(mn, z.snr[c]);
    if (mn < (float)o.min_snr) { z.host_status = HS_POOR_SNR; z.reads.clear(); return; }
    std::vector<size_t> lens;
    for (auto &r : z.reads) lens.push_back(r.size());
    std::vector<size_t> s = lens;
    std::nth_element(s.begin(), s.begin() + s.size() / 2, s.end());
    const double med = (double)s[s.size() / 2];
    std::vector<Subread> keep, part;
    bool any_len_ok = false;
    for (auto &r : z.reads) {
        const double l = (double)r.size();
        const bool full = (r.cx < 0) || ((r.cx & 3) == 3);       // flanked by adapters (docs/faq/accuracy-vs-passes.md:17-18)
        // partial passes (one adapter only: the first / last subread of the polymerase read) are not passes, but the polish uses them
        // where they reach (docs/faq/accuracy-vs-passes.md:26-29: ec ~ np + 1); they are shorter by nature: no lower length bound
        const bool partial = !full && r.cx >= 0 && (r.cx & 3) != 0 && l >= 50.0 && l <= 2.0 * med && !o.no_partial;
        if (partial && !r.has_n) {
            // a subread inside [0.5, 2] x median is not a "Median length filter" case whatever its adapters (docs/faq/reports-aux-files.md:26-27:
            // that category means ALL subreads are outside); a ZMW with one-adapter subreads only ends as "Lacking full passes" 
            if (l >= 0.5 * med) any_len_ok = true;
            // the engine's subread limit and the --max-length margin hold for partial passes too: an over-long one is dropped, it never
            // fails the ZMW or the batch (a > 65535-base partial pass used to make ccsx_submit refuse the whole batch)
            if (r.size() > 65535 || l > 1.3 * (double)o.o.max_length + 1000.0) continue;
            r.partial = (uint8_t)((r.cx & 3) == 2 ? 6 : 2);      // cx ADAPTER_AFTER only: the adapter is at the pass's end
            part.push_back(std::move(r));
            continue;
        }
        if (l < 0.5 * med || l > 2.0 * med) continue;            // length filter
        any_len_ok = true;
        if (!full || r.has_n || r.size() == 0) continue;
        keep.push_back(std::move(r));
    }
    z.reads.swap(keep);
    if (!any_len_ok) { z.host_status = HS_NO_SUBREADS; z.reads.clear(); return; }
    // the engine handles subreads up to 65535 bases; longer inserts cannot pass --max-length (<= 50000) anyway
    for (auto &r : z.reads) if (r.size() > 65535 || (double)r.size() > 1.3 * (double)o.o.max_length + 1000.0) { z.host_status = HS_TOO_LONG; z.reads.clear(); return; }
    if ((int)z.reads.size() < o.o.min_passes) { z.host_status = HS_TOO_FEW; z.reads.clear(); return; }
    // --top-passes: "at most the top 60 full-length passes after sorting by median length" (docs/faq/accuracy-vs-passes.md:49-52):
    // the N passes whose length is closest to the median are kept, in their original order
    const size_t top = (size_t)o.o.top_passes;
    if (z.reads.size() > top) {
        std::vector<size_t> idx(z.reads.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        auto dist = [&](size_t i) { const double d = (double)z.reads[i].size() - med; return d < 0 ? -d : d; };
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return dist(a) < dist(b); });
        idx.resize(top);
        std::sort(idx.begin(), idx.end());
        std::vector<Subread> sel;
        for (size_t i : idx) sel.push_back(std::move(z.reads[i]));
        z.reads.swap(sel);
    }
    // the engine wants a ZMW's partial passes behind its full-length passes; it uses at most CCSX_MAX_PASSES in all
    for (auto &r : part) if (z.reads.size() < (size_t)CCSX_MAX_PASSES) z.reads.push_back(std::move(r));
}

struct ArenaPool {                          // free list of page-locked arenas: a batch holds its staging until its results are written
    std::mutex m;
    std::vector<std::unique_ptr<Arena>> free_;
    std::unique_ptr<Arena> get()
    {
        std::lock_guard<std::mutex> l(m);
        if (free_.empty()) return std::unique_ptr<Arena>(new Arena());
        auto a = std::move(free_.back()); free_.pop_back();
        return a;
    }
    void put(std::unique_ptr<Arena> a) { if (a) { std::lock_guard<std::mutex> l(m); free_.push_back(std::move(a)); } }
};

void pack(Batch &b, Arena &arena)
{
    b.read_off.assign(1, 0); b.base_off.assign(1, 0);
    b.slot.assign(b.zmws.size(), -1);
    int64_t total = 0;
    for (const ZmwIn &z : b.zmws) if (z.host_status == HS_OK) for (const Subread &r : z.reads) total += (int64_t)r.size();
    uint8_t *base = arena.reserve((size_t)3 * (size_t)total);
    uint8_t *bases = base, *pw = base + total, *ipd = base + 2 * total;
    int64_t at = 0;
    for (size_t i = 0; i < b.zmws.size(); ++i) {
        ZmwIn &z = b.zmws[i];
        if (z.host_status != HS_OK) continue;
        b.slot[i] = (int)b.zmw_id.size();
        b.zmw_id.push_back(z.zm);
        b.snr.insert(b.snr.end(), z.snr, z.snr + 4);
        for (size_t k = 0; k < z.reads.size(); ++k) {
            Subread &r = z.reads[k];
            const size_t L = r.size();
            r.decode_bases(bases + at); r.decode_pw(pw + at); r.decode_ip(ipd + at);      // the one and only decode: record -> staging
            at += (int64_t)L;
            b.flags.push_back((uint8_t)(r.strand | r.partial));
            b.base_off.push_back(at);
            r.keep.reset(); r.seq = r.pw_p = r.ip_p = nullptr;                              // the inflated bytes may go
        }
        b.read_off.push_back((int32_t)b.flags.size());
    }
    b.bases = bases; b.pw = pw; b.ipd = ipd; b.n_bases = total;
}

struct Report {
    int64_t input = 0, pass = 0;
    std::map<std::string, int64_t> fail;
    std::vector<int32_t> lens; std::vector<float> rqs; std::vector<int32_t> nps;   // one entry per written read
    int64_t bases_q30 = 0;        // written bases with a phred QV >= 30 ("Base quality >=Q30 (bp)", docs/faq/reports-aux-files.md:66)
};

const char *fail_label(int st)
{
    switch (st) {
        case HS_POOR_SNR: return "Below SNR threshold";
        case HS_NO_SUBREADS: return "Median length filter";
        case HS_TOO_FEW: case CCSX_TOO_FEW_PASSES: return "Lacking full passes";
        case CCSX_DRAFT_FAILURE: return "Draft generation error";
        case CCSX_TOO_MANY_UNUSABLE: return "Reads failed polishing";
        case CCSX_NON_CONVERGENT: return "CCS did not converge";
        case CCSX_TOO_SHORT: return "Draft below --min-length";
        case CCSX_TOO_LONG: case HS_TOO_LONG: return "Draft above --max-length";
        case CCSX_LOW_RQ: return "CCS below minimum RQ";
        case CCSX_EMPTY_WINDOW: return "Empty coverage windows";
        case CCSX_CAPACITY: return "Consensus outgrew its buffer";
        default: return "Unknown error";
    }
}

const char *status_name(int st)
{
    switch (st) {   // docs/faq/reports-aux-files.md:143-159
        case CCSX_SUCCESS: return "SUCCESS";
        case HS_POOR_SNR: return "POOR_SNR";
        case HS_NO_SUBREADS: return "NO_SUBREADS";
        case HS_TOO_FEW: case CCSX_TOO_FEW_PASSES: return "TOO_FEW_PASSES";
        case CCSX_DRAFT_FAILURE: return "DRAFT_FAILURE";
        case CCSX_TOO_MANY_UNUSABLE: return "TOO_MANY_UNUSABLE";
        case CCSX_NON_CONVERGENT: return "NON_CONVERGENT";
        case CCSX_TOO_SHORT: return "TOO_SHORT";
        case CCSX_TOO_LONG: case HS_TOO_LONG: return "TOO_LONG";
        case CCSX_LOW_RQ: return "POOR_QUALITY";
        case CCSX_EMPTY_WINDOW: return "EMPTY_WINDOW_DURING_POLISHING";
        case CCSX_CAPACITY: return "CAPACITY";
        default: return "EXCEPTION_THROWN";
    }
}

// the rows of "Exclusive failed counts" this path can produce, in the order of docs/faq/reports-aux-files.md:24-46 (the rows of the subsystems
// outside this path — heteroduplex, coverage drops, adapter / control classes — cannot occur and are not listed)
const char *const kFailOrder[] = {"Below SNR threshold", "Median length filter", "Lacking full passes", "Draft generation error",
                                  "Draft above --max-length", "Draft below --min-length", "Reads failed polishing", "Empty coverage windows",
                                  "CCS did not converge", "CCS below minimum RQ", "Consensus outgrew its buffer", "Unknown error"};

// statistics of one class of written reads (docs/faq/reports-aux-files.md:52-64: HiFi = predicted accuracy >= Q20, "<Q20", ">=Q30")
struct ReadClass {
    int64_t reads = 0, yield = 0, len_mean = 0, len_median = 0, n50 = 0, np_mean = 0;
    int qual_median = 0;
};

int rq_to_q(float rq) { return rq >= 1.0f ? 60 : (int)std::floor(-10.0 * std::log10(1.0 - (double)rq) + 1e-9); }

ReadClass classify(const Report &r, float rq_lo, float rq_hi)
{
    ReadClass c;
    std::vector<int32_t> len; std::vector<float> rq; int64_t nps = 0;
    for (size_t i = 0; i < r.lens.size(); ++i)
        if (r.rqs[i] >= rq_lo && r.rqs[i] < rq_hi) { len.push_back(r.lens[i]); rq.push_back(r.rqs[i]); nps += r.nps[i]; c.yield += r.lens[i]; }
    c.reads = (int64_t)len.size();
    if (len.empty()) return c;
    std::sort(len.begin(), len.end()); std::sort(rq.begin(), rq.end());
    c.len_mean = c.yield / c.reads;
    c.len_median = len[len.size() / 2];
    c.qual_median = rq_to_q(rq[rq.size() / 2]);
    c.np_mean = nps / c.reads;
    int64_t acc = 0;                                            // N50: the length L such that reads of length >= L hold half of the yield
    for (size_t i = len.size(); i-- > 0;) { acc += len[i]; if (2 * acc >= c.yield) { c.n50 = len[i]; break; } }
    return c;
}

std::string with_commas(int64_t v)                              // 63881 -> "63,881", as the reference prints its yield / length rows
{
    std::string d = std::to_string(v < 0 ? -v : v), o;
    for (size_t i = 0; i < d.size(); ++i) { if (i && (d.size() - i) % 3 == 0) o += ','; o += d[i]; }
    return v < 0 ? "-" + o : o;
}

void write_report(const Options &o, const Report &r)
{
    const int64_t failed = r.input - r.pass;
    auto pct = [](int64_t a, int64_t b) { return b ? 100.0 * (double)a / (double)b : 0.0; };
    const ReadClass hifi = classify(r, 0.99f, 2.0f), lowq = classify(r, -1.0f, 0.99f), q30 = classify(r, 0.999f, 2.0f);
    int64_t all_bases = 0;
    for (int32_t l : r.lens) all_bases += l;
    if (FILE *f = std::fopen(o.report.c_str(), "w")) {
        std::fprintf(f, "ZMWs input                    : %" PRId64 "\n\n", r.input);
        std::fprintf(f, "ZMWs pass filters             : %" PRId64 " (%.2f%%)\n", r.pass, pct(r.pass, r.input));
        std::fprintf(f, "ZMWs fail filters             : %" PRId64 " (%.2f%%)\n", failed, pct(failed, r.input));
        std::fprintf(f, "ZMWs shortcut filters         : 0 (0.00%%)\n\n");
        std::fprintf(f, "Exclusive failed counts\n");
        for (const char *k : kFailOrder) {
            auto it = r.fail.find(k);
            const int64_t c = it == r.fail.end() ? 0 : it->second;
            if (c == 0 && std::string(k) == "Consensus outgrew its buffer") continue;      // (not a row of the reference: shown only when it happened)
            std::fprintf(f, "%-30s: %" PRId64 " (%.2f%%)\n", k, c, pct(c, failed));
        }
        std::fprintf(f, "\n- - - - - - - - - - - - - - - : - - - - -\n\n");
        std::fprintf(f, "HiFi Reads                    : %s\n", with_commas(hifi.reads).c_str());
        std::fprintf(f, "HiFi Yield (bp)               : %s\n", with_commas(hifi.yield).c_str());
        std::fprintf(f, "HiFi Read Length (mean, bp)   : %s\n", with_commas(hifi.len_mean).c_str());
        std::fprintf(f, "HiFi Read Length (median, bp) : %s\n", with_commas(hifi.len_median).c_str());
        std::fprintf(f, "HiFi Read Length N50 (bp)     : %s\n", with_commas(hifi.n50).c_str());
        std::fprintf(f, "HiFi Read Quality (median)    : %d\n", hifi.qual_median);
        std::fprintf(f, "HiFi Number of Passes (mean)  : %" PRId64 "\n", hifi.np_mean);
        if (lowq.reads > 0) {                                    // reads below Q20 are written only under --min-rq < 0.99
            std::fprintf(f, "\n<Q20 Reads                    : %s\n", with_commas(lowq.reads).c_str());
            std::fprintf(f, "<Q20 Yield (bp)               : %s\n", with_commas(lowq.yield).c_str());
            std::fprintf(f, "<Q20 Read Length (mean, bp)   : %s\n", with_commas(lowq.len_mean).c_str());
            std::fprintf(f, "<Q20 Read Length (median, bp) : %s\n", with_commas(lowq.len_median).c_str());
            std::fprintf(f, "<Q20 Read Quality (median)    : %d\n", lowq.qual_median);
        }
        std::fprintf(f, "\n>=Q30 Reads                   : %s\n", with_commas(q30.reads).c_str());
        std::fprintf(f, ">=Q30 Yield (bp)              : %s\n", with_commas(q30.yield).c_str());
        std::fprintf(f, ">=Q30 Read Length (mean, bp)  : %s\n", with_commas(q30.len_mean).c_str());
        std::fprintf(f, ">=Q30 Read Length (median, bp): %s\n", with_commas(q30.len_median).c_str());
        std::fprintf(f, ">=Q30 Read Quality (median)   : %d\n", q30.qual_median);
        std::fprintf(f, "\nBase quality >=Q30 (bp)       : %s (%.1f%%)\n", with_commas(r.bases_q30).c_str(), pct(r.bases_q30, hifi.yield));
        std::fclose(f);
    }
}

// --report-json (docs/changelog.md:72 "JSON output of ccs_reports"; docs/faq/sqiie.md:42): the counts of ccs_report.txt, keyed by the row labels
void write_report_json(const Options &o, const Report &r)
{
    FILE *f = std::fopen(o.report_json.c_str(), "w");
    if (!f) { std::fprintf(stderr, "ccs: cannot write %s\n", o.report_json.c_str()); return; }
    std::fprintf(f, "{\n  \"zmws_input\": %" PRId64 ",\n  \"zmws_pass_filters\": %" PRId64 ",\n  \"zmws_fail_filters\": %" PRId64 ",\n  \"zmws_shortcut_filters\": 0,\n  \"exclusive_failed_counts\": {",
                 r.input, r.pass, r.input - r.pass);
    bool first = true;
    for (const char *k : kFailOrder) {
        auto it = r.fail.find(k);
        std::fprintf(f, "%s\n    \"%s\": %" PRId64, first ? "" : ",", k, it == r.fail.end() ? (int64_t)0 : it->second);
        first = false;
    }
    std::fprintf(f, "\n  }\n}\n");
    std::fclose(f);
}

// --hifi-summary-json (docs/faq/sqiie.md:45 "summary JSON file for hifi statistics"): the statistics block of ccs_report.txt
void write_hifi_summary(const Options &o, const Report &r)
{
    FILE *f = std::fopen(o.hifi_summary.c_str(), "w");
    if (!f) { std::fprintf(stderr, "ccs: cannot write %s\n", o.hifi_summary.c_str()); return; }
    int64_t all_bases = 0;
    for (int32_t l : r.lens) all_bases += l;
    auto cls = [&](const char *name, const ReadClass &c, bool last) {
        std::fprintf(f, "  \"%s\": {\"reads\": %" PRId64 ", \"yield_bp\": %" PRId64 ", \"read_length_mean\": %" PRId64 ", \"read_length_median\": %" PRId64
                        ", \"read_length_n50\": %" PRId64 ", \"read_quality_median\": %d, \"number_of_passes_mean\": %" PRId64 "}%s\n",
                     name, c.reads, c.yield, c.len_mean, c.len_median, c.n50, c.qual_median, c.np_mean, last ? "" : ",");
    };
    std::fprintf(f, "{\n");
    cls("hifi", classify(r, 0.99f, 2.0f), false);
    cls("below_q20", classify(r, -1.0f, 0.99f), false);
    cls("q30_and_above", classify(r, 0.999f, 2.0f), false);
    std::fprintf(f, "  \"bases\": %" PRId64 ",\n  \"bases_q30_and_above\": %" PRId64 "\n}\n", all_bases, r.bases_q30);
    std::fclose(f);
}

}  // namespace

int main(int argc, char **argv)
{
    Options opt;
    if (!parse(argc, argv, opt)) { usage(); return 2; }
    if (!opt.log_file.empty() && !std::freopen(opt.log_file.c_str(), "w", stderr)) {      // --log-file (docs/faq/sqiie.md:40): every log line goes there
        std::printf("ccs: cannot write %s\n", opt.log_file.c_str());
        return 1;
    }
    int nthreads = opt.threads > 0 ? opt.threads : effective_cores();
    if (nthreads < 1) nthreads = 1;
    try {
        ThreadPool pool(nthreads);
        if (!opt.write_synth.empty()) return write_synthetic(opt, pool);

        BgzfReader in(opt.in, pool);
        BamHeader hdr; read_header(in, hdr);

        // ---- model parameters: --model-file, else by the chemistry triple of the header (docs/faq/chemistry.md:27-56);
        // "Abort if chemistry information is missing in BAM header" (docs/changelog.md:66)
        ccsx_model model;
        std::string chem_desc;
        if (!opt.dump && !opt.host_only) {
            auto ds_value = [&](const char *key) -> std::string {
                const size_t rg = hdr.text.find("@RG");
                const size_t k = rg == std::string::npos ? rg : hdr.text.find(std::string(key) + "=", rg);
                if (k == std::string::npos) return "";
                const size_t a = k + std::strlen(key) + 1, e = hdr.text.find_first_of(";\t\n", a);
                return hdr.text.substr(a, e == std::string::npos ? std::string::npos : e - a);
            };
            const std::string bk = ds_value("BINDINGKIT"), sk = ds_value("SEQUENCINGKIT"), bc = ds_value("BASECALLERVERSION");
            if (!opt.model_file.empty()) {
                if (ccsx_model_load(opt.model_file.c_str(), &model)) { std::fprintf(stderr, "ccs: %s\n", ccsx_last_error()); return 1; }
                chem_desc = std::string(model.name) + " (" + opt.model_file + ")";
            } else {
                if (bk.empty() || sk.empty() || bc.empty()) {
                    std::fprintf(stderr, "ccs: missing chemistry information in the BAM header (@RG DS: BINDINGKIT, SEQUENCINGKIT, BASECALLERVERSION)\n");
                    return 1;
                }
                if (ccsx_model_for_chemistry(bk.c_str(), sk.c_str(), bc.c_str(), &model)) { std::fprintf(stderr, "ccs: %s\n", ccsx_last_error()); return 1; }
                chem_desc = std::string(model.name) + " for " + bk + "/" + sk + "/" + bc;
            }
            if (opt.log_level >= 2) std::fprintf(stderr, "ccs: consensus model %s\n", chem_desc.c_str());   // docs/changelog.md:101
        }

        // ---- devices (not needed for --dump-zmws): one engine handle per device
        std::vector<ccsx_handle> handles;
        if (!opt.dump && !opt.host_only) {
            const int ndev = ccsx_device_count();
            if (ndev <= 0) { std::fprintf(stderr, "ccs: no gfx950 GPU available (this build has no CPU consensus path)\n"); return 1; }
            if (opt.all_gpus) for (int d = 0; d < ndev; ++d) opt.gpus.push_back(d);
            if (opt.gpus.empty()) opt.gpus.push_back(0);
            for (int d : opt.gpus) {
                // (the handle's page-locked layout arrays are allocated by the creating thread: create it from the device's NUMA node, then come back)
                cpu_set_t aff; const bool have_aff = sched_getaffinity(0, sizeof(aff), &aff) == 0;
                const int node = ccsx_bind_thread_to_device(d);
                ccsx_handle h = nullptr;
                const int rc_create = ccsx_create(d, &model, &opt.o, &h);
                if (have_aff) sched_setaffinity(0, sizeof(aff), &aff);
                if (rc_create) { std::fprintf(stderr, "ccs: %s\n", ccsx_last_error()); return 1; }
                if (opt.log_level >= 2) std::fprintf(stderr, "ccs: device %d: NUMA node %d%s\n", d, ccsx_device_numa_node(d), node >= 0 ? " (its threads and page-locked staging are bound to it)" : " (no binding)");
                handles.push_back(h);
            }
        }

        g_plain_arenas = opt.host_only;
        // Round 6 (NUMA): every device has its own packing threads, staging pools and queue to its worker, all bound to the device's NUMA node, so a batch's
        // page-locked staging is local to the GPU that uploads it.  The ZMW stream is still ONE queue (to_pack): a packer whose device queue is full stops
        // drawing from it, so the work flows to the devices that are free (dynamic, not an i/N split).
        const size_t ndev_q = std::max<size_t>(1, handles.size());
        Channel<std::shared_ptr<Batch>> to_pack(2 * ndev_q), to_writer(4 * ndev_q);
        std::vector<std::unique_ptr<Channel<std::shared_ptr<Batch>>>> to_gpu_q;
        for (size_t d = 0; d < ndev_q; ++d) to_gpu_q.emplace_back(new Channel<std::shared_ptr<Batch>>(2));
        std::string movie;
        const auto t_start = std::chrono::steady_clock::now();

        // an exception on any pipeline thread ends the run with a message and exit code 1 (the other threads are drained, not killed)
        std::atomic<int> failed{0};
        std::mutex err_m;
        std::string err_msg;
        auto fail = [&](const std::string &m) { std::lock_guard<std::mutex> l(err_m); if (!failed.exchange(1)) err_msg = m; };

        // ---- IN.bam.pbi: random access for --chunk, and the number of ZMWs ahead for the progress line.  --chunk i/N with N > 1
        // REQUIRES the index (docs/faq/parallelize.md:9-13): every job of a sharded run must partition the ZMWs the same way, so a
        // missing, unreadable or stale index is fatal instead of a silent switch to another scheme
        bool have_pbi = false, chunk_done = false;
        int64_t chunk_zmws = 0, total_zmws = -1;
        int pbi_first_hole = -1;                              // the ZMW the index promises at the seek target (a stale index is an error)
        int pbi_next_hole = -1;                               // ... and right after the chunk's last record (-1: the chunk ends the file)
        int64_t pbi_chunk_records = -1, seen_records = 0;     // records the index counts inside the chunk
        {
            PbiIndex pbi;
            bool ok = false;
            std::string why = "not found";
            try { ok = read_pbi(opt.in + ".pbi", pool, pbi); }
            catch (const std::exception &e) { why = e.what(); if (opt.chunk_n <= 1) std::fprintf(stderr, "ccs: warning: ignoring %s.pbi: %s\n", opt.in.c_str(), e.what()); }
            if (opt.chunk_n > 1 && !(ok && pbi.size() > 0))
                throw std::runtime_error("--chunk needs a usable " + opt.in + ".pbi (" + why + "): generate the index first (docs/faq/parallelize.md:9-13)");
            if (ok && pbi.size() > 0) {
                std::vector<size_t> first;                        // first record of every ZMW (consecutive records of one hole number)
                for (size_t i = 0; i < pbi.size(); ++i) if (i == 0 || pbi.hole[i] != pbi.hole[i - 1]) first.push_back(i);
                const int64_t Z = (int64_t)first.size();
                total_zmws = Z;
                if (opt.chunk_n > 1) {
                    const int64_t lo = (opt.chunk_i - 1) * Z / opt.chunk_n, hi = (int64_t)opt.chunk_i * Z / opt.chunk_n;
                    have_pbi = true; chunk_zmws = hi - lo; total_zmws = chunk_zmws;
                    if (chunk_zmws == 0) chunk_done = true;
                    else {
                        in.seek((uint64_t)pbi.file_offset[first[(size_t)lo]]); pbi_first_hole = pbi.hole[first[(size_t)lo]];
                        pbi_chunk_records = (int64_t)((hi < Z ? first[(size_t)hi] : pbi.size()) - first[(size_t)lo]);
                        pbi_next_hole = hi < Z ? pbi.hole[first[(size_t)hi]] : -1;
                    }
                    std::fprintf(stderr, "ccs: chunk %d/%d = ZMWs %" PRId64 "..%" PRId64 " of %" PRId64 " by random access (%s.pbi)\n", opt.chunk_i, opt.chunk_n, lo, hi, Z, opt.in.c_str());
                }
            }
        }

        // ---- reader
        long long rd_us[3] = {0, 0, 0};                  // framing (BGZF inflate wait), waiting for record decode, grouping + filters + queue
        std::thread reader([&] {
            try {
            ZmwIn cur; bool have = false;
            int64_t nz = 0, nb = 0;
            auto batch = std::make_shared<Batch>();
            long long batch_cost = 0;                        // subread bases of the ZMWs in `batch` that will reach the engine
            auto emit_zmw = [&](ZmwIn &zin) {
                finish_zmw(zin, opt);
                if (opt.dump) {
                    unsigned long long hsh = 1469598103934665603ull;       // FNV-1a over bases, pw, ip of the kept passes (reader self-check)
                    std::vector<uint8_t> tmp;
                    for (const Subread &r : zin.reads) for (int a = 0; a < 3; ++a) {
                        tmp.resize(r.size());
                        if (a == 0) r.decode_bases(tmp.data()); else if (a == 1) r.decode_pw(tmp.data()); else r.decode_ip(tmp.data());
                        for (uint8_t x : tmp) { hsh ^= x; hsh *= 1099511628211ull; }
                    }
                    std::printf("%d%s\t%d\t%zu\t%.2f,%.2f,%.2f,%.2f\t%016llx\n", zin.zm, zin.strand_tag == 1 ? "/fwd" : (zin.strand_tag == 2 ? "/rev" : ""), zin.host_status,
                                zin.reads.size(), zin.snr[0], zin.snr[1], zin.snr[2], zin.snr[3], hsh);
                }
                else {
                    // cost-binned batches: the engine's work grows with passes x length (= the subread bases), which spreads over
                    // ~400x in a Sequel-II-like mix; a batch closes at --batch-size ZMWs or --batch-bases bases, whichever comes first, so every
                    // ticket the GPU workers draw from the shared queue costs about the same (and its staging stays bounded)
                    long long cost = 0;
                    if (zin.host_status == HS_OK) for (const Subread &r : zin.reads) cost += (long long)r.size();
                    const int lim_zmws = opt.batch > 0 ? opt.batch : auto_batch_zmws(nb);
                    const long long lim_bases = opt.batch_bases > 0 ? opt.batch_bases : kDefaultBasesPerZmw * lim_zmws;
                    if (!batch->zmws.empty() && batch_cost + cost > lim_bases) { batch->index = nb++; to_pack.push(batch); batch = std::make_shared<Batch>(); batch_cost = 0; }
                    batch->zmws.push_back(std::move(zin));
                    batch_cost += cost;
                    if ((int)batch->zmws.size() >= (opt.batch > 0 ? opt.batch : auto_batch_zmws(nb))) { batch->index = nb++; to_pack.push(batch); batch = std::make_shared<Batch>(); batch_cost = 0; }
                }
            };
            auto flush_zmw = [&] {
                if (!have) return;
                // --chunk i/N: with IN.bam.pbi a contiguous range of ZMWs (the reader has seeked to its first record and stops after
                // its last ZMW: docs/faq/parallelize.md:9-13); without an index round-robin over the ZMWs of the whole file
                const bool mine = have_pbi ? (nz < chunk_zmws) : true;
                cur.order = nz++;
                if (have_pbi && nz >= chunk_zmws) chunk_done = true;
                if (mine) {
                    // strand of every pass: cx REVERSE_PASS (32) / FORWARD_PASS (16) when present, else consecutive subreads alternate
                    for (size_t k = 0; k < cur.reads.size(); ++k) {
                        Subread &r = cur.reads[k];
                        r.strand = (r.cx >= 0 && (r.cx & 48)) ? (uint8_t)((r.cx & 32) ? 1 : 0) : (uint8_t)(k & 1);
                    }
                    if (!opt.by_strand) emit_zmw(cur);
                    else {                                   // each strand is treated as an individual entity (mode-by-strand.md:16-23)
                        ZmwIn f, rv;
                        f.zm = rv.zm = cur.zm; f.order = rv.order = cur.order; f.strand_tag = 1; rv.strand_tag = 2;
                        std::memcpy(f.snr, cur.snr, 16); std::memcpy(rv.snr, cur.snr, 16);
                        for (Subread &r : cur.reads) (r.strand ? rv : f).reads.push_back(std::move(r));
                        emit_zmw(f); emit_zmw(rv);
                    }
                }
                cur = ZmwIn(); have = false;
            };
            // this thread only frames records; decoding runs on the pool, results are consumed in order
            std::deque<std::future<std::vector<Subread>>> pending;
            auto consume = [&](std::vector<Subread> recs) {
                for (Subread &rec : recs) {
                    if (pbi_first_hole >= 0) {                       // first record after the seek
                        if (rec.zm != pbi_first_hole) throw std::runtime_error(opt.in + ".pbi does not match the BAM (stale index?): delete it or re-index");
                        pbi_first_hole = -1;
                    }
                    if (movie.empty()) movie = movie_of(rec.name);
                    if (chunk_done) break;                           // records of the next chunk that shared the last inflated run
                    if (!have || rec.zm != cur.zm) {
                        flush_zmw(); cur.zm = rec.zm; have = true;
                        if (chunk_done) {                            // the chunk's last ZMW has just ended: the index must agree on where
                            if (seen_records != pbi_chunk_records || rec.zm != pbi_next_hole)
                                throw std::runtime_error(opt.in + ".pbi does not match the BAM (record count / next ZMW of the chunk differ: stale index?)");
                            break;
                        }
                    }
                    ++seen_records;
                    if (rec.has_snr) std::memcpy(cur.snr, rec.snr, 16);
                    cur.reads.push_back(std::move(rec));
                }
            };
            for (;;) {
                auto raw = std::make_shared<RawChunk>();
                auto t0 = std::chrono::steady_clock::now();
                const bool more = !chunk_done && read_raw_chunk(in, *raw);
                rd_us[0] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                if (failed) break;                                  // an engine / writer failure ends the run: stop feeding it
                if (more) pending.push_back(pool.submit([raw] { return decode_chunk(std::shared_ptr<const RawChunk>(raw)); }));
                while (!pending.empty() && (!more || pending.size() > (size_t)(2 * pool.size() + 4))) {
                    t0 = std::chrono::steady_clock::now();
                    auto recs = pending.front().get();
                    auto t1 = std::chrono::steady_clock::now();
                    consume(std::move(recs)); pending.pop_front();
                    rd_us[1] += std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
                    rd_us[2] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
                }
                if (!more) break;
            }
            if (!chunk_done) {                                     // (a finished chunk ends inside the next chunk's first ZMW)
                flush_zmw();
                if (have_pbi && (seen_records != pbi_chunk_records || pbi_next_hole != -1 || !chunk_done))
                    throw std::runtime_error(opt.in + ".pbi does not match the BAM (the file ends before the chunk does: stale index?)");
            }
            if (!opt.dump && !batch->zmws.empty()) { batch->index = nb++; to_pack.push(batch); }
            } catch (const std::exception &e) { fail(std::string("reading ") + opt.in + ": " + e.what()); }
            to_pack.close();
        });
        if (opt.dump) {
            reader.join();
            if (opt.log_level >= 2)
                std::fprintf(stderr, "ccs: reader thread: framing/inflate %.2f s, waiting for record decode %.2f s, grouping+filters+queue %.2f s\n",
                             rd_us[0] * 1e-6, rd_us[1] * 1e-6, rd_us[2] * 1e-6);
            if (failed) std::fprintf(stderr, "ccs: %s\n", err_msg.c_str());
            return failed ? 1 : 0;
        }

        // ---- pack workers: SoA packing off the reader thread, straight into page-locked staging; result views are carved from a
        // second page-locked arena (the downloads are asynchronous DMA)
        std::vector<std::unique_ptr<ArenaPool>> in_pools, out_pools;
        for (size_t d = 0; d < ndev_q; ++d) { in_pools.emplace_back(new ArenaPool()); out_pools.emplace_back(new ArenaPool()); }
        std::atomic<long long> us_pack{0}, us_engine{0}, us_wait{0};       // summed over workers (--log-level INFO)
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us_since = [](std::chrono::steady_clock::time_point t) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count(); };
        const bool kin = opt.o.hifi_kinetics != 0;
        std::vector<std::thread> packers;
        const size_t n_packers = std::max<size_t>(1, handles.size()) * (size_t)opt.workers_per_gpu;
        std::vector<std::atomic<int>> packers_left(ndev_q);
        for (size_t d = 0; d < ndev_q; ++d) packers_left[d] = (int)(n_packers / ndev_q);
        for (size_t pk = 0; pk < n_packers; ++pk) packers.emplace_back([&, pk] {
            const int dev = (int)(pk % ndev_q);
            if (!handles.empty()) ccsx_bind_thread_to_device(opt.gpus[dev]);
            ArenaPool &in_pool = *in_pools[dev], &out_pool = *out_pools[dev];
            Channel<std::shared_ptr<Batch>> &to_gpu = *to_gpu_q[dev];
            std::shared_ptr<Batch> b;
            while (to_pack.pop(b)) {
                auto t0 = now();
                b->dev = dev;
                try {
                    b->in_arena = in_pool.get();
                    pack(*b, *b->in_arena);
                    const int n = (int)b->zmw_id.size();
                    b->n = n;
                    if (n > 0 && !opt.host_only) {
                        b->cb = ccsx_batch{n, (int32_t)b->flags.size(), b->n_bases, b->zmw_id.data(), b->snr.data(), b->read_off.data(),
                                           b->base_off.data(), b->bases, b->pw, b->ipd, b->flags.data()};
                        b->seq_off.resize(n + 1);
                        const int64_t cap = ccsx_result_layout(&b->cb, b->seq_off.data());
                        b->cap = cap;
                        b->out_arena = out_pool.get();
                        const size_t n4 = ((size_t)n * 4 + 63) & ~(size_t)63, c1 = ((size_t)cap + 63) & ~(size_t)63;
                        uint8_t *o = b->out_arena->reserve(9 * n4 + (kin ? 6 : 2) * c1);
                        auto take = [&](size_t bytes) { uint8_t *r = o; o += bytes; return r; };
                        b->status = (int32_t *)take(n4); b->seq_len = (int32_t *)take(n4); b->np = (int32_t *)take(n4); b->iters = (int32_t *)take(n4);
                        b->n_windows = (int32_t *)take(n4); b->fn = (int32_t *)take(n4); b->rn = (int32_t *)take(n4);
                        b->rq = (float *)take(n4); b->ec = (float *)take(n4);
                        b->seq = take(c1); b->qual = take(c1);
                        b->cr = ccsx_results{n, cap, b->seq_off.data(), b->status, b->seq_len, b->seq, b->qual, nullptr, b->rq, b->np, b->ec, b->iters, b->n_windows};
                        b->cr.fn = b->fn; b->cr.rn = b->rn;
                        if (kin) {
                            b->kin = take(4 * c1);
                            // planes are cap bytes apart in the library's result layout
                            b->cr.fi = b->kin; b->cr.fp = b->kin + cap; b->cr.ri = b->kin + 2 * cap; b->cr.rp = b->kin + 3 * cap;
                        }
                    }
                } catch (const std::exception &e) { fail(std::string("packing a batch: ") + e.what()); b->n = -1; }
                us_pack += us_since(t0);
                to_gpu.push(b);
            }
            if (--packers_left[dev] == 0) to_gpu.close();
        });

        if (opt.host_only) {                                  // no engine: count what arrives, hand the staging back
            int64_t nz = 0, nzok = 0, nbases = 0, nbatches = 0;
            std::thread sink([&] {
                std::shared_ptr<Batch> b;
                while (to_gpu_q[0]->pop(b)) { ++nbatches; nz += (int64_t)b->zmws.size(); if (b->n > 0) { nzok += b->n; nbases += b->n_bases; } in_pools[0]->put(std::move(b->in_arena)); }
            });
            reader.join();
            for (auto &w : packers) w.join();
            sink.join();
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            std::fprintf(stderr, "ccs: reader thread: framing/inflate %.2f s, waiting for record decode %.2f s, grouping+filters+queue %.2f s; packing (sum over %zu threads) %.2f s\n",
                         rd_us[0] * 1e-6, rd_us[1] * 1e-6, rd_us[2] * 1e-6, n_packers, us_pack.load() * 1e-6);
            std::printf("host-only: %" PRId64 " ZMWs read, %" PRId64 " packed (%" PRId64 " bases) in %.2f s = %.1f ZMWs/s on %d host threads + %zu pack threads; %" PRId64 " batches\n", nz, nzok, nbases, el,
                        nz / el, nthreads, n_packers, nbatches);
            if (failed) std::fprintf(stderr, "ccs: %s\n", err_msg.c_str());
            return failed ? 1 : 0;
        }

        // ---- GPU workers: one per device; up to three batches in flight through the asynchronous boundary
        std::vector<std::thread> workers;
        for (size_t wd = 0; wd < handles.size(); ++wd) workers.emplace_back([&, wd] {
            ccsx_handle h = handles[wd];
            ccsx_bind_thread_to_device(opt.gpus[wd]);
            ArenaPool &in_pool = *in_pools[wd];
            Channel<std::shared_ptr<Batch>> &to_gpu = *to_gpu_q[wd];
            std::deque<std::pair<ccsx_ticket, std::shared_ptr<Batch>>> inflight;
            auto retire = [&] {
                auto t0 = now();
                auto fr = std::move(inflight.front()); inflight.pop_front();
                std::shared_ptr<Batch> &b = fr.second;
                if (ccsx_wait(h, fr.first)) fail(std::string("consensus engine: ") + ccsx_last_error());
                else b->have_results = true;
                in_pool.put(std::move(b->in_arena));                // inputs are on the device (and consumed): the staging is free again
                b->bases = b->pw = b->ipd = nullptr;
                us_engine += us_since(t0);
                to_writer.push(b);
            };
            std::shared_ptr<Batch> b;
            for (;;) {
                auto t0 = now();
                if (!to_gpu.pop(b)) break;
                us_wait += us_since(t0);
                if (b->n <= 0 || failed) {                          // nothing for the GPU (all ZMWs failed on the host), or the run is ending
                    if (b->n > 0 || b->n < 0) b->have_results = false;
                    while (!inflight.empty()) retire();             // keep batch order towards the writer cheap: drain first
                    in_pool.put(std::move(b->in_arena));
                    to_writer.push(b);
                    continue;
                }
                if (inflight.size() >= 3) retire();
                t0 = now();
                ccsx_ticket t = -1;
                if (ccsx_submit(h, &b->cb, &b->cr, &t)) {
                    // a failed batch must never reach the writer looking like SUCCESS records: it carries no results,
                    // the run is marked failed and ends with exit code 1 and no output file
                    fail(std::string("consensus engine: ") + ccsx_last_error());
                    b->have_results = false;
                    while (!inflight.empty()) retire();
                    in_pool.put(std::move(b->in_arena));
                    to_writer.push(b);
                } else inflight.emplace_back(t, b);
                us_engine += us_since(t0);
            }
            while (!inflight.empty()) retire();
        });

        // ---- writer (restores batch order)
        Report rep;
        const bool fastq = opt.out.size() > 9 && opt.out.compare(opt.out.size() - 9, 9, ".fastq.gz") == 0;   // OUT.fastq.gz (docs/index.md:55-58)
        std::thread writer([&] {
            try {
            std::unique_ptr<BgzfWriter> outp;
            gzFile gzq = nullptr;
            if (fastq) { gzq = gzopen(opt.out.c_str(), "wb4"); if (!gzq) throw std::runtime_error("cannot create " + opt.out); }
            else outp.reset(new BgzfWriter(opt.out, pool));
            std::string fq;
            bool header_done = false;
            std::map<int64_t, std::shared_ptr<Batch>> hold;
            int64_t next = 0;
            RecordBuilder rb;
            PbiIndex pbi; std::vector<uint64_t> marks;             // OUT.bam.pbi: one entry per HiFi record
            std::vector<uint8_t> kin_rev;
            std::shared_ptr<Batch> b;
            double last_progress = -1e30;
            std::string metrics = "{\n  \"zmws\": [\n";
            bool first_metric = true;
            const bool want_metrics = !opt.suppress_reports || opt.metrics_named;
            gzFile gzm = want_metrics ? gzopen(opt.metrics.c_str(), "wb") : nullptr;
            auto flush_metrics = [&](bool force) {
                if (gzm && (force || metrics.size() > (1u << 20))) { gzwrite(gzm, metrics.data(), (unsigned)metrics.size()); metrics.clear(); }
            };
            static const uint8_t qvbin[94] = {3,3,3,3,3,3,3, 10,10,10,10,10,10,10, 17,17,17,17,17,17, 22,22,22,22,22, 27,27,27,27,27,
                                              35,35,35,35,35,35,35,35,35,35, 40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,
                                              40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40};
            auto emit = [&](Batch &bt) {
                if (!header_done && !fastq) {
                    write_header(*outp, "@HD\tVN:1.6\tSO:unknown\tpb:5.0.0\n@RG\tID:ccsamd01\tPL:PACBIO\tDS:READTYPE=CCS" +
                                          std::string(opt.o.hifi_kinetics && opt.by_strand ? ";Ipd:CodecV1=ip;PulseWidth:CodecV1=pw" : "") + "\tPU:" + movie +
                                          "\tPM:SEQUELII\n@PG\tID:ccs\tPN:ccs\tVN:amd-mi355x-r1\tDS:Generate circular consensus sequences (ccs) from subreads.\n");
                }
                header_done = true;
                for (size_t i = 0; i < bt.zmws.size(); ++i) {
                    ++rep.input;
                    const ZmwIn &z = bt.zmws[i];
                    int st = z.host_status;
                    const int s = bt.slot[i];
                    if (st == HS_OK) st = bt.have_results ? bt.status[s] : -1;      // -1: EXCEPTION_THROWN / "Unknown error" (engine failure)
                    if (want_metrics) {
                        const bool have = (s >= 0 && bt.have_results);
                        const int32_t isz = (have && bt.seq_len[s] > 0) ? bt.seq_len[s] : z.median_len;
                        char line[512];
                        std::snprintf(line, sizeof(line), "%s    {\"effective_coverage\": %.2f, \"has_tandem_repeat\": false, \"insert_size\": %d, \"num_full_passes\": %d, "
                                      "\"polymerase_length\": %" PRId64 ", \"predicted_accuracy\": %.6f, \"status\": \"%s\", \"zmw\": \"%s/%d%s\"}",
                                      first_metric ? "" : ",\n", have ? bt.ec[s] : 0.0f, isz, have ? bt.np[s] : z.n_full, z.polymerase_len,
                                      (have && bt.seq_len[s] > 0) ? bt.rq[s] : -1.0f, status_name(st), movie.c_str(), z.zm,
                                      z.strand_tag == 1 ? "/fwd" : (z.strand_tag == 2 ? "/rev" : ""));
                        metrics += line; first_metric = false;
                        flush_metrics(false);
                    }
                    if (st != CCSX_SUCCESS) { rep.fail[fail_label(st)]++; continue; }
                    ++rep.pass;
                    const int64_t o = bt.seq_off[s]; const int32_t len = bt.seq_len[s];
                    if (opt.qv_binning) for (int32_t q = 0; q < len; ++q) { uint8_t &v = bt.qual[o + q]; v = qvbin[v > 93 ? 93 : v]; }   // after rq (qv-binning.md:19-21)
                    // "Base quality >=Q30 (bp)" is a statement about the HiFi yield (docs/faq/reports-aux-files.md:66: 62,526 of the 63,881 HiFi bases while <Q20 reads
                    // are present): only reads with rq >= 0.99 count, and the percentage is taken over their bases
                    if (bt.rq[s] >= 0.99f) for (int32_t q = 0; q < len; ++q) rep.bases_q30 += bt.qual[o + q] >= 30 ? 1 : 0;
                    const std::string qname = movie + "/" + std::to_string(z.zm) + "/ccs" + (z.strand_tag == 1 ? "/fwd" : (z.strand_tag == 2 ? "/rev" : ""));
                    if (fastq) {
                        fq.clear(); fq += '@'; fq += qname; fq += '\n';
                        for (int32_t q = 0; q < len; ++q) fq += "ACGT"[bt.seq[o + q] & 3];
                        fq += "\n+\n";
                        for (int32_t q = 0; q < len; ++q) fq += (char)(33 + (bt.qual[o + q] > 93 ? 93 : bt.qual[o + q]));
                        fq += '\n';
                        if (gzwrite(gzq, fq.data(), (unsigned)fq.size()) != (int)fq.size()) throw std::runtime_error("short write (fastq.gz)");
                        rep.lens.push_back(len); rep.rqs.push_back(bt.rq[s]); rep.nps.push_back(bt.np[s]);
                        continue;
                    }
                    rb.begin(qname, bt.seq + o, bt.qual + o, (uint32_t)len);
                    rb.tagZ("RG", "ccsamd01");
                    rb.tagf("ec", bt.ec[s]);
                    rb.tagi("np", bt.np[s]);
                    rb.tagf("rq", bt.rq[s]);
                    rb.tagBf("sn", z.snr, 4);
                    rb.tagi("zm", z.zm);
                    if (opt.o.hifi_kinetics) {                      // docs/faq/kinetics.md:8-18, tag table docs/faq/bam-output.md:13-23
                        const size_t cap = (size_t)bt.cap;
                        const uint8_t *fi = bt.kin + o, *fp = fi + cap, *ri = fp + cap, *rp = ri + cap;
                        const uint32_t nf = bt.fn[s] > 0 ? (uint32_t)len : 0, nr = bt.rn[s] > 0 ? (uint32_t)len : 0;   // a strand without passes: empty lists
                        if (z.strand_tag) {                         // single-strand record: its own strand is the forward pair
                            rb.tagBC("ip", fi, nf); rb.tagBC("pw", fp, nf);
                        } else {
                            rb.tagBC("fi", fi, nf); rb.tagBC("fp", fp, nf); rb.tagi("fn", bt.fn[s]);
                            kin_rev.assign(ri, ri + nr); std::reverse(kin_rev.begin(), kin_rev.end());   // reverse strand: its own orientation
                            rb.tagBC("ri", kin_rev.data(), nr);
                            kin_rev.assign(rp, rp + nr); std::reverse(kin_rev.begin(), kin_rev.end());
                            rb.tagBC("rp", kin_rev.data(), nr); rb.tagi("rn", bt.rn[s]);
                        }
                    }
                    marks.push_back(outp->mark());
                    pbi.rg_id.push_back(0); pbi.q_start.push_back(0); pbi.q_end.push_back(len); pbi.hole.push_back(z.zm);
                    pbi.read_qual.push_back(bt.rq[s]); pbi.ctxt.push_back(0);
                    rb.finish(*outp);
                    rep.lens.push_back(len); rep.rqs.push_back(bt.rq[s]); rep.nps.push_back(bt.np[s]);
                }
            };
            while (to_writer.pop(b)) {
                hold[b->index] = b;
                while (!hold.empty() && hold.begin()->first == next) {
                    emit(*hold.begin()->second);
                    out_pools[hold.begin()->second->dev]->put(std::move(hold.begin()->second->out_arena));
                    hold.erase(hold.begin()); ++next;
                }
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
                if (opt.log_level >= 2 && el - last_progress >= opt.refresh_rate) {     // --refresh-rate (docs/faq/reports-aux-files.md:176-177)
                    last_progress = el;
                    if (total_zmws > 0 && rep.input > 0)               // with an index the number of ZMWs ahead is known: ETA (reports-aux-files.md:183-192)
                        std::fprintf(stderr, "%" PRId64 "/%.1f %" PRId64 "/%.1f ETA %.0f s\n", rep.input, rep.input / el * 60, rep.pass, rep.pass / el * 60,
                                     el * (double)(total_zmws - rep.input) / (double)rep.input);
                    else
                    std::fprintf(stderr, "%" PRId64 "/%.1f %" PRId64 "/%.1f\n", rep.input, rep.input / el * 60, rep.pass, rep.pass / el * 60);
                }
            }
            for (auto &kv : hold) emit(*kv.second);
            if (fastq) { if (gzclose(gzq) != Z_OK) throw std::runtime_error("short write (fastq.gz)"); }
            else {
                if (!header_done) write_header(*outp, "@HD\tVN:1.6\tSO:unknown\tpb:5.0.0\n@RG\tID:ccsamd01\tPL:PACBIO\tDS:READTYPE=CCS\tPU:unknown\n");
                outp->close();
                for (uint64_t m : marks) pbi.file_offset.push_back((int64_t)outp->virtual_offset(m));
                write_pbi(opt.out + ".pbi", pool, pbi);
            }
            if (gzm) { metrics += "\n  ]\n}\n"; flush_metrics(true); gzclose(gzm); }
            } catch (const std::exception &e) {
                fail(std::string("writing ") + opt.out + ": " + e.what());
                std::shared_ptr<Batch> drop;
                while (to_writer.pop(drop)) {}                   // keep the workers from blocking on a full queue
            }
        });

        reader.join();
        for (auto &w : packers) w.join();
        for (auto &w : workers) w.join();
        to_writer.close();
        writer.join();
        for (ccsx_handle h : handles) ccsx_destroy(h);
        if (!opt.suppress_reports || opt.report_named) write_report(opt, rep);
        if (!opt.report_json.empty()) write_report_json(opt, rep);            // named files are written even under --suppress-reports (docs/faq/sqiie.md:36-46
        if (!opt.hifi_summary.empty()) write_hifi_summary(opt, rep);          //  combines --suppress-reports with explicit report names)
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        if (opt.log_level >= 2)
            std::fprintf(stderr, "ccs: reader thread: framing/inflate %.2f s, waiting for record decode %.2f s, grouping+filters+queue %.2f s\n",
                         rd_us[0] * 1e-6, rd_us[1] * 1e-6, rd_us[2] * 1e-6);
        if (opt.log_level >= 2)
            std::fprintf(stderr, "ccs: GPU workers (sum over %zu): waiting for input %.2f s, packing (pack threads) %.2f s, submit + wait %.2f s\n", handles.size(),
                         us_wait.load() * 1e-6, us_pack.load() * 1e-6, us_engine.load() * 1e-6);
        if (opt.log_level >= 1)
            std::fprintf(stderr, "ccs: %" PRId64 " ZMWs in, %" PRId64 " HiFi reads out, %.2f s (%.1f ZMWs/s, %d host threads, %zu GPU worker%s)\n", rep.input, rep.pass, el,
                         rep.input / el, nthreads, handles.size(), handles.size() == 1 ? "" : "s");
        if (failed) {
            if (!err_msg.empty()) std::fprintf(stderr, "ccs: %s\n", err_msg.c_str());
            // no plausible-looking partial output after a failed run
            std::remove(opt.out.c_str());
            std::remove((opt.out + ".pbi").c_str());
            std::fprintf(stderr, "ccs: run failed, %s removed\n", opt.out.c_str());
        }
        return failed ? 1 : 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ccs: %s\n", e.what());
        return 1;
    }
}
