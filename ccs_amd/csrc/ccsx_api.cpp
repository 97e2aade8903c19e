// ccsx_api.cpp — C ABI over the HIP kernels: handle lifecycle, HBM layout, the asynchronous batch pipeline.
//
// One handle = one GPU, four HIP streams (H2D, draft stage, polish stage, D2H) and CCSX_SLOTS batch slots.  A slot owns
// the device copies of one batch (inputs, host-derived layout, per-ZMW state, outputs) and the page-locked host arrays its
// asynchronous uploads read from; the large POA / alignment scratch is shared: only the draft stage touches it, and the
// draft stages of all batches run back to back on the draft stream.  ccsx_submit() enqueues upload -> draft stage ->
// polish stage -> download of a batch and returns; while batch k is polished, batch k+1 uploads and is drafted / aligned
// and batch k-1 downloads (submit/wait tickets, double-buffered staging; the two-stage queue of
// docs/img/ccs-impl.png).  The synchronous entry points (upload / run / sync / download, consensus_batch) are the
// same machinery on slot 0.  There is no CPU fallback: without a usable device every entry point fails with a
// message (ccsx_last_error).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ccsx.h"
#include "ccsx_internal.h"
#include "ccsx_kernels.h"

#define HIPTRY(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            ccsx_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
            return -2;                                                                                         \
        }                                                                                                      \
    } while (0)

#define CCSX_SLOTS 3

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        // the new block first: a failed growth leaves the old buffer (and every KParams that points into it) intact
        // (a buffer that has to grow AGAIN belongs to a stream of unequal batches — the driver's cost-binned tickets of a mixed run: every regrowth is a
        // synchronous hipMalloc + hipFree, for the shared POA scratch tens of GB behind a stream synchronisation — so regrowth takes half as much again)
        // (the headroom is capped at 1 GB, and a request that fails WITH headroom is repeated at the exact size before anything is given up — the
        // POA scratch is sized to 3/4 of the handle's budget, half as much again may simply not exist)
        const size_t head = std::min<size_t>(p ? bytes / 2 : bytes / 8, (size_t)1 << 30);
        size_t want = bytes + head + 256;
        void *np_ = nullptr;
        hipError_t e = hipMalloc(&np_, want);
        if (e != hipSuccess) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&np_, want); }
        if (e != hipSuccess && p) {                      // not enough room for both: give the old block back and try once more, with and without headroom
            (void)hipGetLastError();
            (void)hipFree(p); p = nullptr; cap = 0;
            want = bytes + head + 256;
            e = hipMalloc(&np_, want);
            if (e != hipSuccess) { (void)hipGetLastError(); want = bytes + 256; e = hipMalloc(&np_, want); }
        }
        if (e != hipSuccess) { (void)hipGetLastError(); ccsx_set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); return -2; }
        if (p) (void)hipFree(p);
        p = np_; cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// grow-only page-locked host array: the source of asynchronous H2D copies must outlive the call that enqueues them
template <typename T> struct PinVec {
    T *p = nullptr;
    size_t cap = 0, n = 0;
    int assign(size_t count, T fill)
    {
        if (resize(count)) return -2;
        std::fill(p, p + count, fill);
        return 0;
    }
    int resize(size_t count)
    {
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = count + count / 8 + 64;
            hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
            if (e != hipSuccess) { ccsx_set_error(std::string("hipHostMalloc(") + std::to_string(want * sizeof(T)) + "): " + hipGetErrorString(e)); return -2; }
            cap = want;
        }
        n = count;
        return 0;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
};

struct Slot {
    // inputs
    DevBuf d_snr, d_read_off, d_base_off, d_bases, d_pw, d_ipd, d_flags;
    // layout
    DevBuf d_read_zmw, d_vcap, d_dcap, d_seq_off, d_wb_off, d_ent_off, d_wslot, d_zperm, d_rperm, d_quads, d_retry;
    // state
    DevBuf d_tabME, d_tabINS, d_tabDL, d_tabZ, d_dmask, d_draft, d_zmw_i32 /* 6 x n int32 */, d_wbounds, d_ticket;
    DevBuf d_avalid, d_ascore, d_ent;
    DevBuf d_wseq, d_wqv, d_wsum, d_wmeta;
    DevBuf d_out_seq, d_out_qual, d_out_raw, d_out_i32 /* 6 x n */, d_out_f32 /* 2 x n */;
    DevBuf d_wtpl, d_wtmeta, d_wkin, d_out_kin;   // HiFi kinetics only
    DevBuf d_din_len, d_din_bb;                   // caller-supplied drafts (ccsx_polish_batch): lengths, orientation references
    // host copies of the layout (page-locked: sources of the asynchronous uploads)
    PinVec<int32_t> read_zmw, vcap, dcap, zperm, rperm, wb_off, read_off, quads, qperm;
    PinVec<int64_t> seq_off, ent_off, base_off;
    KParams P;
    hipEvent_t ev[7] = {}, ev_up = nullptr, ev_done = nullptr;   // ev[0..5]: stage boundaries, ev[6]: start of the polish stage
    hipEvent_t ev_aux[7] = {};                                   // second stream: fork / first DP done / join of the POA stage; k_align16 launch done x 2, its trace-back done x 2
    bool staged = false, ran = false, inflight = false;
    bool tm_ok = false; ccsx_timings tm{};   // the slot's timings as taken by ccsx_wait (valid until the slot is staged again)
    ccsx_results *res = nullptr;      // destination of an in-flight submit
    ccsx_drafts *drafts_out = nullptr; // ... of an in-flight ccsx_submit_draft
    int mode = CCSX_RUN_FUSED;
    int64_t ticket = -1;
    // scratch this batch needs per resident POA graph / alignment
    size_t poa_slot_bytes = 0, align_slot_i32 = 0, align16_slot_i32 = 0;
    int align16_regions = 1;

    void release()
    {
        DevBuf *bufs[] = {&d_snr, &d_read_off, &d_base_off, &d_bases, &d_pw, &d_ipd, &d_flags, &d_read_zmw, &d_vcap, &d_dcap, &d_seq_off,
                          &d_wb_off, &d_ent_off, &d_wslot, &d_zperm, &d_rperm, &d_quads, &d_retry, &d_tabME, &d_tabINS, &d_tabDL, &d_tabZ, &d_dmask, &d_draft,
                          &d_zmw_i32, &d_wbounds, &d_ticket, &d_avalid, &d_ascore, &d_ent, &d_wseq, &d_wqv, &d_wsum, &d_wmeta, &d_out_seq,
                          &d_out_qual, &d_out_raw, &d_out_i32, &d_out_f32, &d_wtpl, &d_wtmeta, &d_wkin, &d_out_kin, &d_din_len, &d_din_bb};
        for (auto *b : bufs) b->release();
        read_zmw.release(); vcap.release(); dcap.release(); zperm.release(); rperm.release(); quads.release(); qperm.release(); wb_off.release();
        read_off.release(); seq_off.release(); ent_off.release(); base_off.release();
        for (auto &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (auto &e : ev_aux) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (ev_up) { (void)hipEventDestroy(ev_up); ev_up = nullptr; }
        if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
    }
};

// POA / alignment scratch is sized from the free memory of the device; handles created on the same device by
// different host threads must not all claim the same free bytes: sizing is serialised per process.
std::mutex g_scratch_mutex;

}  // namespace

struct ccsx_handle_s {
    int device = 0;
    hipStream_t s_in = nullptr, s_draft = nullptr, s_comp = nullptr, s_out = nullptr;   // s_comp: polish stage (and the synchronous calls' copies)
    hipStream_t s_aux = nullptr;                              // second stream of the POA stage (forked from and joined into s_draft inside one launch)
    hipEvent_t ev_epoch = nullptr, ev_epoch_nx = nullptr;   // origin of ccsx_timings.start_ms / end_ms: recorded at creation and moved forward every few
    float age_ms = 0.0f;              // end of the latest batch whose timings were read, relative to the current origin
    double epoch_ms = 0.0;            // minutes (epoch_ms = its distance from the creation), so that the float milliseconds HIP reports stay well below
                                      // 2^23 ms and keep their sub-microsecond resolution in long runs
    bool poisoned = false;            // a submit failed after work was enqueued: the handle refuses further batches
    ccsx_model model;
    ccsx_opts opts;
    DevBuf d_model, d_poa, d_align;   // shared by all slots
    Slot slot[CCSX_SLOTS];
    int last = 0;                     // slot of the most recent stage / run (stage accessors, timings)
    int64_t next_ticket = 0;
    int handles_on_device = 1;        // share of the device's free memory this handle may take for scratch
};

static void destroy_handle(ccsx_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->s_draft) (void)hipStreamSynchronize(h->s_draft);
    if (h->s_aux) (void)hipStreamSynchronize(h->s_aux);
    if (h->s_comp) (void)hipStreamSynchronize(h->s_comp);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    for (auto &s : h->slot) s.release();
    if (h->ev_epoch) (void)hipEventDestroy(h->ev_epoch);
    if (h->ev_epoch_nx) (void)hipEventDestroy(h->ev_epoch_nx);
    h->d_model.release(); h->d_poa.release(); h->d_align.release();
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_draft && h->s_draft != h->s_comp) (void)hipStreamDestroy(h->s_draft);
    if (h->s_aux) (void)hipStreamDestroy(h->s_aux);
    if (h->s_comp) (void)hipStreamDestroy(h->s_comp);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    delete h;
}

extern "C" {

int ccsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ccsx_device_numa_node(int device)
{
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id), device) != hipSuccess) return -1;
    return ccsx_pci_numa_node(id);
}

int ccsx_bind_thread_to_device(int device) { return ccsx_bind_thread_to_node(ccsx_device_numa_node(device)); }

void *ccsx_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { ccsx_set_error(std::string("hipHostMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e)); return nullptr; }
    return p;
}

void ccsx_free_pinned(void *p)
{
    if (p) (void)hipHostFree(p);
}

static int create_impl(ccsx_handle h)
{
    HIPTRY(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    // The polish stream gets the device's highest priority, the draft stage's streams the lowest: without the second POA stream the priorities change nothing
    // (tools/r03_prio.sh), with it they decide — three plain streams: 473 ms per 16384-ZMW step, polish high: 436, draft high: 438, no second stream: 441
    // (profiles/r05_poa_half_batches.txt).  CCSX_STAGE_PRIO=draft|polish|none overrides (none: plain streams).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char *pe_env = getenv("CCSX_STAGE_PRIO");
    const char *pe = pe_env ? (strcmp(pe_env, "none") ? pe_env : nullptr) : "polish";
    const int p_draft = (pe && !strcmp(pe, "draft")) ? prio_hi : prio_lo, p_polish = (pe && !strcmp(pe, "polish")) ? prio_hi : prio_lo;
    if (pe) HIPTRY(hipStreamCreateWithPriority(&h->s_comp, hipStreamNonBlocking, p_polish));
    else HIPTRY(hipStreamCreateWithFlags(&h->s_comp, hipStreamNonBlocking));
    if (h->opts.serial_stages) h->s_draft = h->s_comp;
    else if (pe) HIPTRY(hipStreamCreateWithPriority(&h->s_draft, hipStreamNonBlocking, p_draft));
    else HIPTRY(hipStreamCreateWithFlags(&h->s_draft, hipStreamNonBlocking));
    HIPTRY(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    // the POA stage's second stream (two half-batches, the threading of one under the DP of the other; not with serial stages, whose point is that a kernel's
    // counters are its own — CCSX_POA_SPLIT=2 forces it there for a measurement of the draft stage alone —; CCSX_POA_SPLIT=0 turns it off for an A/B; plain streams
    // (CCSX_STAGE_PRIO=none) get none: the three-way competition costs more than the overlap buys)
    { const char *e = getenv("CCSX_POA_SPLIT"); if (((!h->opts.serial_stages && pe) || (e && !strcmp(e, "2"))) && !(e && !strcmp(e, "0"))) { if (pe) HIPTRY(hipStreamCreateWithPriority(&h->s_aux, hipStreamNonBlocking, p_draft)); else HIPTRY(hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking)); } }
    HIPTRY(hipEventCreate(&h->ev_epoch));
    HIPTRY(hipEventCreate(&h->ev_epoch_nx));
    HIPTRY(hipEventRecord(h->ev_epoch, h->s_draft));         // the stream the tickets' first events are recorded on
    for (auto &s : h->slot) {
        for (auto &ev : s.ev) HIPTRY(hipEventCreate(&ev));
        for (auto &ev : s.ev_aux) HIPTRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPTRY(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
        HIPTRY(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    }
    if (h->d_model.reserve(sizeof(ccsx_model))) return -2;
    HIPTRY(hipMemcpy(h->d_model.p, &h->model, sizeof(ccsx_model), hipMemcpyHostToDevice));
    return 0;
}

int ccsx_create(int device_ordinal, const ccsx_model *model, const ccsx_opts *opts, ccsx_handle *out)
{
    if (!model || !opts || !out) { ccsx_set_error("ccsx_create: null argument"); return -1; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { ccsx_set_error("ccsx_create: no HIP device available (this library has no CPU fallback)"); return -2; }
    if (device_ordinal < 0 || device_ordinal >= n) { ccsx_set_error("ccsx_create: bad device ordinal"); return -1; }
    HIPTRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIPTRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        ccsx_set_error(std::string("ccsx_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
        return -2;
    }
    ccsx_handle h = new ccsx_handle_s();
    h->device = device_ordinal;
    h->model = *model;
    h->opts = *opts;
    if (h->opts.max_poa_cov < 1) h->opts.max_poa_cov = 1;
    if (h->opts.max_qv <= 0) h->opts.max_qv = 50;                    // (0 = default; a caller that zero-initialises the struct gets SPEC v7's cap)
    if (h->opts.max_qv > 93) h->opts.max_qv = 93;
    if (const char *e = std::getenv("CCSX_SERIAL_STAGES")) h->opts.serial_stages = std::atoi(e) != 0;   // A/B switch without a rebuild
    h->handles_on_device = opts->handles_per_device > 1 ? opts->handles_per_device : 1;
    const int rc = create_impl(h);
    if (rc) { destroy_handle(h); return rc; }        // no leak on a failed create (streams, events, device memory)
    *out = h;
    return 0;
}

int ccsx_destroy(ccsx_handle h)
{
    if (!h) return -1;
    destroy_handle(h);
    return 0;
}

static int validate(const ccsx_batch *b)
{
    if (!b || b->n_zmw <= 0 || !b->read_off || !b->base_off || !b->bases || !b->pw || !b->flags || !b->snr) {
        ccsx_set_error("ccsx_upload: null or empty batch");
        return -1;
    }
    if (b->read_off[0] != 0 || b->base_off[0] != 0) { ccsx_set_error("ccsx_upload: offsets must start at 0"); return -1; }
    for (int z = 0; z < b->n_zmw; ++z)
        if (b->read_off[z + 1] < b->read_off[z]) { ccsx_set_error("ccsx_upload: read_off not monotone"); return -1; }
    const int R = b->read_off[b->n_zmw];
    if (R != b->n_reads) { ccsx_set_error("ccsx_upload: n_reads != read_off[n_zmw]"); return -1; }
    for (int r = 0; r < R; ++r)
        if (b->base_off[r + 1] < b->base_off[r]) { ccsx_set_error("ccsx_upload: base_off not monotone"); return -1; }
    if (b->base_off[R] != b->n_bases) { ccsx_set_error("ccsx_upload: n_bases != base_off[n_reads]"); return -1; }
    for (int z = 0; z < b->n_zmw; ++z) {                  // partial passes (flag bit 1) must follow the ZMW's full-length passes
        bool partial = false;
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) {
            if (b->flags[r] & 2) partial = true;
            else if (partial) { ccsx_set_error("ccsx_upload: a full-length pass follows a partial pass (flags bit 1) in a ZMW"); return -1; }
        }
    }
    return 0;
}

// Stage a batch into a slot: host-derived layout, device buffers, H2D copies enqueued on `st` (nothing waits here except
// hipMalloc growth).  The batch's own arrays must stay valid until the copies have run (pinned arrays copy by DMA).
static int stage(ccsx_handle h, Slot &S, const ccsx_batch *b, hipStream_t st)
{
    if (validate(b)) return -1;
    const bool kin = h->opts.hifi_kinetics != 0;
    if (kin && !b->ipd) { ccsx_set_error("ccsx_upload: opts.hifi_kinetics needs batch.ipd"); return -1; }
    const int n = b->n_zmw, R = b->n_reads;
    const int64_t NB = b->n_bases;
    // ---- host-derived layout
    if (S.read_zmw.resize(R > 0 ? R : 1) || S.vcap.resize(n) || S.dcap.resize(n) || S.seq_off.assign(n + 1, 0) || S.wb_off.assign(n + 1, 0) ||
        S.ent_off.assign(R + 1, 0) || S.read_off.resize(n + 1) || S.base_off.resize(R + 1) || S.zperm.resize(n) || S.rperm.resize(R > 0 ? R : 1))
        return -2;
    int64_t maxL_max = 1, vcap_max = 1; int need_max = 2, nr_max = 1, nr_min = 1 << 30, n_quads = 0;
    for (int z = 0; z < n; ++z) {
        int64_t maxL = 0;
        int nr = b->read_off[z + 1] - b->read_off[z];
        { const int top = (h->opts.top_passes <= 0 || h->opts.top_passes > CCSX_MAX_PASSES) ? CCSX_MAX_PASSES : h->opts.top_passes; if (nr > top) nr = top; }
        nr_max = std::max(nr_max, nr); nr_min = std::min(nr_min, nr);
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) {
            S.read_zmw[r] = z;
            const int64_t L = b->base_off[r + 1] - b->base_off[r];
            if (L > maxL) maxL = L;
        }
        S.dcap[z] = (int32_t)ccsx_draft_cap(maxL);
        S.vcap[z] = (int32_t)ccsx_vertex_cap(maxL);
        const int wcap = S.dcap[z] / (CCSX_WIN_CORE - 3) + 4;   // cores are 19..25 columns (SPEC windows)
        S.seq_off[z + 1] = S.seq_off[z] + S.dcap[z];
        S.wb_off[z + 1] = S.wb_off[z] + wcap;
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) S.ent_off[r + 1] = S.ent_off[r] + ((2 * (wcap - 1) + 3) & ~3);   // (a multiple of 4: k_align16_tb stores four entries at a time)
        maxL_max = std::max(maxL_max, maxL); vcap_max = std::max<int64_t>(vcap_max, S.vcap[z]); need_max = std::max(need_max, 2 * (wcap - 1));
    }
    if (maxL_max > 65535) { ccsx_set_error("ccsx_upload: subreads longer than 65535 bases are not supported"); return -1; }
    std::memcpy(S.read_off.p, b->read_off, (size_t)(n + 1) * 4);
    std::memcpy(S.base_off.p, b->base_off, (size_t)(R + 1) * 8);
    // launch order: longest first (stable within 256-base classes, so a uniform batch keeps its input order and its
    // locality).  Mixed batches (BASELINE configs[4]: 1-25 kb, 3-50 passes) otherwise end on a few long stragglers.
    {
        auto order_by = [](PinVec<int32_t> &perm, size_t cnt_items, auto len_of) {
            const int NBK = 258;
            std::vector<int32_t> cnt(NBK + 1, 0);
            auto cls = [&](int64_t l) { int c = (int)(l >> 8); return NBK - 1 - (c > NBK - 1 ? NBK - 1 : c); };   // descending length
            for (size_t i = 0; i < cnt_items; ++i) ++cnt[cls(len_of(i)) + 1];
            for (int i = 0; i < NBK; ++i) cnt[i + 1] += cnt[i];
            for (size_t i = 0; i < cnt_items; ++i) perm[cnt[cls(len_of(i))]++] = (int32_t)i;
        };
        order_by(S.zperm, (size_t)n, [&](size_t z) { return (int64_t)S.dcap[z]; });
        if (R > 0) order_by(S.rperm, (size_t)R, [&](size_t r) { return b->base_off[r + 1] - b->base_off[r]; });
        // k_align16's work items: up to four consecutive passes of one ZMW (of the passes the engine uses), longest first
        std::vector<int32_t> qpack; std::vector<int64_t> qlen;
        const int top = (h->opts.top_passes <= 0 || h->opts.top_passes > CCSX_MAX_PASSES) ? CCSX_MAX_PASSES : h->opts.top_passes;
        for (int z = 0; z < n; ++z) {
            const int r0 = b->read_off[z];
            int nr = std::min(b->read_off[z + 1] - r0, top);
            while (nr > 0 && (b->flags[r0 + nr - 1] & 2)) --nr;   // partial passes are aligned by k_rescue (anchored at one end), not here
            for (int g = 0; g < nr; g += 4) {
                const int c = std::min(4, nr - g);
                int64_t ml = 0;
                for (int q = 0; q < c; ++q) ml = std::max<int64_t>(ml, b->base_off[r0 + g + q + 1] - b->base_off[r0 + g + q]);
                qpack.push_back(((r0 + g) << 2) | (c - 1)); qlen.push_back(ml);
            }
        }
        n_quads = (int)qpack.size();
        if (S.quads.resize(n_quads > 0 ? n_quads : 1)) return -2;
        if (n_quads > 0) {
            if (S.qperm.resize(n_quads)) return -2;
            order_by(S.qperm, (size_t)n_quads, [&](size_t q) { return qlen[q]; });
            for (int q = 0; q < n_quads; ++q) S.quads[q] = qpack[S.qperm[q]];
        }
    }
    const int64_t total_wslots = (int64_t)S.wb_off[n] - n;
    if (total_wslots > 0x7fff0000ll) { ccsx_set_error("ccsx_upload: batch too large (more than 2^31 window slots): split it"); return -1; }

#define UP(buf, src, bytes)                                                                                    \
    do {                                                                                                       \
        if ((buf).reserve(bytes)) return -2;                                                                   \
        HIPTRY(hipMemcpyAsync((buf).p, (src), (bytes), hipMemcpyHostToDevice, st));                             \
    } while (0)
    UP(S.d_snr, b->snr, (size_t)n * 16);
    UP(S.d_read_off, b->read_off, (size_t)(n + 1) * 4);
    UP(S.d_base_off, b->base_off, (size_t)(R + 1) * 8);
    UP(S.d_bases, b->bases, (size_t)NB);
    UP(S.d_pw, b->pw, (size_t)NB);
    if (kin) UP(S.d_ipd, b->ipd, (size_t)NB);
    UP(S.d_flags, b->flags, (size_t)(R > 0 ? R : 1));
    UP(S.d_read_zmw, S.read_zmw.p, (size_t)(R > 0 ? R : 1) * 4);
    UP(S.d_vcap, S.vcap.p, (size_t)n * 4);
    UP(S.d_dcap, S.dcap.p, (size_t)n * 4);
    UP(S.d_seq_off, S.seq_off.p, (size_t)(n + 1) * 8);
    UP(S.d_wb_off, S.wb_off.p, (size_t)(n + 1) * 4);
    UP(S.d_ent_off, S.ent_off.p, (size_t)(R + 1) * 8);
    UP(S.d_zperm, S.zperm.p, S.zperm.size() * 4);
    UP(S.d_rperm, S.rperm.p, S.rperm.size() * 4);
    UP(S.d_quads, S.quads.p, S.quads.size() * 4);
#undef UP

    const int64_t cap_total = S.seq_off[n];
#define RES(buf, bytes) do { if ((buf).reserve(bytes)) return -2; } while (0)
    RES(S.d_tabME, (size_t)n * 192 * 4); RES(S.d_tabINS, (size_t)n * 192 * 4); RES(S.d_tabDL, (size_t)n * 16 * 4); RES(S.d_tabZ, (size_t)n * 32 * 4);
    RES(S.d_draft, (size_t)cap_total);
    RES(S.d_zmw_i32, (size_t)n * 4 * 7);   // draft_len, nwin, zstat, nreads_used, np, zref, nfull
    RES(S.d_wbounds, (size_t)S.wb_off[n] * 4);
    RES(S.d_wslot, (size_t)(total_wslots + 1) * 4 + ((size_t)n + 2) * 4);      // compact window map + first index per ZMW, built by k_wmap
    RES(S.d_ticket, 256);
    RES(S.d_avalid, (size_t)(R > 0 ? R : 1)); RES(S.d_ascore, (size_t)(R > 0 ? R : 1) * 4);
    RES(S.d_retry, ((size_t)(R > 0 ? R : 1) + 16) * 4);
    RES(S.d_ent, (size_t)(S.ent_off[R] + 1) * 4); RES(S.d_dmask, (size_t)(S.ent_off[R] + 1) * 4);
    RES(S.d_wseq, (size_t)(total_wslots + 1) * 32); RES(S.d_wqv, (size_t)(total_wslots + 1) * 32 * 4);
    RES(S.d_wsum, (size_t)(total_wslots + 1) * 4); RES(S.d_wmeta, (size_t)(total_wslots + 1) * 16);
    RES(S.d_out_seq, (size_t)cap_total); RES(S.d_out_qual, (size_t)cap_total); RES(S.d_out_raw, (size_t)cap_total * 4);
    RES(S.d_out_i32, (size_t)n * 4 * 6); RES(S.d_out_f32, (size_t)n * 4 * 2);
    if (kin) {
        RES(S.d_wtpl, (size_t)(total_wslots + 1) * 32); RES(S.d_wtmeta, (size_t)(total_wslots + 1) * 4);
        RES(S.d_wkin, (size_t)(total_wslots + 1) * 32 * 4); RES(S.d_out_kin, (size_t)cap_total * 4);
    }

    // ---- resident POA graphs / alignment slots: as many as fit this handle's share of the free memory, never more than
    // the work.  The scratch is shared by the handle's batch slots; it only grows, and growing waits for the compute stream.
    // (per vertex CCSX_POA_BYTES_PER_VERTEX, ccsx_kernels.h; the kernels' layout static_asserts the same figure) + the pass's path + the state block
    S.poa_slot_bytes = (((size_t)vcap_max + 64) * CCSX_POA_BYTES_PER_VERTEX + (size_t)maxL_max * 4 + 1024 + 255) & ~(size_t)255;
    S.align_slot_i32 = (size_t)need_max * 128 + 4 * (size_t)need_max + 64;   // (origin, dirty bits) per cell and edge + band starts + best cell (score, row, entry row) per edge
    // k_align16 stores a quad's moves instead (2 bits per band row and column + 2 bits of band step and an edge flag per column: 18 words per block of 16 draft columns and pass)
    int64_t dcap_max = 16;
    for (int z = 0; z < n; ++z) dcap_max = std::max<int64_t>(dcap_max, S.dcap[z]);
    S.align16_slot_i32 = (size_t)4 * (size_t)ccsx_tb_stride((int)dcap_max) + 16;        // (four passes' moves + the four final band starts, padded)
    int poa_slots, align_slots, align16_slots;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mutex);
        size_t freeb = 0, totalb = 0;
        HIPTRY(hipMemGetInfo(&freeb, &totalb));
        freeb /= (size_t)h->handles_on_device;
        freeb += h->d_poa.cap + h->d_align.cap;                      // what we already hold is reusable
        const size_t budget = freeb > (size_t)6 << 30 ? freeb - ((size_t)4 << 30) : freeb / 2;
        poa_slots = h->opts.poa_slots > 0 ? h->opts.poa_slots : 16384;   // four graphs per wave: 16384 = 4 waves per SIMD
        poa_slots = std::min(poa_slots, n);
        poa_slots = (int)std::min<size_t>((size_t)poa_slots, std::max<size_t>(1, (budget * 3 / 4) / S.poa_slot_bytes));
        align_slots = std::min(4096, std::max(R, 2));             // the 64-row retry and the split alignment (two slots per pass) run grid-stride loops of <= 4096 workgroups
        align_slots = (int)std::min<size_t>((size_t)align_slots, std::max<size_t>(2, (budget / 16) / (S.align_slot_i32 * 4)));
        // every quad of the batch in ONE launch if that fits 12 GB (16384 ZMWs x 10 passes x 10 kb: 9.3 GB — the slot is sized for the draft CAPACITY of the batch's
        // longest insert); a batch of long inserts and many passes takes a few equal launches instead of tens of GB of scratch (a launch's trace-back still has
        // thousands of waves)
        int align16_regions = 1;
        {
            const size_t fit = std::max<size_t>(1, std::min<size_t>(budget / 8, (size_t)12 << 30) / (S.align16_slot_i32 * 4));
            const size_t nq = (size_t)std::max(n_quads, 1);
            static const bool one_region = [] { const char *e = getenv("CCSX_A16_ONE_REGION"); return e && e[0] == '1'; }();   // (A/B switch: equal launches in ONE region, in sequence)
            if (nq <= fit) align16_slots = (int)nq;
            else if (one_region) { const size_t launches = (nq + fit - 1) / fit; align16_slots = (int)((nq + launches - 1) / launches); }
            else {                                                       // two regions of half the space: launch c's trace-back runs under launch c + 1
                const size_t half = std::max<size_t>(1, fit / 2), launches = (nq + half - 1) / half;
                align16_slots = (int)((nq + launches - 1) / launches);
                align16_regions = 2;
            }
        }
        static const int max16 = [] { const char *e = getenv("CCSX_ALIGN16_MAX_SLOTS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();   // test hook: forces several launches
        if (max16 > 0 && align16_slots > max16) { align16_slots = max16; align16_regions = 2; }
        const size_t need_align = ((size_t)align_slots * S.align_slot_i32 + (size_t)align16_regions * align16_slots * S.align16_slot_i32) * 4;   // the retry's slots behind k_align16's
        S.align16_regions = align16_regions;
        if ((size_t)poa_slots * S.poa_slot_bytes > h->d_poa.cap || need_align > h->d_align.cap) {
            HIPTRY(hipStreamSynchronize(h->s_draft));                // kernels of an earlier batch may still use the old scratch
            HIPTRY(hipStreamSynchronize(h->s_comp));
            for (int attempt = 0;; ++attempt) {                      // another process / handle may have taken the memory meanwhile
                if (h->d_poa.reserve((size_t)poa_slots * S.poa_slot_bytes) == 0) break;
                if (attempt >= 4 || poa_slots <= 1) return -2;
                poa_slots = std::max(1, poa_slots / 2);
            }
            RES(h->d_align, need_align);
        }
    }
#undef RES

    KParams &P = S.P;
    std::memset(&P, 0, sizeof(P));
    P.n_zmw = n; P.n_reads = R; P.maxL_max = (int32_t)maxL_max; P.vcap_max = (int32_t)vcap_max; P.need_max = need_max; P.max_reads = nr_max;
    (void)nr_min;
    P.opts = h->opts;
    P.perr_floor = ccsx_perr_floor(h->opts.max_qv);
    P.model = (const ccsx_model *)h->d_model.p;
    P.snr = (const float *)S.d_snr.p; P.read_off = (const int32_t *)S.d_read_off.p; P.base_off = (const int64_t *)S.d_base_off.p;
    P.bases = (const uint8_t *)S.d_bases.p; P.pw = (const uint8_t *)S.d_pw.p; P.flags = (const uint8_t *)S.d_flags.p;
    P.read_zmw = (const int32_t *)S.d_read_zmw.p; P.vcap = (const int32_t *)S.d_vcap.p; P.dcap = (const int32_t *)S.d_dcap.p;
    P.seq_off = (const int64_t *)S.d_seq_off.p; P.wb_off = (const int32_t *)S.d_wb_off.p; P.ent_off = (const int64_t *)S.d_ent_off.p; P.wslot_zmw = (int32_t *)S.d_wslot.p; P.wstart = P.wslot_zmw + (total_wslots + 1);
    P.zmw_perm = (const int32_t *)S.d_zperm.p; P.read_perm = (const int32_t *)S.d_rperm.p;
    P.quads = (const int32_t *)S.d_quads.p; P.n_quads = n_quads; P.align_retry = (int32_t *)S.d_retry.p;
    P.tabME = (float *)S.d_tabME.p; P.tabINS = (float *)S.d_tabINS.p; P.tabDL = (float *)S.d_tabDL.p; P.tabZ = (float *)S.d_tabZ.p;
    P.draft = (uint8_t *)S.d_draft.p;
    int32_t *zi = (int32_t *)S.d_zmw_i32.p;
    P.draft_len = zi; P.nwin = zi + n; P.zstat = zi + 2 * (size_t)n; P.nreads_used = zi + 3 * (size_t)n; P.np = zi + 4 * (size_t)n; P.zref = zi + 5 * (size_t)n; P.nfull = zi + 6 * (size_t)n;
    P.wbounds = (int32_t *)S.d_wbounds.p;
    P.ticket_poa = (int32_t *)S.d_ticket.p; P.ticket_align = P.ticket_poa + 1; P.debug = P.ticket_poa + 4; P.phase = (unsigned long long *)(P.ticket_poa + 16);
    P.poa_scratch = (uint8_t *)h->d_poa.p; P.poa_slot_bytes = S.poa_slot_bytes; P.poa_slots = poa_slots;
    P.align_scratch = (int32_t *)h->d_align.p; P.align_slot_i32 = S.align_slot_i32; P.align_slots = align_slots;
    P.align16_slot_i32 = S.align16_slot_i32; P.align16_slots = align16_slots;
    P.align16_regions = S.align16_regions;
    P.retry_scratch = P.align_scratch + (size_t)S.align16_regions * align16_slots * S.align16_slot_i32;
    P.avalid = (uint8_t *)S.d_avalid.p; P.ascore = (int32_t *)S.d_ascore.p; P.ent = (int32_t *)S.d_ent.p; P.dmask = (uint32_t *)S.d_dmask.p;
    P.total_wslots = total_wslots;
    if (ccsx_polish_lds(nr_max, &P.pw_obs_bytes, &P.pw_gb_floats)) { ccsx_set_error("ccsx_upload: cannot size the polish kernel's LDS"); return -2; }
    P.wseq = (uint8_t *)S.d_wseq.p; P.wqv = (float *)S.d_wqv.p; P.wsum = (float *)S.d_wsum.p; P.wmeta = (int4 *)S.d_wmeta.p;
    P.out_seq = (uint8_t *)S.d_out_seq.p; P.out_qual = (uint8_t *)S.d_out_qual.p; P.out_raw = (float *)S.d_out_raw.p;
    int32_t *oi = (int32_t *)S.d_out_i32.p;
    P.out_status = oi; P.out_len = oi + n; P.out_iters = oi + 2 * (size_t)n; P.out_nwin = oi + 3 * (size_t)n;
    P.out_fn = oi + 4 * (size_t)n; P.out_rn = oi + 5 * (size_t)n;
    if (kin) {
        P.ipd = (const uint8_t *)S.d_ipd.p;
        P.wtpl = (uint8_t *)S.d_wtpl.p; P.wtmeta = (short2 *)S.d_wtmeta.p; P.wkin = (uchar4 *)S.d_wkin.p;
        P.out_kin = (uint8_t *)S.d_out_kin.p; P.kin_plane = cap_total;
    }
    float *of = (float *)S.d_out_f32.p;
    P.out_rq = of; P.out_ec = of + n;
    S.staged = true; S.ran = false; S.tm_ok = false;
    return 0;
}

// all kernels of one staged batch; every launch status is captured (a bad launch configuration fails here, not at the next sync)
static int launch(ccsx_handle h, Slot &S)
{
    (void)hipGetLastError();
    // the shared scratch may have been re-allocated (a later, larger batch) since this slot was staged
    S.P.poa_scratch = (uint8_t *)h->d_poa.p; S.P.align_scratch = (int32_t *)h->d_align.p;
    if ((size_t)S.P.poa_slots * S.P.poa_slot_bytes > h->d_poa.cap) S.P.poa_slots = (int)std::max<size_t>(1, h->d_poa.cap / S.P.poa_slot_bytes);
    {   // (the scratch only grows, so what was sized at staging still fits; the clamps are for a failed growth)
        const size_t words = h->d_align.cap / 4, reg = (size_t)std::max(1, S.P.align16_regions);
        if ((size_t)S.P.align_slots * S.P.align_slot_i32 + reg * S.P.align16_slots * S.P.align16_slot_i32 > words) {
            if ((size_t)S.P.align_slots * S.P.align_slot_i32 > words / 2) S.P.align_slots = (int)std::max<size_t>(1, (words / 2) / S.P.align_slot_i32);
            const size_t used = (size_t)S.P.align_slots * S.P.align_slot_i32, left = words > used ? words - used : 0;
            S.P.align16_slots = (int)std::max<size_t>(1, left / (reg * S.P.align16_slot_i32));
        }
        S.P.retry_scratch = S.P.align_scratch + reg * S.P.align16_slots * S.P.align16_slot_i32;
    }
    if (!h->d_poa.p || !h->d_align.p) { ccsx_set_error("kernel launch refused: the POA / alignment scratch is not allocated (an earlier allocation failed)"); return -2; }
    const char *failed = ccsx_launch_all(S.P, h->s_draft, h->s_comp, S.ev, S.mode, h->s_aux, h->s_aux ? S.ev_aux : nullptr);
    if (failed) { ccsx_set_error(std::string("kernel launch failed: ") + failed); return -2; }
    S.ran = true;
    return 0;
}

static int check_results(const Slot &S, const ccsx_results *res, bool have_kin)
{
    const int n = S.P.n_zmw;
    if (!res || res->n_zmw != n || res->seq_capacity < S.seq_off[n]) { ccsx_set_error("ccsx_download: result buffers too small"); return -1; }
    if (!have_kin && (res->fi || res->fp || res->ri || res->rp)) {
        ccsx_set_error("ccsx_download: kinetics buffers given but the handle was created without opts.hifi_kinetics");
        return -1;
    }
    return 0;
}

static int enqueue_download(Slot &S, ccsx_results *res, hipStream_t s)
{
    const int n = S.P.n_zmw;
    const KParams &P = S.P;
#define DOWN(dst, src, bytes) do { if (dst) HIPTRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, s)); } while (0)
    DOWN(res->status, P.out_status, (size_t)n * 4);
    DOWN(res->seq_len, P.out_len, (size_t)n * 4);
    DOWN(res->iters, P.out_iters, (size_t)n * 4);
    DOWN(res->n_windows, P.out_nwin, (size_t)n * 4);
    DOWN(res->rq, P.out_rq, (size_t)n * 4);
    DOWN(res->ec, P.out_ec, (size_t)n * 4);
    DOWN(res->np, P.np, (size_t)n * 4);
    DOWN(res->seq, P.out_seq, (size_t)S.seq_off[n]);
    DOWN(res->qual, P.out_qual, (size_t)S.seq_off[n]);
    DOWN(res->raw_qv, P.out_raw, (size_t)S.seq_off[n] * 4);
    DOWN(res->fn, P.out_fn, (size_t)n * 4);
    DOWN(res->rn, P.out_rn, (size_t)n * 4);
    if (P.out_kin) {
        const size_t pl = (size_t)S.seq_off[n];
        DOWN(res->fi, P.out_kin, pl); DOWN(res->fp, P.out_kin + pl, pl);
        DOWN(res->ri, P.out_kin + 2 * pl, pl); DOWN(res->rp, P.out_kin + 3 * pl, pl);
    }
#undef DOWN
    return 0;
}

// the host-side part of a finished slot's outputs: the capacity-layout offsets the caller's structs carry, and the drafts' backbone words without the cascade's marks
static void finish_outputs(Slot &S)
{
    if (S.res && S.res->seq_off) std::memcpy(S.res->seq_off, S.seq_off.p, (size_t)(S.P.n_zmw + 1) * 8);
    if (ccsx_drafts *d = S.drafts_out) {
        const int n = S.P.n_zmw;
        std::memcpy(d->seq_off, S.seq_off.p, (size_t)(n + 1) * 8);
        if (d->win_off) for (int z = 0; z <= n; ++z) d->win_off[z] = S.wb_off[z];
        for (int z = 0; z < n; ++z) d->backbone[z] &= 255;   // (the device word also carries the cascade's marks)
    }
    S.drafts_out = nullptr;                              // (done once)
}

// ---- asynchronous pipeline -------------------------------------------------------------------------------
// The origin of ccsx_timings.start_ms / end_ms moves forward every 5 minutes of handle lifetime so that HIP's float milliseconds keep their resolution.
// not inside the getter (it stalled behind every queued download and relied on negative elapsed times for slots recorded before the new origin) — only when
// NO slot is in flight, i.e. every event of a completed slot lies before the new origin and none is pending: called from submit.  CCSX_EPOCH_REBASE_MS: test hook.
static int rebase_epoch(ccsx_handle h)
{
    static const float limit = [] { const char *e = std::getenv("CCSX_EPOCH_REBASE_MS"); return e ? (float)std::atof(e) : 300000.0f; }();
    if (h->age_ms <= limit) return 0;
    for (auto &S : h->slot) if (S.inflight) return 0;
    float d = 0.0f;
    HIPTRY(hipEventRecord(h->ev_epoch_nx, h->s_draft));
    HIPTRY(hipEventSynchronize(h->ev_epoch_nx));
    HIPTRY(hipEventElapsedTime(&d, h->ev_epoch, h->ev_epoch_nx));
    h->epoch_ms += (double)d;
    std::swap(h->ev_epoch, h->ev_epoch_nx);
    h->age_ms = 0.0f;
    for (auto &S : h->slot) S.ran = false;               // events recorded before the new origin can no longer be measured against it; what ccsx_wait took (Slot::tm) stays readable
    return 0;
}

static int check_drafts(const Slot &S, const ccsx_drafts *d, bool input)
{
    const int n = S.P.n_zmw;
    if (!d || d->n_zmw != n || !d->seq || !d->len || !d->backbone || !d->seq_off || d->seq_capacity < S.seq_off[n] || (!input && !d->status)) {
        ccsx_set_error("ccsx_drafts: missing arrays, or sized for another batch (ccsx_draft_layout)"); return -1;
    }
    if (!input && d->win_bounds && (!d->win_off || d->win_capacity < S.wb_off[n])) { ccsx_set_error("ccsx_drafts: win_bounds needs win_off and ccsx_draft_layout's capacity"); return -1; }
    if (input) for (int z = 0; z <= n; ++z) if (d->seq_off[z] != S.seq_off[z]) { ccsx_set_error("ccsx_polish_batch: drafts.seq_off is not the capacity layout of ccsx_draft_layout"); return -1; }
    return 0;
}

// one batch through the handle's pipeline: the fused path (ccsx_submit), the draft seam or the polish seam
static int submit_impl(ccsx_handle h, const ccsx_batch *b, ccsx_results *res, ccsx_ticket *ticket, int mode, ccsx_drafts *dr_out, const ccsx_drafts *dr_in, uint32_t flags)
{
    if (!h || !b || !ticket || (mode != CCSX_RUN_DRAFT && !res) || (mode == CCSX_RUN_DRAFT && !dr_out) || (mode == CCSX_RUN_POLISH && !dr_in)) { ccsx_set_error("ccsx_submit: null argument"); return -1; }
#ifdef CCSX_FAULT_INJECTION                                          // test builds only (tests/test_cli_bam.py builds its own copy of the library)
    if (const char *e = std::getenv("CCSX_TEST_FAIL_SUBMIT"))        // fault injection for the driver's error-path test
        if (std::atoll(e) == (long long)h->next_ticket) { ++h->next_ticket; ccsx_set_error("injected failure (CCSX_TEST_FAIL_SUBMIT)"); return -2; }
#endif
    if (h->poisoned) { ccsx_set_error("ccsx_submit: an earlier submit failed after work had been enqueued; destroy the handle"); return -2; }
    HIPTRY(hipSetDevice(h->device));
    if (int rc0 = rebase_epoch(h)) return rc0;
    Slot &S = h->slot[h->next_ticket % CCSX_SLOTS];
    if (S.inflight) {                                    // the slot's previous batch was never waited for: finish it first — INCLUDING the host-side part of its
        HIPTRY(hipEventSynchronize(S.ev_done));          // outputs (offsets, the drafts' backbone words): the caller's buffers are complete whether or not it ever waits
        S.inflight = false;
        finish_outputs(S);
    }
    // A failure after the first enqueue must not leave copies or kernels running on a slot the next submit would rewrite
    // (ADVICE r02): drain every stream, mark the slot unusable and refuse further batches on this handle.
    auto fail = [&](int rc_) {
        const std::string msg = ccsx_last_error();
        (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->s_draft); if (h->s_aux) (void)hipStreamSynchronize(h->s_aux); (void)hipStreamSynchronize(h->s_comp); (void)hipStreamSynchronize(h->s_out);
        (void)hipGetLastError();
        S.staged = false; S.ran = false; S.inflight = false; S.ticket = -1;
        h->poisoned = true;
        ccsx_set_error(msg);
        return rc_;
    };
    int rc = stage(h, S, b, h->s_in);
    if (rc) return rc == -1 ? (S.staged = false, rc) : fail(rc);     // -1: rejected by validation before anything was enqueued
    if (mode != CCSX_RUN_DRAFT && (rc = check_results(S, res, S.P.out_kin != nullptr))) { (void)hipStreamSynchronize(h->s_in); S.staged = false; return rc; }
    if (mode != CCSX_RUN_FUSED && (rc = check_drafts(S, mode == CCSX_RUN_DRAFT ? dr_out : dr_in, mode == CCSX_RUN_POLISH))) { (void)hipStreamSynchronize(h->s_in); S.staged = false; return rc; }
    S.mode = mode;
    S.P.qv_only = (mode == CCSX_RUN_POLISH && (flags & CCSX_QV_ONLY)) ? 1 : 0;
#define HIPTRY_F(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ccsx_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return fail(-2); } } while (0)
    if (mode == CCSX_RUN_POLISH) {                       // the caller's drafts: bases into the slot's draft buffer, lengths + orientation references beside them
        const int n = S.P.n_zmw;
        if (S.d_din_len.reserve((size_t)n * 4) || S.d_din_bb.reserve((size_t)n * 4)) return fail(-2);
        HIPTRY_F(hipMemcpyAsync(S.d_draft.p, dr_in->seq, (size_t)S.seq_off[n], hipMemcpyHostToDevice, h->s_in));
        HIPTRY_F(hipMemcpyAsync(S.d_din_len.p, dr_in->len, (size_t)n * 4, hipMemcpyHostToDevice, h->s_in));
        HIPTRY_F(hipMemcpyAsync(S.d_din_bb.p, dr_in->backbone, (size_t)n * 4, hipMemcpyHostToDevice, h->s_in));
        S.P.din_len = (const int32_t *)S.d_din_len.p; S.P.din_bb = (const int32_t *)S.d_din_bb.p;
        S.P.opts.no_fallback_draft = 1;                   // (this slot's copy of the options: the alignment's outcome on a given draft is final)
    }
    HIPTRY_F(hipEventRecord(S.ev_up, h->s_in));
    HIPTRY_F(hipStreamWaitEvent(h->s_draft, S.ev_up, 0));
    if ((rc = launch(h, S))) return fail(rc);
    // ev[5] (end of the last kernel) doubles as the "results ready" event of the download stream
    HIPTRY_F(hipStreamWaitEvent(h->s_out, S.ev[5], 0));
    if (mode == CCSX_RUN_DRAFT) {
        const int n = S.P.n_zmw;
        const KParams &P = S.P;
        HIPTRY_F(hipMemcpyAsync(dr_out->status, P.zstat, (size_t)n * 4, hipMemcpyDeviceToHost, h->s_out));
        HIPTRY_F(hipMemcpyAsync(dr_out->len, P.draft_len, (size_t)n * 4, hipMemcpyDeviceToHost, h->s_out));
        HIPTRY_F(hipMemcpyAsync(dr_out->backbone, P.zref, (size_t)n * 4, hipMemcpyDeviceToHost, h->s_out));   // (low byte: masked in ccsx_wait)
        HIPTRY_F(hipMemcpyAsync(dr_out->seq, P.draft, (size_t)S.seq_off[n], hipMemcpyDeviceToHost, h->s_out));
        if (dr_out->n_windows) HIPTRY_F(hipMemcpyAsync(dr_out->n_windows, P.nwin, (size_t)n * 4, hipMemcpyDeviceToHost, h->s_out));
        if (dr_out->win_bounds) HIPTRY_F(hipMemcpyAsync(dr_out->win_bounds, P.wbounds, (size_t)S.wb_off[n] * 4, hipMemcpyDeviceToHost, h->s_out));
    } else if ((rc = enqueue_download(S, res, h->s_out))) return fail(rc);
    HIPTRY_F(hipEventRecord(S.ev_done, h->s_out));
#undef HIPTRY_F
    S.res = mode == CCSX_RUN_DRAFT ? nullptr : res; S.drafts_out = mode == CCSX_RUN_DRAFT ? dr_out : nullptr; S.inflight = true; S.ticket = h->next_ticket;
    h->last = (int)(h->next_ticket % CCSX_SLOTS);
    *ticket = h->next_ticket++;
    return 0;
}

int ccsx_submit(ccsx_handle h, const ccsx_batch *b, ccsx_results *res, ccsx_ticket *ticket)
{
    return submit_impl(h, b, res, ticket, CCSX_RUN_FUSED, nullptr, nullptr, 0);
}
int ccsx_submit_draft(ccsx_handle h, const ccsx_batch *b, ccsx_drafts *drafts, ccsx_ticket *ticket)
{
    return submit_impl(h, b, nullptr, ticket, CCSX_RUN_DRAFT, drafts, nullptr, 0);
}
int ccsx_submit_polish(ccsx_handle h, const ccsx_batch *b, const ccsx_drafts *drafts, ccsx_results *res, uint32_t flags, ccsx_ticket *ticket)
{
    return submit_impl(h, b, res, ticket, CCSX_RUN_POLISH, nullptr, drafts, flags);
}
int ccsx_draft_batch(ccsx_handle h, const ccsx_batch *b, ccsx_drafts *drafts)
{
    ccsx_ticket t;
    if (int rc = ccsx_submit_draft(h, b, drafts, &t)) return rc;
    return ccsx_wait(h, t);
}
int ccsx_polish_batch(ccsx_handle h, const ccsx_batch *b, const ccsx_drafts *drafts, ccsx_results *res, uint32_t flags)
{
    ccsx_ticket t;
    if (int rc = ccsx_submit_polish(h, b, drafts, res, flags, &t)) return rc;
    return ccsx_wait(h, t);
}

static Slot *slot_of(ccsx_handle h, ccsx_ticket t)
{
    if (!h || t < 0) return nullptr;
    Slot &S = h->slot[t % CCSX_SLOTS];
    return S.ticket == t ? &S : nullptr;
}

static int slot_timings(ccsx_handle h, Slot &S, ccsx_timings *t);

int ccsx_wait(ccsx_handle h, ccsx_ticket ticket)
{
    Slot *S = slot_of(h, ticket);
    if (!S) { ccsx_set_error("ccsx_wait: unknown or recycled ticket"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    if (S->inflight) {
        HIPTRY(hipEventSynchronize(S->ev_done));
        S->inflight = false;
        // (the ticket's timings are taken NOW, as doubles against the current origin — a later move of the origin (rebase_epoch) cannot invalidate them,
        // and a caller may read them any time before the slot is reused)
        S->tm_ok = S->ran && slot_timings(h, *S, &S->tm) == 0;
        finish_outputs(*S);
    }
    return 0;
}

int ccsx_poll(ccsx_handle h, ccsx_ticket ticket)
{
    Slot *S = slot_of(h, ticket);
    if (!S) { ccsx_set_error("ccsx_poll: unknown or recycled ticket"); return -1; }
    if (!S->inflight) return 1;
    const hipError_t e = hipEventQuery(S->ev_done);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    ccsx_set_error(std::string("ccsx_poll: ") + hipGetErrorString(e));
    return -2;
}

// stage durations of a slot's last run.  draft_ms = the first POA pass, align_ms = alignment cascade + accounting + the whole
// fallback round (POA included), polish_ms from the polish stage's own start event (it may have queued behind the previous batch)
static int slot_timings(ccsx_handle h, Slot &S, ccsx_timings *t)
{
    HIPTRY(hipEventSynchronize(S.ev[5]));
    std::memset(t, 0, sizeof(*t));
    HIPTRY(hipEventElapsedTime(&t->setup_ms, S.ev[0], S.ev[1]));
    HIPTRY(hipEventElapsedTime(&t->draft_ms, S.ev[1], S.ev[2]));
    HIPTRY(hipEventElapsedTime(&t->align_ms, S.ev[2], S.ev[3]));
    HIPTRY(hipEventElapsedTime(&t->queue_ms, S.ev[3], S.ev[6]));
    HIPTRY(hipEventElapsedTime(&t->polish_ms, S.ev[6], S.ev[4]));
    HIPTRY(hipEventElapsedTime(&t->stitch_ms, S.ev[4], S.ev[5]));
    HIPTRY(hipEventElapsedTime(&t->total_ms, S.ev[0], S.ev[5]));
    float a = 0.0f, b = 0.0f;
    HIPTRY(hipEventElapsedTime(&a, h->ev_epoch, S.ev[0]));
    HIPTRY(hipEventElapsedTime(&b, h->ev_epoch, S.ev[5]));
    t->start_ms = h->epoch_ms + (double)a; t->end_ms = h->epoch_ms + (double)b;
    h->age_ms = b;                    // (the origin moves forward at the next submit that finds the handle idle: rebase_epoch)
    return 0;
}

int ccsx_ticket_timings(ccsx_handle h, ccsx_ticket ticket, ccsx_timings *t)
{
    Slot *S = slot_of(h, ticket);
    if (!S || !t || !(S->ran || S->tm_ok)) { ccsx_set_error("ccsx_ticket_timings: unknown ticket"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    if (S->tm_ok) *t = S->tm;                         // taken when the ticket was waited for
    else if (int rc = slot_timings(h, *S, t)) return rc;
    t->polish_workgroups = 0;
    if (!S->inflight && S->res && S->res->n_windows) for (int z = 0; z < S->P.n_zmw; ++z) t->polish_workgroups += S->res->n_windows[z];
    return 0;
}

// ---- synchronous form (slot 0): parity tests, stage read-backs, the resident-input leg of the benchmark ----
int ccsx_upload(ccsx_handle h, const ccsx_batch *b)
{
    if (!h) { ccsx_set_error("ccsx_upload: null handle"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    for (auto &S : h->slot) if (S.inflight) { HIPTRY(hipEventSynchronize(S.ev_done)); S.inflight = false; }
    Slot &S = h->slot[0];
    const int rc = stage(h, S, b, h->s_comp);
    if (rc) { S.staged = false; return rc; }
    S.mode = CCSX_RUN_FUSED; S.drafts_out = nullptr;
    HIPTRY(hipStreamSynchronize(h->s_comp));
    h->last = 0;
    return 0;
}

int ccsx_run(ccsx_handle h)
{
    if (!h || !h->slot[0].staged) { ccsx_set_error("ccsx_run: no batch uploaded"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    h->last = 0;
    return launch(h, h->slot[0]);
}

int ccsx_sync(ccsx_handle h)
{
    if (!h) return -1;
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->s_draft));
    HIPTRY(hipStreamSynchronize(h->s_comp));
    Slot &S = h->slot[h->last];
#ifdef CCSX_PROFILE_PHASES
    if (S.staged && S.ran) {
        unsigned long long ph[16];
        HIPTRY(hipMemcpy(ph, S.P.phase, sizeof(ph), hipMemcpyDeviceToHost));
        static const char *nm[8] = {"prologue", "tables+lanes", "chunk plan", "fill", "score", "select/apply", "qv+store", "validity+list"};
        unsigned long long tot = 0;
        for (int i = 0; i < 8; ++i) tot += ph[i];
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, "[ccsx phase] %-14s %6.2f %%  (%llu cycles)\n", nm[i], tot ? 100.0 * ph[i] / tot : 0.0, ph[i]);
        static const char *pn[6] = {"thr setup+read", "thr traceback", "thr ids+zero", "thr records", "thr order", "thr col records"};
        unsigned long long pt = 0;
        for (int i = 8; i < 14; ++i) pt += ph[i];
        for (int i = 8; i < 14; ++i) std::fprintf(stderr, "[ccsx phase] %-14s %6.2f %%  (%llu cycles)\n", pn[i - 8], pt ? 100.0 * ph[i] / pt : 0.0, ph[i]);
    }
#endif
#ifdef CCSX_DEBUG_CHECKS
    if (S.staged) {
        int32_t dbg[2] = {0, 0};
        HIPTRY(hipMemcpy(dbg, S.P.debug, 8, hipMemcpyDeviceToHost));
        if (dbg[0]) { ccsx_set_error("device bounds check failed: code " + std::to_string(dbg[0]) + " at kernel line " + std::to_string(dbg[1])); return -3; }
    }
#endif
    (void)S;
    return 0;
}

int ccsx_download(ccsx_handle h, ccsx_results *res)
{
    if (!h || !h->slot[0].ran || !res) { ccsx_set_error("ccsx_download: nothing to download"); return -1; }
    Slot &S = h->slot[0];
    int rc = check_results(S, res, S.P.out_kin != nullptr);
    if (rc) return rc;
    HIPTRY(hipSetDevice(h->device));
    if ((rc = enqueue_download(S, res, h->s_comp))) return rc;
    HIPTRY(hipStreamSynchronize(h->s_comp));
    if (res->seq_off) std::memcpy(res->seq_off, S.seq_off.p, (size_t)(S.P.n_zmw + 1) * 8);
    return 0;
}

int ccsx_consensus_batch(ccsx_handle h, const ccsx_batch *b, ccsx_results *res)
{
    int rc;
    if ((rc = ccsx_upload(h, b))) return rc;
    if ((rc = ccsx_run(h))) return rc;
    if ((rc = ccsx_sync(h))) return rc;
    return ccsx_download(h, res);
}

int ccsx_get_timings(ccsx_handle h, ccsx_timings *t)
{
    if (!h || !t) { ccsx_set_error("ccsx_get_timings: no completed run"); return -1; }
    Slot &S = h->slot[h->last];
    if (!S.ran) { ccsx_set_error("ccsx_get_timings: no completed run"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    if (int rc = slot_timings(h, S, t)) return rc;
    std::vector<int32_t> nwin(S.P.n_zmw);
    HIPTRY(hipMemcpy(nwin.data(), S.P.out_nwin, nwin.size() * 4, hipMemcpyDeviceToHost));
    int64_t tw = 0;
    for (int v : nwin) tw += v;
    t->polish_workgroups = tw;
    return 0;
}

// ---- stage access (parity tests; slot of the last synchronous run) ----
int ccsx_stage_draft(ccsx_handle h, int32_t z, uint8_t *draft, int32_t cap, int32_t *len)
{
    if (!h || !h->slot[0].ran || z < 0 || z >= h->slot[0].P.n_zmw) { ccsx_set_error("ccsx_stage_draft: bad state/index"); return -1; }
    Slot &S = h->slot[0];
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->s_comp));
    int32_t L = 0;
    HIPTRY(hipMemcpy(&L, S.P.draft_len + z, 4, hipMemcpyDeviceToHost));
    if (L > cap) { ccsx_set_error("ccsx_stage_draft: buffer too small"); return -1; }
    if (L > 0) HIPTRY(hipMemcpy(draft, S.P.draft + S.seq_off[z], (size_t)L, hipMemcpyDeviceToHost));
    *len = L;
    return 0;
}

int ccsx_stage_windows(ccsx_handle h, int32_t z, int32_t *bounds, int32_t cap, int32_t *n_windows)
{
    if (!h || !h->slot[0].ran || z < 0 || z >= h->slot[0].P.n_zmw) { ccsx_set_error("ccsx_stage_windows: bad state/index"); return -1; }
    Slot &S = h->slot[0];
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->s_comp));
    int32_t nw = 0, L = 0;
    HIPTRY(hipMemcpy(&L, S.P.draft_len + z, 4, hipMemcpyDeviceToHost));
    // nwin may have been zeroed by a later status; recompute the count from the bounds array
    std::vector<int32_t> wb(S.wb_off[z + 1] - S.wb_off[z]);
    HIPTRY(hipMemcpy(wb.data(), S.P.wbounds + S.wb_off[z], wb.size() * 4, hipMemcpyDeviceToHost));
    if (L > 0) while (nw + 1 < (int)wb.size() && wb[nw] < L) ++nw;
    if (nw + 1 > cap) { ccsx_set_error("ccsx_stage_windows: buffer too small"); return -1; }
    for (int k = 0; k <= nw; ++k) bounds[k] = wb[k];
    *n_windows = nw;
    return 0;
}

// rstart[] holds the entry row for every window-edge column and -1 elsewhere (the kernel never
// materialises the other columns).
int ccsx_stage_align(ccsx_handle h, int32_t r, int32_t *rstart, int32_t cap, int32_t *valid, int32_t *score)
{
    if (!h || !h->slot[0].ran || r < 0 || r >= h->slot[0].P.n_reads) { ccsx_set_error("ccsx_stage_align: bad state/index"); return -1; }
    Slot &S = h->slot[0];
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->s_comp));
    const int n = S.P.n_zmw;
    int z = (int)(std::upper_bound(S.read_off.p, S.read_off.p + n + 1, r) - S.read_off.p) - 1;
    uint8_t v = 0; int32_t sc = 0, L = 0;
    HIPTRY(hipMemcpy(&v, S.P.avalid + r, 1, hipMemcpyDeviceToHost));
    HIPTRY(hipMemcpy(&sc, S.P.ascore + r, 4, hipMemcpyDeviceToHost));
    HIPTRY(hipMemcpy(&L, S.P.draft_len + z, 4, hipMemcpyDeviceToHost));
    *valid = v; *score = sc;
    if (L + 1 > cap) { ccsx_set_error("ccsx_stage_align: buffer too small"); return -1; }
    for (int j = 0; j <= L; ++j) rstart[j] = -1;
    if (v) {
        std::vector<int32_t> wb(S.wb_off[z + 1] - S.wb_off[z]);
        HIPTRY(hipMemcpy(wb.data(), S.P.wbounds + S.wb_off[z], wb.size() * 4, hipMemcpyDeviceToHost));
        int nw = 0;
        while (nw + 1 < (int)wb.size() && wb[nw] < L) ++nw;
        std::vector<int32_t> ent(2 * nw);
        HIPTRY(hipMemcpy(ent.data(), S.P.ent + S.ent_off[r], ent.size() * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < 2 * nw; ++k) {
            int col = (k == 0) ? 0 : (k == 2 * nw - 1) ? L : wb[(k + 1) >> 1] + ((k & 1) ? -CCSX_WIN_OVERHANG : CCSX_WIN_OVERHANG);
            rstart[col] = ent[k];
        }
    }
    return 0;
}

}  // extern "C"
