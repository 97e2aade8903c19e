// ccsx_api.cpp — C ABI over the HIP kernels: handle lifecycle, HBM layout, upload / run / download.
// One handle = one GPU + one HIP stream.  There is no CPU fallback: without a usable device every
// entry point fails with a message (ccsx_last_error).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
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

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { ccsx_set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); return -2; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct ccsx_handle_s {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    ccsx_model model;
    ccsx_opts opts;
    DevBuf d_model;
    // inputs
    DevBuf d_snr, d_read_off, d_base_off, d_bases, d_pw, d_ipd, d_flags;
    // layout
    DevBuf d_read_zmw, d_vcap, d_dcap, d_seq_off, d_wb_off, d_ent_off, d_wslot, d_zperm, d_rperm;
    // state
    DevBuf d_tabME, d_tabINS, d_tabDL, d_tabZ, d_dmask, d_draft, d_zmw_i32 /* 6 x n int32 */, d_wbounds, d_ticket;
    DevBuf d_poa, d_align, d_avalid, d_ascore, d_ent;
    DevBuf d_wseq, d_wqv, d_wsum, d_wmeta;
    DevBuf d_out_seq, d_out_qual, d_out_raw, d_out_i32 /* 6 x n */, d_out_f32 /* 2 x n */;
    DevBuf d_wtpl, d_wtmeta, d_wkin, d_out_kin;   // HiFi kinetics only
    // host copies of the layout
    std::vector<int64_t> seq_off, ent_off;
    std::vector<int32_t> wb_off, read_off;
    std::vector<int64_t> base_off;
    KParams P;
    bool uploaded = false, ran = false;
    size_t free_mem = 0, total_mem = 0;
};

extern "C" {

int ccsx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

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
    HIPTRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto &ev : h->ev) HIPTRY(hipEventCreate(&ev));
    if (h->d_model.reserve(sizeof(ccsx_model))) return -2;
    HIPTRY(hipMemcpy(h->d_model.p, &h->model, sizeof(ccsx_model), hipMemcpyHostToDevice));
    HIPTRY(hipMemGetInfo(&h->free_mem, &h->total_mem));
    *out = h;
    return 0;
}

int ccsx_destroy(ccsx_handle h)
{
    if (!h) return -1;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->d_model, &h->d_snr, &h->d_read_off, &h->d_base_off, &h->d_bases, &h->d_pw, &h->d_flags, &h->d_read_zmw,
                      &h->d_vcap, &h->d_dcap, &h->d_seq_off, &h->d_wb_off, &h->d_ent_off, &h->d_wslot, &h->d_tabME, &h->d_tabINS, &h->d_tabDL, &h->d_tabZ, &h->d_dmask,
                      &h->d_draft, &h->d_zmw_i32, &h->d_wbounds, &h->d_ticket, &h->d_poa, &h->d_align, &h->d_avalid, &h->d_ascore,
                      &h->d_ent, &h->d_wseq, &h->d_wqv, &h->d_wsum, &h->d_wmeta, &h->d_out_seq, &h->d_out_qual, &h->d_out_raw,
                      &h->d_out_i32, &h->d_out_f32, &h->d_zperm, &h->d_rperm, &h->d_ipd, &h->d_wtpl, &h->d_wtmeta, &h->d_wkin, &h->d_out_kin};
    for (auto *b : bufs) b->release();
    for (auto &ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
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
    return 0;
}

int ccsx_upload(ccsx_handle h, const ccsx_batch *b)
{
    if (!h) { ccsx_set_error("ccsx_upload: null handle"); return -1; }
    if (validate(b)) return -1;
    const bool kin = h->opts.hifi_kinetics != 0;
    if (kin && !b->ipd) { ccsx_set_error("ccsx_upload: opts.hifi_kinetics needs batch.ipd"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    const int n = b->n_zmw, R = b->n_reads;
    const int64_t NB = b->n_bases;
    // ---- host-derived layout
    std::vector<int32_t> read_zmw(R), vcap(n), dcap(n);
    h->seq_off.assign(n + 1, 0); h->wb_off.assign(n + 1, 0); h->ent_off.assign(R + 1, 0);
    int64_t maxL_max = 1, vcap_max = 1; int need_max = 2, nr_max = 1;
    for (int z = 0; z < n; ++z) {
        int64_t maxL = 0;
        int nr = b->read_off[z + 1] - b->read_off[z];
        { const int top = (h->opts.top_passes <= 0 || h->opts.top_passes > 64) ? 64 : h->opts.top_passes; if (nr > top) nr = top; }
        nr_max = std::max(nr_max, nr);
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) {
            read_zmw[r] = z;
            const int64_t L = b->base_off[r + 1] - b->base_off[r];
            if (L > maxL) maxL = L;
        }
        dcap[z] = (int32_t)ccsx_draft_cap(maxL);
        vcap[z] = (int32_t)ccsx_vertex_cap(maxL);
        const int wcap = dcap[z] / (CCSX_WIN_CORE - 3) + 4;   // cores are 19..25 columns (SPEC windows)
        h->seq_off[z + 1] = h->seq_off[z] + dcap[z];
        h->wb_off[z + 1] = h->wb_off[z] + wcap;
        for (int r = b->read_off[z]; r < b->read_off[z + 1]; ++r) h->ent_off[r + 1] = h->ent_off[r] + 2 * (wcap - 1);
        maxL_max = std::max(maxL_max, maxL); vcap_max = std::max<int64_t>(vcap_max, vcap[z]); need_max = std::max(need_max, 2 * (wcap - 1));
    }
    if (maxL_max > 65535) { ccsx_set_error("ccsx_upload: subreads longer than 65535 bases are not supported"); return -1; }
    h->read_off.assign(b->read_off, b->read_off + n + 1);
    h->base_off.assign(b->base_off, b->base_off + R + 1);
    // launch order: longest first (stable within 256-base classes, so a uniform batch keeps its input order and its
    // locality).  Mixed batches (BASELINE configs[4]: 1-25 kb, 3-50 passes) otherwise end on a few long stragglers.
    std::vector<int32_t> zperm(n), rperm(R > 0 ? R : 1);
    {
        auto order_by = [](std::vector<int32_t> &perm, const std::vector<int64_t> &len) {
            const int NB = 258;
            std::vector<int32_t> cnt(NB + 1, 0);
            auto cls = [&](int64_t l) { int c = (int)(l >> 8); return NB - 1 - (c > NB - 1 ? NB - 1 : c); };   // descending length
            for (int64_t l : len) ++cnt[cls(l) + 1];
            for (int i = 0; i < NB; ++i) cnt[i + 1] += cnt[i];
            for (size_t i = 0; i < len.size(); ++i) perm[cnt[cls(len[i])]++] = (int32_t)i;
        };
        std::vector<int64_t> zl(n), rl(R);
        for (int z = 0; z < n; ++z) zl[z] = dcap[z];
        for (int r = 0; r < R; ++r) rl[r] = b->base_off[r + 1] - b->base_off[r];
        order_by(zperm, zl);
        if (R > 0) order_by(rperm, rl);
    }
    const int64_t total_wslots = (int64_t)h->wb_off[n] - n;
    std::vector<int32_t> wslot(total_wslots > 0 ? total_wslots : 1);
    for (int z = 0; z < n; ++z) std::fill(wslot.begin() + (h->wb_off[z] - z), wslot.begin() + (h->wb_off[z + 1] - (z + 1)), z);

#define UP(buf, src, bytes)                                                                                    \
    do {                                                                                                       \
        if ((buf).reserve(bytes)) return -2;                                                                   \
        HIPTRY(hipMemcpyAsync((buf).p, (src), (bytes), hipMemcpyHostToDevice, h->stream));                      \
    } while (0)
    UP(h->d_snr, b->snr, (size_t)n * 16);
    UP(h->d_read_off, b->read_off, (size_t)(n + 1) * 4);
    UP(h->d_base_off, b->base_off, (size_t)(R + 1) * 8);
    UP(h->d_bases, b->bases, (size_t)NB);
    UP(h->d_pw, b->pw, (size_t)NB);
    if (kin) UP(h->d_ipd, b->ipd, (size_t)NB);
    UP(h->d_flags, b->flags, (size_t)R);
    UP(h->d_read_zmw, read_zmw.data(), (size_t)R * 4);
    UP(h->d_vcap, vcap.data(), (size_t)n * 4);
    UP(h->d_dcap, dcap.data(), (size_t)n * 4);
    UP(h->d_seq_off, h->seq_off.data(), (size_t)(n + 1) * 8);
    UP(h->d_wb_off, h->wb_off.data(), (size_t)(n + 1) * 4);
    UP(h->d_ent_off, h->ent_off.data(), (size_t)(R + 1) * 8);
    UP(h->d_wslot, wslot.data(), wslot.size() * 4);
    UP(h->d_zperm, zperm.data(), zperm.size() * 4);
    UP(h->d_rperm, rperm.data(), rperm.size() * 4);
#undef UP
    HIPTRY(hipStreamSynchronize(h->stream));   // host staging vectors go out of scope

    const int64_t cap_total = h->seq_off[n];
#define RES(buf, bytes) do { if ((buf).reserve(bytes)) return -2; } while (0)
    RES(h->d_tabME, (size_t)n * 192 * 4); RES(h->d_tabINS, (size_t)n * 192 * 4); RES(h->d_tabDL, (size_t)n * 16 * 4); RES(h->d_tabZ, (size_t)n * 32 * 4);
    RES(h->d_draft, (size_t)cap_total);
    RES(h->d_zmw_i32, (size_t)n * 4 * 6);
    RES(h->d_wbounds, (size_t)h->wb_off[n] * 4);
    RES(h->d_ticket, 256);
    RES(h->d_avalid, (size_t)R); RES(h->d_ascore, (size_t)R * 4);
    RES(h->d_ent, (size_t)h->ent_off[R] * 4); RES(h->d_dmask, (size_t)h->ent_off[R] * 4);
    RES(h->d_wseq, (size_t)total_wslots * 32); RES(h->d_wqv, (size_t)total_wslots * 32 * 4);
    RES(h->d_wsum, (size_t)total_wslots * 4); RES(h->d_wmeta, (size_t)total_wslots * 16);
    RES(h->d_out_seq, (size_t)cap_total); RES(h->d_out_qual, (size_t)cap_total); RES(h->d_out_raw, (size_t)cap_total * 4);
    RES(h->d_out_i32, (size_t)n * 4 * 6); RES(h->d_out_f32, (size_t)n * 4 * 2);
    if (kin) {
        RES(h->d_wtpl, (size_t)total_wslots * 32); RES(h->d_wtmeta, (size_t)total_wslots * 4);
        RES(h->d_wkin, (size_t)total_wslots * 32 * 4); RES(h->d_out_kin, (size_t)cap_total * 4);
    }

    // ---- resident POA graphs / alignment slots: as many as fit a memory budget, never more than the work
    const size_t poa_slot_bytes = (((size_t)vcap_max + 64) * 392 + (size_t)maxL_max * 4 + 1024 + 255) & ~(size_t)255;
    const size_t align_slot_i32 = (size_t)need_max * 128 + need_max + 64;   // (origin, dirty bits) per cell and edge + band starts
    size_t freeb = 0, totalb = 0;
    HIPTRY(hipMemGetInfo(&freeb, &totalb));
    freeb += h->d_poa.cap + h->d_align.cap;                      // what we already hold is reusable
    const size_t budget = freeb > (size_t)6 << 30 ? freeb - ((size_t)4 << 30) : freeb / 2;
    int poa_slots = h->opts.poa_slots > 0 ? h->opts.poa_slots : 8192;
    poa_slots = std::min(poa_slots, n);
    poa_slots = (int)std::min<size_t>((size_t)poa_slots, std::max<size_t>(1, (budget * 3 / 4) / poa_slot_bytes));
    int align_slots = std::min(16384, R);
    align_slots = (int)std::min<size_t>((size_t)align_slots, std::max<size_t>(1, (budget / 8) / (align_slot_i32 * 4)));
    RES(h->d_poa, (size_t)poa_slots * poa_slot_bytes);
    RES(h->d_align, (size_t)align_slots * align_slot_i32 * 4);
#undef RES

    KParams &P = h->P;
    std::memset(&P, 0, sizeof(P));
    P.n_zmw = n; P.n_reads = R; P.maxL_max = (int32_t)maxL_max; P.vcap_max = (int32_t)vcap_max; P.need_max = need_max;
    P.opts = h->opts;
    P.model = (const ccsx_model *)h->d_model.p;
    P.snr = (const float *)h->d_snr.p; P.read_off = (const int32_t *)h->d_read_off.p; P.base_off = (const int64_t *)h->d_base_off.p;
    P.bases = (const uint8_t *)h->d_bases.p; P.pw = (const uint8_t *)h->d_pw.p; P.flags = (const uint8_t *)h->d_flags.p;
    P.read_zmw = (const int32_t *)h->d_read_zmw.p; P.vcap = (const int32_t *)h->d_vcap.p; P.dcap = (const int32_t *)h->d_dcap.p;
    P.seq_off = (const int64_t *)h->d_seq_off.p; P.wb_off = (const int32_t *)h->d_wb_off.p; P.ent_off = (const int64_t *)h->d_ent_off.p; P.wslot_zmw = (const int32_t *)h->d_wslot.p;
    P.zmw_perm = (const int32_t *)h->d_zperm.p; P.read_perm = (const int32_t *)h->d_rperm.p;
    P.tabME = (float *)h->d_tabME.p; P.tabINS = (float *)h->d_tabINS.p; P.tabDL = (float *)h->d_tabDL.p; P.tabZ = (float *)h->d_tabZ.p;
    P.draft = (uint8_t *)h->d_draft.p;
    int32_t *zi = (int32_t *)h->d_zmw_i32.p;
    P.draft_len = zi; P.nwin = zi + n; P.zstat = zi + 2 * (size_t)n; P.nreads_used = zi + 3 * (size_t)n; P.np = zi + 4 * (size_t)n;
    P.wbounds = (int32_t *)h->d_wbounds.p;
    P.ticket_poa = (int32_t *)h->d_ticket.p; P.ticket_align = P.ticket_poa + 1; P.debug = P.ticket_poa + 4; P.phase = (unsigned long long *)(P.ticket_poa + 16);
    P.poa_scratch = (uint8_t *)h->d_poa.p; P.poa_slot_bytes = poa_slot_bytes; P.poa_slots = poa_slots;
    P.align_scratch = (int32_t *)h->d_align.p; P.align_slot_i32 = align_slot_i32; P.align_slots = align_slots;
    P.avalid = (uint8_t *)h->d_avalid.p; P.ascore = (int32_t *)h->d_ascore.p; P.ent = (int32_t *)h->d_ent.p; P.dmask = (uint32_t *)h->d_dmask.p;
    P.total_wslots = total_wslots;
    if (ccsx_polish_lds(nr_max, &P.pw_obs_bytes, &P.pw_gb_floats)) { ccsx_set_error("ccsx_upload: cannot size the polish kernel's LDS"); return -2; }
    P.wseq = (uint8_t *)h->d_wseq.p; P.wqv = (float *)h->d_wqv.p; P.wsum = (float *)h->d_wsum.p; P.wmeta = (int4 *)h->d_wmeta.p;
    P.out_seq = (uint8_t *)h->d_out_seq.p; P.out_qual = (uint8_t *)h->d_out_qual.p; P.out_raw = (float *)h->d_out_raw.p;
    int32_t *oi = (int32_t *)h->d_out_i32.p;
    P.out_status = oi; P.out_len = oi + n; P.out_iters = oi + 2 * (size_t)n; P.out_nwin = oi + 3 * (size_t)n;
    P.out_fn = oi + 4 * (size_t)n; P.out_rn = oi + 5 * (size_t)n;
    if (kin) {
        P.ipd = (const uint8_t *)h->d_ipd.p;
        P.wtpl = (uint8_t *)h->d_wtpl.p; P.wtmeta = (short2 *)h->d_wtmeta.p; P.wkin = (uchar4 *)h->d_wkin.p;
        P.out_kin = (uint8_t *)h->d_out_kin.p; P.kin_plane = cap_total;
    }
    float *of = (float *)h->d_out_f32.p;
    P.out_rq = of; P.out_ec = of + n;
    h->uploaded = true; h->ran = false;
    return 0;
}

int ccsx_run(ccsx_handle h)
{
    if (!h || !h->uploaded) { ccsx_set_error("ccsx_run: no batch uploaded"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    ccsx_launch_all(h->P, h->stream, h->ev);
    HIPTRY(hipGetLastError());
    h->ran = true;
    return 0;
}

int ccsx_sync(ccsx_handle h)
{
    if (!h) return -1;
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->stream));
#ifdef CCSX_PROFILE_PHASES
    if (h->uploaded && h->ran) {
        unsigned long long ph[16];
        HIPTRY(hipMemcpy(ph, h->P.phase, sizeof(ph), hipMemcpyDeviceToHost));
        static const char *nm[7] = {"prologue", "tables+lanes", "chunk plan", "fill", "score", "select/apply", "qv+store"};
        unsigned long long tot = 0;
        for (int i = 0; i < 7; ++i) tot += ph[i];
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, "[ccsx phase] %-14s %6.2f %%  (%llu cycles)\n", nm[i], tot ? 100.0 * ph[i] / tot : 0.0, ph[i]);
        static const char *pn[6] = {"poa load/chain", "poa dp", "poa traceback", "poa thread", "(between reads)", "poa consensus"};
        unsigned long long pt = 0;
        for (int i = 8; i < 14; ++i) pt += ph[i];
        for (int i = 8; i < 14; ++i) std::fprintf(stderr, "[ccsx phase] %-14s %6.2f %%  (%llu cycles)\n", pn[i - 8], pt ? 100.0 * ph[i] / pt : 0.0, ph[i]);
    }
#endif
#ifdef CCSX_DEBUG_CHECKS
    if (h->uploaded) {
        int32_t dbg[2] = {0, 0};
        HIPTRY(hipMemcpy(dbg, h->P.debug, 8, hipMemcpyDeviceToHost));
        if (dbg[0]) { ccsx_set_error("device bounds check failed: code " + std::to_string(dbg[0]) + " at kernel line " + std::to_string(dbg[1])); return -3; }
    }
#endif
    return 0;
}

int ccsx_download(ccsx_handle h, ccsx_results *res)
{
    if (!h || !h->ran || !res) { ccsx_set_error("ccsx_download: nothing to download"); return -1; }
    const int n = h->P.n_zmw;
    if (res->n_zmw != n || res->seq_capacity < h->seq_off[n]) { ccsx_set_error("ccsx_download: result buffers too small"); return -1; }
    if (!h->P.out_kin && (res->fi || res->fp || res->ri || res->rp)) {
        ccsx_set_error("ccsx_download: kinetics buffers given but the handle was created without opts.hifi_kinetics");
        return -1;
    }
    HIPTRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
#define DOWN(dst, src, bytes) do { if (dst) HIPTRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, s)); } while (0)
    DOWN(res->status, h->P.out_status, (size_t)n * 4);
    DOWN(res->seq_len, h->P.out_len, (size_t)n * 4);
    DOWN(res->iters, h->P.out_iters, (size_t)n * 4);
    DOWN(res->n_windows, h->P.out_nwin, (size_t)n * 4);
    DOWN(res->rq, h->P.out_rq, (size_t)n * 4);
    DOWN(res->ec, h->P.out_ec, (size_t)n * 4);
    DOWN(res->np, h->P.np, (size_t)n * 4);
    DOWN(res->seq, h->P.out_seq, (size_t)h->seq_off[n]);
    DOWN(res->qual, h->P.out_qual, (size_t)h->seq_off[n]);
    DOWN(res->raw_qv, h->P.out_raw, (size_t)h->seq_off[n] * 4);
    DOWN(res->fn, h->P.out_fn, (size_t)n * 4);
    DOWN(res->rn, h->P.out_rn, (size_t)n * 4);
    if (h->P.out_kin) {
        const size_t pl = (size_t)h->seq_off[n];
        DOWN(res->fi, h->P.out_kin, pl); DOWN(res->fp, h->P.out_kin + pl, pl);
        DOWN(res->ri, h->P.out_kin + 2 * pl, pl); DOWN(res->rp, h->P.out_kin + 3 * pl, pl);
    }
#undef DOWN
    HIPTRY(hipStreamSynchronize(s));
    if (res->seq_off) std::memcpy(res->seq_off, h->seq_off.data(), (size_t)(n + 1) * 8);
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
    if (!h || !h->ran || !t) { ccsx_set_error("ccsx_get_timings: no completed run"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipEventSynchronize(h->ev[5]));
    float ms[5];
    for (int i = 0; i < 5; ++i) HIPTRY(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    t->setup_ms = ms[0]; t->draft_ms = ms[1]; t->align_ms = ms[2]; t->polish_ms = ms[3]; t->stitch_ms = ms[4];
    HIPTRY(hipEventElapsedTime(&t->total_ms, h->ev[0], h->ev[5]));
    std::vector<int32_t> nwin(h->P.n_zmw);
    HIPTRY(hipMemcpy(nwin.data(), h->P.out_nwin, nwin.size() * 4, hipMemcpyDeviceToHost));
    int64_t tw = 0;
    for (int v : nwin) tw += v;
    t->polish_workgroups = tw;
    return 0;
}

// ---- stage access (parity tests) ----
int ccsx_stage_draft(ccsx_handle h, int32_t z, uint8_t *draft, int32_t cap, int32_t *len)
{
    if (!h || !h->ran || z < 0 || z >= h->P.n_zmw) { ccsx_set_error("ccsx_stage_draft: bad state/index"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->stream));
    int32_t L = 0;
    HIPTRY(hipMemcpy(&L, h->P.draft_len + z, 4, hipMemcpyDeviceToHost));
    if (L > cap) { ccsx_set_error("ccsx_stage_draft: buffer too small"); return -1; }
    if (L > 0) HIPTRY(hipMemcpy(draft, h->P.draft + h->seq_off[z], (size_t)L, hipMemcpyDeviceToHost));
    *len = L;
    return 0;
}

int ccsx_stage_windows(ccsx_handle h, int32_t z, int32_t *bounds, int32_t cap, int32_t *n_windows)
{
    if (!h || !h->ran || z < 0 || z >= h->P.n_zmw) { ccsx_set_error("ccsx_stage_windows: bad state/index"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->stream));
    int32_t nw = 0, L = 0;
    HIPTRY(hipMemcpy(&L, h->P.draft_len + z, 4, hipMemcpyDeviceToHost));
    // nwin may have been zeroed by a later status; recompute the count from the bounds array
    std::vector<int32_t> wb(h->wb_off[z + 1] - h->wb_off[z]);
    HIPTRY(hipMemcpy(wb.data(), h->P.wbounds + h->wb_off[z], wb.size() * 4, hipMemcpyDeviceToHost));
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
    if (!h || !h->ran || r < 0 || r >= h->P.n_reads) { ccsx_set_error("ccsx_stage_align: bad state/index"); return -1; }
    HIPTRY(hipSetDevice(h->device));
    HIPTRY(hipStreamSynchronize(h->stream));
    int z = (int)(std::upper_bound(h->read_off.begin(), h->read_off.end(), r) - h->read_off.begin()) - 1;
    uint8_t v = 0; int32_t sc = 0, L = 0;
    HIPTRY(hipMemcpy(&v, h->P.avalid + r, 1, hipMemcpyDeviceToHost));
    HIPTRY(hipMemcpy(&sc, h->P.ascore + r, 4, hipMemcpyDeviceToHost));
    HIPTRY(hipMemcpy(&L, h->P.draft_len + z, 4, hipMemcpyDeviceToHost));
    *valid = v; *score = sc;
    if (L + 1 > cap) { ccsx_set_error("ccsx_stage_align: buffer too small"); return -1; }
    for (int j = 0; j <= L; ++j) rstart[j] = -1;
    if (v) {
        std::vector<int32_t> wb(h->wb_off[z + 1] - h->wb_off[z]);
        HIPTRY(hipMemcpy(wb.data(), h->P.wbounds + h->wb_off[z], wb.size() * 4, hipMemcpyDeviceToHost));
        int nw = 0;
        while (nw + 1 < (int)wb.size() && wb[nw] < L) ++nw;
        std::vector<int32_t> ent(2 * nw);
        HIPTRY(hipMemcpy(ent.data(), h->P.ent + h->ent_off[r], ent.size() * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < 2 * nw; ++k) {
            int col = (k == 0) ? 0 : (k == 2 * nw - 1) ? L : wb[(k + 1) >> 1] + ((k & 1) ? -CCSX_WIN_OVERHANG : CCSX_WIN_OVERHANG);
            rstart[col] = ent[k];
        }
    }
    return 0;
}

}  // extern "C"
