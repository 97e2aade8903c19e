// ccsx_kernels.h — kernel parameter block shared by ccsx_kernels.hip and ccsx_api.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "ccsx.h"

// bytes of POA scratch per vertex (ccsx_kernels.hip poa_slot, all by topological position): the 32-row score column of far-read columns 128, five 16-byte
// records (kinfo, column records x 2, overflow in-edges x 2), 16 move bytes (a nibble per band row), 3 words of consensus / band state, two flag bytes
#define CCSX_POA_BYTES_PER_VERTEX 238

// the floor of every reported per-base error probability for opts.max_qv (SPEC v7: Q50 = exactly 1e-5f; the oracle computes the same expression)
static inline float ccsx_perr_floor(int max_qv) { return max_qv == 50 ? 1e-5f : (float)pow(10.0, -(double)max_qv / 10.0); }

// words of stored moves per pass of a k_align16 quad: per block of 16 draft columns 16 move words (one per band row) + 2 words of band steps / edge flags = 18,
// rounded up so that a pass's words start on a 16-byte boundary.  ONE definition for the kernels (tb_stride) and the host's slot size.
#define CCSX_TB_WORDS_PER_BLOCK 18
static inline __host__ __device__ int ccsx_tb_blocks(int Ld) { return (Ld + 15) >> 4; }
static inline __host__ __device__ int ccsx_tb_stride(int Ld) { return (ccsx_tb_blocks(Ld) * CCSX_TB_WORDS_PER_BLOCK + 3) & ~3; }

struct KParams {
    int32_t n_zmw, n_reads;
    int32_t maxL_max;          // longest subread of the batch
    int32_t vcap_max;          // POA vertex capacity of the largest ZMW
    int32_t need_max;          // max window-edge columns per read (2 * window slots)
    int32_t max_reads;         // most passes of a ZMW in the batch (after the top_passes cap)
    int32_t qv_only;           // CCSX_QV_ONLY (ccsx_polish_batch): one scoring round, no mutation applied — QVs / rq of the sequence as given
    ccsx_opts opts;
    float perr_floor;          // 10^(-opts.max_qv / 10): floor of every reported per-base error probability (ccsx_perr_floor)
    const ccsx_model *model;   // device copy
    // ---- inputs (HBM resident after ccsx_upload)
    const float *snr;
    const int32_t *read_off;
    const int64_t *base_off;
    const uint8_t *bases, *pw, *flags;
    // ---- host-derived layout
    const int32_t *read_zmw;   // [R] owning ZMW of each read
    const int32_t *vcap;       // [n] POA vertex capacity
    const int32_t *dcap;       // [n] draft / consensus capacity
    const int64_t *seq_off;    // [n+1] offsets of draft and outputs (capacity layout)
    const int32_t *wb_off;     // [n+1] offsets into wbounds; window slots of z = wb_off[z+1]-wb_off[z]-1
    const int64_t *ent_off;    // [R] offsets into ent
    int32_t *wslot_zmw;        // [total window slots] compact map built on the device after the draft stage (k_wmap): owning ZMW of the i-th WINDOW
    int32_t *wstart;           // [n_zmw + 1] first compact index of every ZMW's windows; [n_zmw] = windows of the batch
    // ---- per-ZMW state
    float *tabME, *tabINS, *tabDL;
    float *tabZ;               // [n][32] z-score parameters per context: MU[16], VAR[16]
    uint8_t *draft;
    int32_t *draft_len, *nwin, *zstat, *nreads_used, *wbounds, *np;
    int32_t *nfull;            // [n] full-length passes among nreads_used (the partial passes follow them)
    int32_t *zref;             // [n] backbone pass (index within the ZMW) = orientation reference of draft and consensus; bit 8: a fallback
                               //     draft is requested (set by k_post), bit 9: the fallback draft has been made
    int32_t *ticket_poa, *ticket_align;   // 256-byte scratch block that also holds debug[] and phase[] (no tickets since the chunked launch)
    int32_t *debug;            // [4] first failed bounds check (CCSX_DEBUG_CHECKS builds)
    unsigned long long *phase; // [16] per-phase cycle sums (CCSX_PROFILE_PHASES builds)
    // ---- POA / alignment scratch (per resident slot)
    uint8_t *poa_scratch;
    size_t poa_slot_bytes;
    int32_t poa_slots;
    int32_t *align_scratch;
    size_t align_slot_i32;
    int32_t align_slots;
    uint8_t *avalid;
    int32_t *ascore;
    int32_t *ent;              // entry rows of every window-edge column, per read
    uint32_t *dmask;           // same indexing: pile-up dirty bits of the draft positions between two window-edge columns
    // ---- per-window polish outputs
    long long total_wslots;
    int32_t pw_obs_bytes, pw_gb_floats;   // k_polish dynamic LDS: observation codes, gamma/beta floats
    uint8_t *wseq;             // [wslots][32]
    float *wqv;                // [wslots][32]
    float *wsum;               // [wslots] sum of p_err over the core
    int4 *wmeta;               // [wslots] (core length, usable reads, non-convergent, iterations)
    // ---- results
    uint8_t *out_seq, *out_qual;
    float *out_raw;
    int32_t *out_status, *out_len, *out_iters, *out_nwin;
    float *out_rq, *out_ec;
    int32_t *out_fn, *out_rn;  // passes used per strand
    // ---- launch order (cost-sorted, longest first: the one-wave-per-ZMW / per-read kernels finish together)
    const int32_t *zmw_perm;   // [n_zmw]
    const int32_t *read_perm;  // [n_reads]
    const int32_t *quads;      // [n_quads] (first pass << 2 | passes - 1): up to four consecutive passes of one ZMW, longest first
    int32_t n_quads;
    int32_t *align_retry;      // [16 + n_reads]: [0] = count, [16..] = passes for the 64-row retry (appended by k_align16)
    // ---- HiFi kinetics (NULL unless opts.hifi_kinetics); kept at the end so the hot kernels' kernarg offsets do not move
    const uint8_t *ipd;
    uint8_t *wtpl;             // [wslots][32] converged window template incl. overhangs
    short2 *wtmeta;            // [wslots] (J, core start)
    uchar4 *wkin;              // [wslots][32] (fi, fp, ri, rp) codes of the core positions
    uint8_t *out_kin;          // 4 planes (fi, fp, ri, rp) of seq_off[n] bytes each
    long long kin_plane;       // plane stride = seq_off[n]
    // ---- caller-supplied drafts (ccsx_polish_batch: the polish seam of docs/img/ccs-impl.png); the bases are already in `draft`
    const int32_t *din_len;    // [n] draft length (0 = none)
    const int32_t *din_bb;     // [n] a pass of the ZMW that has the draft's orientation
    // ---- k_align16 / k_align16_tb: a quad's stored moves (4 passes x ccsx_tb_stride(draft length) words + the 4 final band starts, see below); align_slot_i32 / align_slots are the 64-row retry's and the split alignment's
    size_t align16_slot_i32;
    int32_t align16_slots;
    int32_t *retry_scratch;    // the 64-row retry's / split alignment's slots: behind k_align16's in the same buffer (the trace-backs run beside the next launch / the retry)
    int32_t align16_regions;   // 1, or 2 when the batch's quads take several launches: launch c uses region c & 1, so its trace-back runs under launch c + 1
};

// which stages ccsx_launch_all enqueues: the fused path, the draft stage alone (ccsx_draft_batch), or alignment cascade + polish on caller-supplied drafts
enum { CCSX_RUN_FUSED = 0, CCSX_RUN_DRAFT = 1, CCSX_RUN_POLISH = 2 };

const char *ccsx_launch_all(const KParams &P, hipStream_t st_draft, hipStream_t st_polish, hipEvent_t *ev /* [7] */, int mode = CCSX_RUN_FUSED,
                            hipStream_t st_aux = nullptr, hipEvent_t *ev_aux /* [7] */ = nullptr);   // NULL, or the name of the launch that failed; st_aux: second stream of the POA stage (half-batches)
int ccsx_kernel_is_experiment();         // built with -DCCSX_EXPERIMENT (timing studies: wrong results)
const char *ccsx_kernel_build_flags();   // "" for a product build; the experiment switches this translation unit was compiled with otherwise
int ccsx_polish_lds(int max_reads, int *obs_bytes, int *gb_floats);
