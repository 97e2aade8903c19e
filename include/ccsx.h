/*
 * ccsx.h — C ABI of the MI355X-native CCS per-ZMW consensus hot path.
 *
 * Drop-in boundary.  The reference (PacificBiosciences/ccs, docs-only mount)
 * publishes no plugin API; the only documented seam is the block diagram docs/img/ccs-impl.png:
 * "Draft Stage {GPU | CPU pool}" -> queue(ZMWs, Drafts, Windows) -> "Polish Stage {GPU | CPU pool}".
 * The entry points below are the GPU consumers of those two queues plus the fused path:
 *
 *   ccsx_consensus_*  : steps 2,3,4,8,9,10 of docs/how-does-ccs-work.md:34-112 for a batch of ZMWs
 *   ccsx_stage_*      : per-stage access (draft / align / polish) used by the parity tests
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *   - every function returns 0 on success, <0 on a fatal error (HIP error, OOM, bad argument);
 *     text via ccsx_last_error().  Algorithmic failures are reported PER ZMW in status[]
 *     (one status per ZMW, the run continues: docs/faq/reports-aux-files.md:10-12,143-159).
 *   - the caller owns every input and output buffer; the library owns device memory and streams.
 *   - a handle is bound to one GPU and is not thread-safe: one handle per host worker per GPU
 *     (mirrors the reference's "-j" worker pool, docs/faq/parallelize.md:17).
 *   - there is NO CPU fallback: if no gfx950 device is usable, ccsx_create fails loudly.
 */
#ifndef CCSX_H
#define CCSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCSX_ABI_VERSION 6   /* v6: ccsx_opts.max_qv (was reserved), ccsx_runtime_switches() */
#define CCSX_SPEC_VERSION 8   /* DESIGN.md §2; bumped whenever a result-changing rule changes (oracle: ORC_SPEC_VERSION) */

/* ---- fixed constants of the algorithm specification ---- */
#define CCSX_BAND          64   /* DP band rows of the wide alignment (retry of the cascade, split alignment: one wave64) */
#define CCSX_POA_BAND      32   /* DP band rows of the POA (four graphs per wave64, two rows per lane)                    */
#define CCSX_MAX_PASSES    255  /* passes of a ZMW the engine uses (SPEC v5; k_polish works through them in groups of 32)               */
#define CCSX_MAXPRED       7    /* POA in-edge cap per vertex (a move is a nibble: slot * 2 + [deletion], 15 = insertion) */
#define CCSX_WIN_CORE      22   /* target window core size, docs/how-does-ccs-work.md:57-59      */
#define CCSX_WIN_OVERHANG  2    /* +-2 bp overlap, same citation                                 */
#define CCSX_JMAX          31   /* max template columns in a polish window                       */
#define CCSX_IMAX          63   /* max read bases of one subread inside a polish window          */
#define CCSX_MAX_ITER      8    /* polish iterations per window                                  */
#define CCSX_NCTX          16   /* dinucleotide contexts (prev base, cur base)                   */
#define CCSX_NOBS          12   /* emission outcomes: base(4) x pulse-width bin(3)               */

/* ---- per-ZMW status, mirrors docs/faq/reports-aux-files.md:143-159 (subset that this path can raise) ---- */
enum ccsx_status {
    CCSX_SUCCESS               = 0,
    CCSX_TOO_FEW_PASSES        = 1,  /* fewer usable subreads than opts.min_passes                    */
    CCSX_DRAFT_FAILURE         = 2,  /* POA produced no draft / vertex capacity exceeded              */
    CCSX_TOO_MANY_UNUSABLE     = 3,  /* <= 50 % of subreads map to the draft (accuracy-vs-passes.md:37-39) */
    CCSX_NON_CONVERGENT        = 4,  /* some window hit CCSX_MAX_ITER with favourable mutations left   */
    CCSX_TOO_SHORT             = 5,
    CCSX_TOO_LONG              = 6,
    CCSX_LOW_RQ                = 7,  /* predicted accuracy below opts.min_rq                          */
    CCSX_EMPTY_WINDOW          = 8,  /* EMPTY_WINDOW_DURING_POLISHING                                 */
    CCSX_CAPACITY              = 9   /* the polished consensus outgrew its buffer (1.25 x longest subread + 64): reported, never truncated */
};

/* ---- Arrow model parameter blob (interface of docs/faq/chemistry.md:27-56 the "arrow" json files) ----
 * ctx = 4*prev_base + cur_base, bases A,C,G,T = 0..3, obs = base*3 + min(pw,3)-1.
 * transition weights are cubic polynomials in the SNR of the current base's channel:
 *   w = c0 + c1 s + c2 s^2 + c3 s^3, s = clamp(snr[cur], snr_lo, snr_hi), w = max(w, 1e-6)
 *   P(move) = w_move / (1 + w_branch + w_stick + w_del), P(match) = 1 / (1 + ...).            */
typedef struct ccsx_model {
    char  name[32];
    float snr_lo, snr_hi;
    float trans_poly[CCSX_NCTX][3][4];      /* [ctx][branch,stick,deletion][c0..c3]              */
    float em_match [CCSX_NCTX][CCSX_NOBS];  /* P(obs | match, ctx)  (sums to 1 over obs)         */
    float em_branch[CCSX_NCTX][3];          /* P(pw bin | branch, ctx) (base is the cognate)     */
    float em_stick [CCSX_NCTX][3];          /* P(pw bin | stick, ctx)  (base uniform over 3)     */
} ccsx_model;

/* ---- run options (CLI names from SURVEY.md App. C) ---- */
typedef struct ccsx_opts {
    int32_t max_poa_cov;     /* --maxPoaCoverage (docs/changelog.md:114): subreads threaded into the POA */
    int32_t min_passes;      /* --min-passes                                                    */
    int32_t top_passes;      /* at most this many of a ZMW's passes are used, the FIRST ones in batch order (0 or > CCSX_MAX_PASSES = all, up to CCSX_MAX_PASSES).  The reference's
                                --top-passes ("closest to the median length", docs/faq/accuracy-vs-passes.md:49-52) is a selection the
                                caller makes when it builds the batch: the `ccs` driver does (ccs_main.cpp finish_zmw)             */
    int32_t min_length;      /* --min-length                                                    */
    int32_t max_length;      /* --max-length                                                    */
    float   min_rq;          /* --min-rq                                                        */
    int32_t poa_slots;       /* concurrent POA graphs resident on the device (0 = auto)          */
    int32_t hifi_kinetics;   /* --hifi-kinetics (docs/faq/kinetics.md:8-18): per-strand averaged IPD / PW; needs batch.ipd */
    int32_t disable_heuristics; /* --disable-heuristics (docs/faq/low-complexity.md:15): no candidate filter, every position is polished */
    float   min_zscore;      /* a pass is dropped from a window when its z-score (log-likelihood vs the model's expectation for
                                the window template) is below this; 0 = gate off                                   */
    int32_t handles_per_device; /* handles the caller runs on this GPU (0/1 = one): each takes 1/N of the free HBM for its POA scratch */
    int32_t no_fallback_draft;  /* 1: a failed / unmappable first draft is final (default 0: one fallback draft, SPEC "fallback draft") */
    int32_t max_insertion_size; /* SPEC "trim large insertions" (docs/how-does-ccs-work.md:74-78, --max-insertion-size): a subread segment more
                                 * than this many bases longer than its window is cut down to the window's length before polishing;
                                 * 0 = the default 30, < 0 = never trim (such a segment then leaves the window when it exceeds 63 bases) */
    int32_t serial_stages;      /* 1: draft and polish stage of all batches on ONE compute stream (A/B switch; default 0: the draft stage
                                 * of batch k+1 runs on its own stream under the polish stage of batch k, docs/img/ccs-impl.png)            */
    int32_t max_qv;             /* largest per-base QV that is reported: every per-base error probability (and with it rq) is floored at 10^(-max_qv/10).
                                 * 0 = the default 50 (SPEC v7 "honest QVs": nothing measured on synthetic data supports a higher claim);
                                 * 93 = the reference's documented range (docs/faq/qv-binning.md:31 bins [40, 93]); values above 93 mean 93.  A deviation
                                 * from the reference's output that a caller can switch off: INTEGRATION.md "QV policy"                                  */
} ccsx_opts;

/* ---- input batch: SoA + CSR ---- */
typedef struct ccsx_batch {
    int32_t        n_zmw;
    int32_t        n_reads;      /* R = read_off[n_zmw]                                         */
    int64_t        n_bases;      /* base_off[R]                                                 */
    const int32_t *zmw_id;       /* [n_zmw]  hole number (zm tag)                               */
    const float   *snr;          /* [n_zmw][4] A,C,G,T (sn tag)                                 */
    const int32_t *read_off;     /* [n_zmw+1] first read of each ZMW                            */
    const int64_t *base_off;     /* [R+1] first base of each read                               */
    const uint8_t *bases;        /* [n_bases] codes 0..3 = A,C,G,T, native (sequenced) orientation; only the low two bits of
                                    a byte are used, so no input byte can index out of range on the device */
    const uint8_t *pw;           /* [n_bases] pulse width, CodecV1 code as stored in the pw:B,C tag (codes < 64 ARE the
                                    frame count, so the HMM's pulse-width bin min(pw,3) needs no decoding)        */
    const uint8_t *ipd;          /* [n_bases] inter-pulse duration, CodecV1 code (ip:B,C tag).  Unused by the HMM;
                                    may be NULL unless opts.hifi_kinetics is set                                   */
    const uint8_t *flags;        /* [R] bit0: pass is on the reverse strand (cx REVERSE_PASS)
                                        bit1: PARTIAL pass (not flanked by adapters on both sides: the first / last subread of the
                                              polymerase read).  Not used for the draft, not counted in np; aligned to the draft anchored
                                              at one end and used by the polish where it reaches (docs/faq/accuracy-vs-passes.md:26-29:
                                              ec ~ np + 1).  A ZMW's partial passes must FOLLOW its full-length passes.
                                        bit2: (partial only) the adapter is at the pass's END (cx ADAPTER_AFTER only), else at its start */
} ccsx_batch;

/* ---- results: caller-allocated; seq/qual/raw_qv are laid out at seq_off[z] (capacity layout
 *      from ccsx_result_layout), seq_len[z] bases valid ---- */
typedef struct ccsx_results {
    int32_t  n_zmw;
    int64_t  seq_capacity;       /* total elements in seq / qual / raw_qv                       */
    int64_t *seq_off;            /* [n_zmw+1] (filled by ccsx_result_layout)                    */
    int32_t *status;             /* [n_zmw] enum ccsx_status                                    */
    int32_t *seq_len;            /* [n_zmw]                                                     */
    uint8_t *seq;                /* [seq_capacity] codes 0..3                                   */
    uint8_t *qual;               /* [seq_capacity] phred 0..93                                  */
    float   *raw_qv;             /* [seq_capacity] un-rounded QV (may be NULL)                  */
    float   *rq;                 /* [n_zmw] predicted accuracy (rq tag)                         */
    int32_t *np;                 /* [n_zmw] passes used (np tag)                                */
    float   *ec;                 /* [n_zmw] effective coverage (ec tag)                         */
    int32_t *iters;              /* [n_zmw] total polish iterations over all windows            */
    int32_t *n_windows;          /* [n_zmw]                                                     */
    /* HiFi kinetics (opts.hifi_kinetics; any of these may be NULL).  CodecV1 codes, indexed like seq (orientation of
     * SEQ; a BAM writer that stores ri/rp in the reverse strand's own orientation reverses them).  Tags fi fp ri rp
     * fn rn of docs/faq/bam-output.md:13-23; with --by-strand the forward pair is the record's ip / pw.           */
    uint8_t *fi, *fp;            /* [seq_capacity] mean IPD / pulse width of the passes on SEQ's strand         */
    uint8_t *ri, *rp;            /* [seq_capacity] same for the passes on the opposite strand                   */
    int32_t *fn, *rn;            /* [n_zmw] passes used per strand (fn + rn = np); filled with or without kinetics */
} ccsx_results;

/* ---- per-kernel device timings of the last run, HIP events on the handle's stream (ms) ---- */
typedef struct ccsx_timings {
    float setup_ms, draft_ms, align_ms, polish_ms, stitch_ms, total_ms;
    int64_t polish_workgroups;   /* launched polish workgroups that had work                     */
    float queue_ms;              /* between the two stages: end of the draft stage -> start of the polish stage (the polish stream
                                    was still busy with the previous batch); total_ms includes it               */
    float reserved_;
    double start_ms, end_ms;     /* device time of the batch's first / last kernel since the handle's creation: over a run of
                                    tickets (end of the last - start of the first) is the time the kernels alone took     */
} ccsx_timings;

typedef struct ccsx_handle_s *ccsx_handle;

/* library / device */
int         ccsx_abi_version(void);
int         ccsx_spec_version(void);                    /* version of the algorithm specification the kernels implement:
                                                           golden vectors carry it, so SPEC drift fails loudly           */
const char *ccsx_build_flags(void);                     /* "" for the product library.  Anything else names the compile-time switches of an experimental
                                                           build (tuning overrides, timing-only variants that compute wrong results: those also make
                                                           ccsx_spec_version() negative).  tests/test_abi.py requires "" of the library it ships   */
const char *ccsx_runtime_switches(void);                /* the CCSX_* scheduling / debugging overrides found in this process's environment, "NAME=VALUE ..." ("" = none).
                                                           They never change a result, they change timings: a benchmark line states them (bench.py config.runtime_switches)
                                                           and tests/test_abi.py requires "" under a clean environment                                                       */
const char *ccsx_last_error(void);
int         ccsx_device_count(void);

/* model + options */
void        ccsx_model_default(ccsx_model *m);          /* synthetic parameter set "SYN-1"      */
void        ccsx_opts_default(ccsx_opts *o);

/* model parameter files and chemistry lookup (docs/faq/chemistry.md:27-56, docs/changelog.md:66,101).  The json schema is
 * this library's own ("ConsensusModelVersion": "ccsx-1"): the ccsx_model blob plus the (BindingKit, SequencingKit,
 * BasecallerVersion) triples it supports; floats round-trip exactly.                                                      */
int         ccsx_model_from_json(const char *json_text, ccsx_model *m);
int         ccsx_model_load(const char *path, ccsx_model *m);
/* writes at most cap-1 bytes + NUL, returns the bytes needed (call with buf = NULL to size); the triple may be NULL      */
int64_t     ccsx_model_to_json(const ccsx_model *m, const char *binding_kit, const char *sequencing_kit,
                               const char *basecaller_version, char *buf, int64_t cap);
/* $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/ (every .json there; injected models take precedence), then the built-in set.  <0 with
 * "Unsupported chemistries found: (...)" when no model supports the triple (basecaller versions match on major.minor)    */
int         ccsx_model_for_chemistry(const char *binding_kit, const char *sequencing_kit, const char *basecaller_version,
                                     ccsx_model *m);

/* lifecycle: binds to GPU `device_ordinal`, creates a stream, copies the model */
int         ccsx_create(int device_ordinal, const ccsx_model *model, const ccsx_opts *opts, ccsx_handle *out);
int         ccsx_destroy(ccsx_handle h);

/* page-locked host memory for batch arrays (optional: any host memory works; pinned buffers upload by DMA at PCIe
 * rate instead of through the runtime's staging copy).  NULL on failure (ccsx_last_error). */
/* NUMA placement of a device's host threads on a multi-GPU node (docs/faq/parallelize.md:8-29): the device's NUMA node from its PCI address
 * (/sys/bus/pci/devices/<id>/numa_node; -1 = unknown / the platform reports none), and binding the CALLING thread to that node's CPUs (intersected with the
 * CPUs the process may use).  Bind a device's worker / packing threads BEFORE they allocate page-locked staging: first touch then puts the staging on the
 * device's node.  Both return the node, or -1 when nothing was (or could be) done — never an error: placement is an optimisation.  CCSX_NUMA=0 turns binding off. */
int         ccsx_pci_numa_node(const char *pci_bus_id);
int         ccsx_device_numa_node(int device);
int         ccsx_bind_thread_to_node(int node);
int         ccsx_bind_thread_to_device(int device);
void       *ccsx_alloc_pinned(size_t bytes);
void        ccsx_free_pinned(void *p);

/* result sizing: fills res->seq_off[0..n] and returns the total capacity needed (elements) */
int64_t     ccsx_result_layout(const ccsx_batch *b, int64_t *seq_off);

/* fused path, host buffers in / host buffers out (one synchronous call = upload + run + download) */
int         ccsx_consensus_batch(ccsx_handle h, const ccsx_batch *b, ccsx_results *res);

/* asynchronous pipeline (submit / wait tickets).  ccsx_submit enqueues upload -> kernels -> download of one
 * batch on the handle's streams (H2D, draft stage, polish stage, D2H) and returns; up to three batches are in flight, so the
 * copies of batch k+1 / k-1 and the draft stage of batch k+1 run under the polish stage of batch k.  The batch arrays and the result buffers must stay valid (and
 * should be page-locked: ccsx_alloc_pinned) until ccsx_wait returns for that ticket.  Tickets complete in order.      */
typedef int64_t ccsx_ticket;
int         ccsx_submit(ccsx_handle h, const ccsx_batch *b, ccsx_results *res, ccsx_ticket *ticket);
int         ccsx_wait(ccsx_handle h, ccsx_ticket ticket);              /* results of that batch are in `res`   */
int         ccsx_poll(ccsx_handle h, ccsx_ticket ticket);              /* 1 done, 0 not yet, <0 error          */
int         ccsx_ticket_timings(ccsx_handle h, ccsx_ticket ticket, ccsx_timings *t);

/* ---- the two seams of the reference's block diagram (docs/img/ccs-impl.png: a GPU consumer on the DRAFT queue and one on the POLISH queue;
 * docs/faq/revio.md:35-53: Arrow runs a second time, for QVs only, on a sequence that was made elsewhere).  ccsx_consensus_batch / ccsx_submit are the two fused.
 *   ccsx_draft_batch   draft stage only: draft cascade (POA -> fallback POA -> last resort) incl. the alignments that drive it; the drafts, their window
 *                      bounds and per-ZMW statuses go to caller buffers.
 *   ccsx_polish_batch  alignment cascade + windowing + Arrow polish + QVs on CALLER-SUPPLIED drafts (a host that drafts on its CPU pool, e.g. with SPOA).  With
 *                      CCSX_QV_ONLY no mutation is applied: one scoring round, QVs / rq / np / ec for the sequence as given.
 * ccsx_polish_batch(ccsx_draft_batch(b)) returns byte for byte what ccsx_consensus_batch(b) returns (tests/test_gpu_parity.py).  A draft nothing maps to yields
 * a per-ZMW status (TOO_MANY_UNUSABLE, DRAFT_FAILURE for length 0 or a length beyond the slot), never an error of the call.                                   */
/* Lifetime and memory: with the TICKETED forms the arrays of a ccsx_drafts are read (ccsx_submit_polish) / written (ccsx_submit_draft) by
 * asynchronous copies: they must stay valid until ccsx_wait returns for the ticket, and they should be page-locked (ccsx_alloc_pinned) — a copy from / to
 * pageable memory blocks the submitting thread until the stage has finished, so "three in flight" degenerates to one at a time (correct, but serial).
 * As INPUT (ccsx_polish_batch / ccsx_submit_polish) only seq_off, len, seq and backbone are read; status, n_windows, win_bounds and win_off may be NULL.   */
typedef struct ccsx_drafts {
    int32_t  n_zmw;
    int64_t  seq_capacity;       /* elements in seq, from ccsx_draft_layout (= the capacity layout of ccsx_result_layout)               */
    int64_t  win_capacity;       /* elements in win_bounds, from ccsx_draft_layout                                                      */
    int64_t *seq_off;            /* [n_zmw+1] slot of every ZMW's draft in seq (capacity layout; ccsx_draft_layout fills it)            */
    int64_t *win_off;            /* [n_zmw+1] slot of every ZMW's window bounds in win_bounds (ccsx_draft_layout fills it)               */
    int32_t *status;             /* [n_zmw] out of ccsx_draft_batch: enum ccsx_status so far (SUCCESS = a usable draft); ignored as input */
    int32_t *len;                /* [n_zmw] draft length; 0 = none                                                                       */
    uint8_t *seq;                /* [seq_capacity] codes 0..3, len[z] bases at seq_off[z]                                                */
    int32_t *backbone;           /* [n_zmw] index (within the ZMW) of a pass that has the draft's orientation: the strand reference of the
                                    consensus, of fn / rn and of the kinetics planes (ccsx_draft_batch: the backbone pass of the generator
                                    that succeeded)                                                                                      */
    int32_t *n_windows;          /* [n_zmw] out of ccsx_draft_batch (may be NULL)                                                        */
    int32_t *win_bounds;         /* [win_capacity] out of ccsx_draft_batch (may be NULL): n_windows[z] + 1 core bounds at win_off[z]       */
} ccsx_drafts;
#define CCSX_QV_ONLY 1u          /* ccsx_polish_batch flag */
void        ccsx_draft_layout(const ccsx_batch *b, int64_t *seq_off, int64_t *win_off, int64_t *seq_capacity, int64_t *win_capacity);
int         ccsx_draft_batch(ccsx_handle h, const ccsx_batch *b, ccsx_drafts *drafts);
int         ccsx_polish_batch(ccsx_handle h, const ccsx_batch *b, const ccsx_drafts *drafts, ccsx_results *res, uint32_t flags);
/* the same with tickets (ccsx_wait / ccsx_poll / ccsx_ticket_timings as for ccsx_submit; the two kinds share the handle's three slots) */
int         ccsx_submit_draft(ccsx_handle h, const ccsx_batch *b, ccsx_drafts *drafts, ccsx_ticket *ticket);
int         ccsx_submit_polish(ccsx_handle h, const ccsx_batch *b, const ccsx_drafts *drafts, ccsx_results *res, uint32_t flags, ccsx_ticket *ticket);

/* split form used by the benchmark (inputs resident in HBM when the timed region starts)        */
int         ccsx_upload(ccsx_handle h, const ccsx_batch *b);           /* H2D + workspace sizing */
int         ccsx_run(ccsx_handle h);                                   /* all kernels, async     */
int         ccsx_sync(ccsx_handle h);                                  /* wait for the stream    */
int         ccsx_download(ccsx_handle h, ccsx_results *res);           /* D2H                    */
int         ccsx_get_timings(ccsx_handle h, ccsx_timings *t);

/* stage access for parity tests (valid after ccsx_run + ccsx_sync on the uploaded batch)        */
int         ccsx_stage_draft(ccsx_handle h, int32_t zmw_index, uint8_t *draft, int32_t cap, int32_t *len);
int         ccsx_stage_align(ccsx_handle h, int32_t read_index, int32_t *rstart, int32_t cap,
                             int32_t *valid, int32_t *score);
int         ccsx_stage_windows(ccsx_handle h, int32_t zmw_index, int32_t *bounds, int32_t cap, int32_t *n_windows);

/* deterministic synthetic subread generator (SURVEY.md §8d / BASELINE.md §3).  Caller frees with ccsx_synth_free */
typedef struct ccsx_synth {
    ccsx_batch batch;            /* arrays are owned by this object                             */
    int64_t   *tpl_off;          /* [n_zmw+1] */
    uint8_t   *tpl;              /* true templates, codes 0..3, orientation of read 0           */
} ccsx_synth;
int         ccsx_synth_generate(int32_t n_zmw, int32_t first_zmw_id, int32_t passes_lo, int32_t passes_hi,
                                int32_t len_lo, int32_t len_hi, uint64_t seed, ccsx_synth **out);
void        ccsx_synth_free(ccsx_synth *s);

#ifdef __cplusplus
}
#endif
#endif /* CCSX_H */
