/*
 * ccs_oracle.c — CPU restatement of the CCS per-ZMW consensus hot path.  TEST INFRASTRUCTURE ONLY.
 *
 *   *** PARITY UNPINNED ***
 *   /root/reference (PacificBiosciences/ccs) is a documentation-only mount: it holds no source, no
 *   binary, no tests and no golden vectors (docs/faq/source-code.md:11-15).  The
 *   arithmetic of this path lives in PacBio's closed `unanimity` library (last public snapshot
 *   PacificBiosciences/unanimity@6f11a13e, un-vendored, unreachable offline).  This file therefore
 *   restates the algorithm the reference DOCUMENTS (docs/how-does-ccs-work.md:34-112) under the exact
 *   specification written in DESIGN.md §SPEC, and is validated by first-principles tests
 *   (tests/test_oracle_*.py: brute-force HMM, alpha/beta agreement, mutation equivalence,
 *   recover-the-truth), not by reference outputs.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *   The product (ccs_amd/, include/ccsx.h) never links, imports or calls it.
 *
 * Steps restated (docs/how-does-ccs-work.md):
 *   :34-51  step 2  draft      -> orc_poa_*      sparse partial-order alignment, adaptive 32-row band (POA_BAND)
 *   :53-55  step 3  alignment  -> orc_align      subread -> draft banded global alignment, rstart[]
 *   :57-61  step 4  windowing  -> orc_windows    22 bp cores, +-2 bp overhang, homopolymer-safe breaks
 *   :87-101 step 8  polishing  -> orc_polish_window   Arrow pair-HMM (match/branch/stick/deletion,
 *                                 dinucleotide context, PW + SNR dependent), 3 sub + 4 ins + 1 del per position
 *   :103-106 step 9 QV         -> inside orc_polish_window (LL ratios), rq = 1 - mean(p_err)
 *   :108-112 step 10 final     -> orc_consensus_zmw  (concatenate cores, trim overhangs)
 *
 * Build:  make -C oracle   (gcc -O2 -ffp-contract=off; no fast-math: bit-reproducibility matters)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------- specification constants (same values as include/ccsx.h) -------------- */
#define ORC_SPEC_VERSION 8  /* = CCSX_SPEC_VERSION of include/ccsx.h (tests/test_abi.py); the golden vectors carry it */
int orc_spec_version(void) { return ORC_SPEC_VERSION; }
#define BAND      64
#define ALIGN_BAND1 16      /* rows of the FIRST attempt of the subread -> draft alignment (step 3); BAND rows on failure */
#define ALIGN_OFF1  6       /* rows of the narrow band ABOVE the best row: 6 above, 9 below — insertion bursts push the path DOWN, a deleted
                             * stretch leaves it where it is (2.2 % of the 10 kb passes lose a symmetric 16-row band, 0.2 % this one)     */
#define POA_BAND  32        /* rows of the POA's band (step 2): on the device four graphs share a wave, each in a 16-lane DPP row with two rows
                             * per lane.  Drafts equal those of a 64-row band on every test set (tools/acc_eval.py, profiles/r03_spec_studies.txt) */
static __thread int g_bw = BAND;   /* rows of the band in use (alignment retry / split alignment: BAND) */
/* test hooks: the build-defined approximations of the SPEC as variables, so that tools/acc_eval.py can measure what each of them costs on
 * off-model and low-complexity data (profiles/r03_spec_studies.txt).  The product implements the SPEC values only. */
static int g_poa_band = POA_BAND, g_align_band1 = 16, g_score_band = 5, g_skip_margin = 6;
static int g_sat_rows = 1, g_sat_gain = 1;                    /* SPEC v5 "band saturation" (see orc_align_ev_w); 0 / very negative = off */
void orc_set_sat_rows(int n) { g_sat_rows = n; }
void orc_set_sat_gain(int g) { g_sat_gain = g; }
void orc_set_poa_band(int bw) { g_poa_band = bw; }
void orc_set_align_band1(int bw) { g_align_band1 = bw; }      /* 64: no narrow first attempt */
void orc_set_score_band(int w) { g_score_band = w; }          /* >= 64: the full sum over alignments */
void orc_set_skip_margin(int m) { g_skip_margin = m; }
/* SPEC v6 "banded fill": alpha and beta of a (read, window) pair exist on the diagonals d = j - i in
 *   [max(-I, min(0, J - I) - (Wr + FILL_MARGIN_LO)),  min(J, max(0, J - I) + (Wr + FILL_MARGIN_HI))],   Wr = SCORE_BAND + max(0, |I - J| - 2)
 * only; every cell outside is an exact zero for BOTH matrices (the same set of paths: alpha(I,J) and beta(0,0) still agree), and the mutation scoring reads
 * zeros there.  The band holds every cell an unclamped scoring row reads: gamma(i, c) on c - i in [min(0,J-I) - Wr, max(0,J-I) + Wr], beta(i', q) with q <= c + 2 two
 * diagonals higher — hence the asymmetric margins.  On the device gamma / beta are STORED by diagonal (15 instead of 28 floats per row: twice the reads per LDS chunk)
 * and filled on band diagonals.  Measured against the full matrices on six data sets: bit-identical sequences, QVs, rq and round counts
 * (profiles/r05_band_study.txt; tools/band_study.py).  Knobs for that study: orc_set_fill_band(w) = symmetric margins w, < 0 = the full matrices (SPEC v5);
 * orc_set_fill_margins(lo, hi). */
#define FILL_MARGIN_LO 2
#define FILL_MARGIN_HI 2
static int g_fill_lo = FILL_MARGIN_LO, g_fill_hi = FILL_MARGIN_HI;
void orc_set_fill_band(int w) { g_fill_lo = g_fill_hi = w; }
void orc_set_fill_margins(int lo, int hi) { g_fill_lo = lo; g_fill_hi = hi; }
/* SPEC v8 "fused recurrences": every multiply-add of the alpha / beta fill (A1/A2), of the mutation extension (A3) and of the link (A4) is ONE fused
 * multiply-add (IEEE fma: a single rounding), nested in the order written in DESIGN.md §2 — 3 instead of 5 floating-point operations per fill cell and per
 * scoring-row half, and a dependent chain of fma-fma instead of mul-add-add on the device.  SURVEY.md §7.3 H1(a) names "explicit fmaf everywhere on both" as
 * the sanctioned alternative to -ffp-contract=off.  orc_set_fma(0) restores the separate roundings of SPEC v7 (the study of profiles/r06_spec_v8_study.txt). */
#define FMA_DEFAULT 1
static int g_fma = FMA_DEFAULT;
void orc_set_fma(int on) { g_fma = on; }
/* SPEC v8 "joint band test": row i of a mutation's scoring band reads gamma(i, c) and beta(i+1, q), which sit q - c - 1 (0 or 1) diagonals apart; BOTH are taken as
 * zeros unless both lie on the fill band of SPEC v6: dlo <= c - i <= dhi - (q - c - 1) (v7 tested each cell's own diagonal).  Only rows of a scoring band that is
 * clamped into a corner of the window are affected, at the band's outermost diagonal; the device needs one compare per row in place of two. */
#define CLIP_DEFAULT 1
static int g_clip = CLIP_DEFAULT;
void orc_set_score_clip(int on) { g_clip = on; }
static inline float mad(float a, float b, float c) { return g_fma ? __builtin_fmaf(a, b, c) : (a * b) + c; }   /* a*b + c */
/* the band of one (read, window) pair */
static inline void fill_band_of(int I, int J, int score_band, int *dlo, int *dhi)
{
    if (g_fill_lo < 0 || g_fill_hi < 0) { *dlo = -(1 << 20); *dhi = 1 << 20; return; }
    int dIJ = I > J ? I - J : J - I;
    int Wr = score_band + (dIJ > 2 ? dIJ - 2 : 0);
    int lo = (J - I < 0 ? J - I : 0) - (Wr + g_fill_lo), hi = (J - I > 0 ? J - I : 0) + (Wr + g_fill_hi);
    *dlo = lo < -I ? -I : lo; *dhi = hi > J ? J : hi;
}
/* ---- path / work counters of the tests and of bench.py's counted-work figure (SURVEY.md §8d "algorithmic work per ZMW: counted on an
 * instrumented CPU path, not estimated"); per-thread tallies are flushed into the global sums once per ZMW ---- */
enum { CNT_TRIM, CNT_SPLIT, CNT_SPLIT_S0, CNT_SPLIT_SLD, CNT_FALLBACK, CNT_RETRY64, CNT_ZDROP, CNT_NONCONV_WIN, CNT_POA_WIDE, CNT_THIRD_DRAFT,
       CNT_PARTIAL_USED, CNT_CELLS_POA, CNT_CELLS_ALIGN, CNT_CELLS_FILL, CNT_CELLS_SCORE, CNT_ZMWS, CNT_SPLIT2, CNT_SATURATED, CNT_CLOSED_TRACT, CNT_N };
static int64_t orc_cnt_global[CNT_N];
static __thread int64_t orc_cnt[CNT_N];
static __thread int g_cells_kind = CNT_CELLS_ALIGN;         /* which tally dp_column feeds */
void orc_counts_reset(void) { memset(orc_cnt_global, 0, sizeof(orc_cnt_global)); }
void orc_counts_get(int64_t *out) { memcpy(out, orc_cnt_global, sizeof(orc_cnt_global)); }
int orc_counts_n(void) { return CNT_N; }
static void orc_counts_flush(void)
{
    for (int k = 0; k < CNT_N; ++k) if (orc_cnt[k]) {
#pragma omp atomic
        orc_cnt_global[k] += orc_cnt[k];
        orc_cnt[k] = 0;
    }
}
static __thread int g_poa_scores[64], g_poa_nscores = 0;   /* test hook: end scores of the passes threaded into the last POA */
int orc_poa_last_scores(int *out) { for (int i = 0; i < g_poa_nscores; ++i) out[i] = g_poa_scores[i]; return g_poa_nscores; }
#define MAX_PASSES 255      /* passes of a ZMW that are used (include/ccsx.h CCSX_MAX_PASSES) */
#define MAXPRED   7         /* in-edge cap of a POA vertex (SPEC v3; the device stores a move in a nibble) */
#define WIN_CORE  22
#define WIN_OVH   2
#define JMAX      31
#define IMAX      63
#define MAX_ITER  8
#define NCTX      16
#define NOBS      12
#define SC_MATCH    3
#define SC_MISMATCH (-5)
#define SC_INS      (-4)
#define SC_DEL      (-4)
#define NEG       (-(1 << 28))
#define MUT_EPS   0.01f     /* favourable iff summed log2-likelihood gain > MUT_EPS                 */
#define RESCUE_MIN_EXCESS 24 /* a failed pass is tried as prefix + insertion + suffix iff it is this much longer than the draft */
#define SCORE_BAND 5        /* half width (read rows) of the mutation scoring band around the window diagonal */
#define MUT_SEP   5         /* accepted mutations of one round are >= MUT_SEP columns apart          */
#define MULTI_ROUNDS 2      /* rounds >= MULTI_ROUNDS apply only the single best mutation (cycle guard) */
#define JMIN_DEL  4         /* deletions are not applied when the window would shrink to <= JMIN_DEL */
#define AB_TOL    0.01f     /* |log2 alpha(I,J) - log2 beta(0,0)| tolerance (alpha/beta agreement)   */
#define TINY_P    1e-30f    /* a read whose scaled likelihood falls below this is unusable in the window */
#define SKIP_MARGIN 6       /* candidate filter: a non-homopolymer position is skipped iff clean - dirty >= SKIP_MARGIN   */
#define SKIP_SPREAD 3       /* ... and no position within SKIP_SPREAD of it has a dirty majority (clean - dirty < 0); the
                               same neighbourhood of every applied mutation is polished in the following rounds           */
#define DQ_SCALE  65536.0f  /* per-read log2-likelihood gains are summed as fixed point (2^-16): order independent      */
#define DQ_CLAMP  100.0f
#define PERR_FLOOR g_perr_floor      /* SPEC v7: smallest per-base error probability that is reported (Q50 by default; opts.max_qv; v6: 1e-10) */
#define SKIP_PERR_FLOOR g_perr_floor /* SPEC v7: the same floor for a position the candidate filter skips */

typedef struct orc_model {
    char  name[32];
    float snr_lo, snr_hi;
    float trans_poly[NCTX][3][4];
    float em_match[NCTX][NOBS];
    float em_branch[NCTX][3];
    float em_stick[NCTX][3];
} orc_model;

typedef struct orc_opts {
    int32_t max_poa_cov, min_passes, top_passes, min_length, max_length;
    float   min_rq;
    int32_t poa_slots;
    int32_t hifi_kinetics;
    int32_t disable_heuristics;     /* --disable-heuristics: no candidate filter (every position is polished) */
    float   min_zscore;             /* a pass is dropped from a window when its z-score is below this (0 = gate off) */
    int32_t handles_per_device;     /* (engine only) */
    int32_t no_fallback_draft;      /* 1: a failed / unmappable first draft is final */
    int32_t max_insertion_size;     /* trim segments longer than window + this (0 = 30, < 0 = off) */
    int32_t serial_stages;          /* (engine only) */
    int32_t max_qv;                 /* largest per-base QV reported: p_err >= 10^(-max_qv/10); <= 0 = 50 (SPEC v7), > 93 = 93 (include/ccsx.h ccsx_opts.max_qv) */
} orc_opts;
/* the floor of every reported per-base error probability (the same expression as ccsx_kernels.h ccsx_perr_floor) */
static float perr_floor_of(int max_qv) { if (max_qv <= 0) max_qv = 50; if (max_qv > 93) max_qv = 93; return max_qv == 50 ? 1e-5f : (float)pow(10.0, -(double)max_qv / 10.0); }
static __thread float g_perr_floor = 1e-5f;   /* set per ZMW from its options */

/* ---------------- deterministic log2 / exp2 (DESIGN.md §SPEC "det math") -------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_log2f(float x)
{
    uint32_t u = f2u(x);
    if ((int32_t)u < 0x00800000) return -127.0f;         /* zero, denormal, negative */
    int e = (int)(u >> 23) - 127;
    float f = u2f((u & 0x007fffffu) | 0x3f800000u);       /* [1,2) */
    if (f > 1.41421356f) { f = f * 0.5f; e = e + 1; }
    float t = f - 1.0f;
    float s = t / (2.0f + t);
    float z = s * s;
    float p = z * 0.111111111f;
    p = p + 0.142857143f;
    p = p * z;
    p = p + 0.2f;
    p = p * z;
    p = p + 0.333333333f;
    p = p * z;
    p = p + 1.0f;
    float ln = (2.0f * s) * p;
    return (float)e + ln * 1.44269504f;
}

float orc_exp2f(float x)
{
    if (x < -125.0f) x = -125.0f;
    if (x > 60.0f) x = 60.0f;
    float n = floorf(x + 0.5f);
    float f = (x - n) * 0.693147181f;                     /* [-0.3466, 0.3466] */
    float p = f * 1.98412698e-4f;                         /* 1/5040 */
    p = p + 1.38888889e-3f;                               /* 1/720 */
    p = p * f;
    p = p + 8.33333333e-3f;
    p = p * f;
    p = p + 4.16666667e-2f;
    p = p * f;
    p = p + 0.166666667f;
    p = p * f;
    p = p + 0.5f;
    p = p * f;
    p = p + 1.0f;
    p = p * f;
    p = p + 1.0f;
    int ni = (int)n;
    return p * u2f((uint32_t)(ni + 127) << 23);
}

/* ---------------- A0: per-ZMW parameter tables (docs/how-does-ccs-work.md:90-94) ------------------------ */
/* ME[k][o] = 4*P(match|k)*P(o|match,k); INS[k][o] = 4*P(branch or stick emitting o | k); DL[k] = P(deletion|k).
 * The factor 4 per emitted base is an exact power-of-two range shift (DESIGN.md §SPEC "scaling").            */
void orc_tables(const orc_model *m, const float *snr, float *ME, float *INS, float *DL)
{
    for (int k = 0; k < NCTX; ++k) {
        int cur = k & 3;
        float s = snr[cur];
        if (s < m->snr_lo) s = m->snr_lo;
        if (s > m->snr_hi) s = m->snr_hi;
        float w[3];
        for (int mv = 0; mv < 3; ++mv) {
            const float *c = m->trans_poly[k][mv];
            float t = c[3] * s;
            t = t + c[2];
            t = t * s;
            t = t + c[1];
            t = t * s;
            t = t + c[0];
            if (t < 1e-6f) t = 1e-6f;
            w[mv] = t;
        }
        float den = 1.0f + w[0];
        den = den + w[1];
        den = den + w[2];
        float pM = 1.0f / den, pB = w[0] / den, pS = w[1] / den, pD = w[2] / den;
        for (int o = 0; o < NOBS; ++o) {
            int b = o / 3, pwb = o % 3;
            ME[k * NOBS + o] = (pM * m->em_match[k][o]) * 4.0f;
            if (b == cur) INS[k * NOBS + o] = (pB * m->em_branch[k][pwb]) * 4.0f;
            else          INS[k * NOBS + o] = ((pS * m->em_stick[k][pwb]) * 0.333333333f) * 4.0f;
        }
        DL[k] = pD;
    }
}

/* A7, z-score gate ([RECALL] unanimity AddRead: reads whose likelihood is improbably low under the model are not
 * used; docs name the effect only: docs/faq/accuracy-vs-passes.md:22-24).  Per context k the mean MU[k] and variance
 * VAR[k] of the log2-likelihood one template position contributes when a read is GENERATED by the model: a geometric
 * number of stay events (branch / stick, each with its emission) followed by one advance (match with its emission, or
 * deletion).  A read's z-score in a window is (log2 P(read | template) - sum_j MU[k_j]) / sqrt(sum_j VAR[k_j]).      */
void orc_zparams(const orc_model *m, const float *snr, float *MU, float *VAR)
{
    for (int k = 0; k < NCTX; ++k) {
        int cur = k & 3;
        float s = snr[cur];
        if (s < m->snr_lo) s = m->snr_lo;
        if (s > m->snr_hi) s = m->snr_hi;
        float w[3];
        for (int mv = 0; mv < 3; ++mv) {
            const float *c = m->trans_poly[k][mv];
            float t = c[3] * s;
            t = t + c[2];
            t = t * s;
            t = t + c[1];
            t = t * s;
            t = t + c[0];
            if (t < 1e-6f) t = 1e-6f;
            w[mv] = t;
        }
        float den = 1.0f + w[0];
        den = den + w[1];
        den = den + w[2];
        float pM = 1.0f / den, pB = w[0] / den, pS = w[1] / den, pD = w[2] / den;
        float pA = pM + pD, pI = pB + pS;
        float lM = orc_log2f(pM), lD = orc_log2f(pD), lB = orc_log2f(pB), lS = orc_log2f(pS * 0.333333333f);
        float e1m = 0.0f, e2m = 0.0f, e1b = 0.0f, e2b = 0.0f, e1s = 0.0f, e2s = 0.0f;
        for (int o = 0; o < NOBS; ++o) { float p = m->em_match[k][o], l = orc_log2f(p), t = p * l; e1m = e1m + t; e2m = e2m + t * l; }
        for (int b = 0; b < 3; ++b) { float p = m->em_branch[k][b], l = orc_log2f(p), t = p * l; e1b = e1b + t; e2b = e2b + t * l; }
        for (int b = 0; b < 3; ++b) { float p = m->em_stick[k][b], l = orc_log2f(p), t = p * l; e1s = e1s + t; e2s = e2s + t * l; }
        float a1 = (pM * (lM + e1m) + pD * lD) / pA;
        float a2 = (pM * ((lM * lM + (2.0f * lM) * e1m) + e2m) + pD * (lD * lD)) / pA;
        float s1 = (pB * (lB + e1b) + pS * (lS + e1s)) / pI;
        float s2 = (pB * ((lB * lB + (2.0f * lB) * e1b) + e2b) + pS * ((lS * lS + (2.0f * lS) * e1s) + e2s)) / pI;
        float vA = a2 - a1 * a1, vS = s2 - s1 * s1;
        float EN = pI / pA, VN = pI / (pA * pA);
        MU[k] = EN * s1 + a1;
        VAR[k] = (EN * vS + VN * (s1 * s1)) + vA;
    }
}

static inline int ctx_of(int prev, int cur) { if (prev > 3) prev = (cur + 2) & 3; return prev * 4 + cur; }
static inline int obs_of(int base, int pw) { int b = pw; if (b < 1) b = 1; if (b > 3) b = 3; return (base & 3) * 3 + (b - 1); }   /* only the low two bits of a base code count */

/* ---------------- steps 2+3: banded DP column shared by POA and alignment ------------------------------- */
typedef struct {
    int       cap, n;                 /* vertex capacity / count                                   */
    uint8_t  *base;
    int32_t  *nreads;
    uint8_t  *npred;
    int32_t  *pred;                   /* [cap][MAXPRED]                                            */
    int32_t  *next, *prev;            /* topological linked list; head = first, -1 terminated      */
    int32_t   head, tail;
    int32_t  *order;                  /* [n] topological order                                     */
    int32_t  *lo, *colmax, *bestrow;  /* per vertex band start, column max, row of max             */
    int32_t  *M;                      /* [cap][BAND] scores                                        */
    uint8_t  *mv;                     /* [cap][BAND] move | pred slot << 2                         */
    int32_t   nadded;
} poa_t;

enum { MV_DIAG = 0, MV_DEL = 1, MV_INS = 2 };

static poa_t *poa_new(int cap)
{
    poa_t *g = (poa_t *)calloc(1, sizeof(poa_t));
    g->cap = cap;
    g->base = (uint8_t *)malloc(cap);
    g->nreads = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->npred = (uint8_t *)malloc(cap);
    g->pred = (int32_t *)malloc(sizeof(int32_t) * cap * MAXPRED);
    g->next = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->prev = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->order = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->lo = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->colmax = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->bestrow = (int32_t *)malloc(sizeof(int32_t) * cap);
    g->M = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap * BAND);
    g->mv = (uint8_t *)malloc((size_t)cap * BAND);
    g->head = g->tail = -1;
    return g;
}
static void poa_free(poa_t *g)
{
    free(g->base); free(g->nreads); free(g->npred); free(g->pred); free(g->next); free(g->prev);
    free(g->order); free(g->lo); free(g->colmax); free(g->bestrow); free(g->M); free(g->mv); free(g);
}
static int poa_new_vertex(poa_t *g, int base, int after /* -1 = list head */)
{
    if (g->n >= g->cap) return -1;
    int v = g->n++;
    g->base[v] = (uint8_t)base; g->nreads[v] = 1; g->npred[v] = 0;
    if (after < 0) { g->next[v] = g->head; g->prev[v] = -1; if (g->head >= 0) g->prev[g->head] = v; g->head = v; if (g->tail < 0) g->tail = v; }
    else { int nx = g->next[after]; g->next[v] = nx; g->prev[v] = after; g->next[after] = v; if (nx >= 0) g->prev[nx] = v; else g->tail = v; }
    return v;
}
static void poa_add_edge(poa_t *g, int from, int to)
{
    int np = g->npred[to];
    for (int k = 0; k < np; ++k) if (g->pred[to * MAXPRED + k] == from) return;
    if (np >= MAXPRED) return;                         /* SPEC: in-edge cap, extra edges are dropped */
    g->pred[to * MAXPRED + np] = from; g->npred[to] = (uint8_t)(np + 1);
}
static void poa_renumber(poa_t *g) { int k = 0; for (int v = g->head; v >= 0; v = g->next[v]) g->order[k++] = v; }

/* band start of a column whose best predecessor column has (lo_u, bestrow_u); I = read length */
static inline int band_lo(int lo_u, int bestrow_u, int I)
{
    int lo = bestrow_u + 1 - (g_bw == ALIGN_BAND1 ? ALIGN_OFF1 : g_bw / 2);   /* (16 rows: 6 above the best row, 9 below) */
    if (lo < lo_u) lo = lo_u;
    if (lo > lo_u + 2) lo = lo_u + 2;
    int hi = I - (g_bw - 1); if (hi < 0) hi = 0;
    if (lo > hi) lo = hi;
    if (lo < 0) lo = 0;
    return lo;
}

/* One DP column.  preds: npred columns given by (plo[k], pM[k]) ; START column is lo=0, M[l] = l*INS (l<=I).
 * Writes M[BAND], mv[BAND], returns colmax/bestrow through pointers.                                        */
static void dp_column(int vbase, const uint8_t *r, int I, int lo, int npred, const int32_t *plo, const int32_t *const *pM,
                      int32_t *M, uint8_t *mv, int32_t *colmax, int32_t *bestrow)
{
    orc_cnt[g_cells_kind] += (int64_t)g_bw * (npred > 0 ? npred : 1);
    for (int l = 0; l < g_bw; ++l) {
        int i = lo + l;
        int32_t best = NEG; uint8_t bm = 0;
        if (i <= I) {
            for (int k = 0; k < npred; ++k) {
                int o1 = i - 1 - plo[k], o0 = i - plo[k];
                if (i >= 1 && o1 >= 0 && o1 < g_bw) {
                    int32_t x = pM[k][o1];
                    if (x > NEG / 2) { int32_t c = x + (vbase == r[i - 1] ? SC_MATCH : SC_MISMATCH); if (c > best) { best = c; bm = (uint8_t)(MV_DIAG | (k << 2)); } }
                }
                if (o0 >= 0 && o0 < g_bw) {
                    int32_t y = pM[k][o0];
                    if (y > NEG / 2) { int32_t c = y + SC_DEL; if (c > best) { best = c; bm = (uint8_t)(MV_DEL | (k << 2)); } }
                }
            }
        }
        M[l] = best; mv[l] = bm;
    }
    for (int l = 1; l < g_bw; ++l) {
        if (lo + l > I) break;
        int32_t c = M[l - 1] + SC_INS;
        if (c > M[l]) { M[l] = c; mv[l] = MV_INS; }
    }
    int32_t cm = NEG, br = lo;
    for (int l = 0; l < g_bw; ++l) {
        if (M[l] < NEG / 2) M[l] = NEG;
        if (M[l] > cm) { cm = M[l]; br = lo + l; }
    }
    *colmax = cm; *bestrow = br;
}

static void start_column(int I, int32_t *M)
{
    for (int l = 0; l < BAND; ++l) M[l] = (l <= I && l < g_bw) ? l * SC_INS : NEG;
}

/* STUDY HOOK (never the product's path): take the step-3 alignment of the passes that were threaded into the POA from their PATH through the
 * graph (restricted to the consensus vertices) instead of a second DP against the draft.  orc_set_path_align(1) records every threaded pass's vertex per base and
 * the consensus position of every vertex; the whole-ZMW driver then derives entry rows and dirty bits from them (path_align below).  Default 0 = SPEC. */
static int g_path_align = 0;
void orc_set_path_align(int on) { g_path_align = on; }
#define PA_MAXREADS 64
static __thread int32_t *pa_path[PA_MAXREADS]; static __thread int pa_len[PA_MAXREADS], pa_read[PA_MAXREADS], pa_n = 0;
static __thread int32_t *pa_posv = NULL; static __thread int pa_nv = 0;
static void pa_reset(void) { for (int k = 0; k < pa_n; ++k) free(pa_path[k]); pa_n = 0; free(pa_posv); pa_posv = NULL; pa_nv = 0; }
static __thread int pa_cur_read = -1;

/* Thread one read (draft orientation) into the graph.  Returns 1 if added, 0 if skipped, -1 on capacity overflow. */
static int poa_add_read(poa_t *g, const uint8_t *r, int I, int32_t *pathv /* scratch [I] */, int first)
{
    /* SPEC: the FIRST read (the backbone pass of the draft generator) becomes the chain, whatever its length.  An empty backbone leaves an empty
     * graph to which nothing can be threaded: the generator ends in DRAFT_FAILURE and the cascade moves on (round 4: the restatement used to take
     * the first NON-empty read as backbone, the kernels never did — found by tools/corruption_fuzz.py with a zero-length pass 0) */
    if (!first && g->n == 0) return 0;
    if (first) {
        int prev = -1;
        for (int i = 0; i < I; ++i) {
            int v = poa_new_vertex(g, r[i], prev);
            if (v < 0) return -1;
            if (prev >= 0) poa_add_edge(g, prev, v);
            prev = v;
            pathv[i] = v;
        }
        g->nadded = 1; poa_renumber(g);
        if (g_path_align && pa_n < PA_MAXREADS) { pa_path[pa_n] = (int32_t *)malloc(sizeof(int32_t) * (I + 1)); memcpy(pa_path[pa_n], pathv, sizeof(int32_t) * I); pa_len[pa_n] = I; pa_read[pa_n] = pa_cur_read; ++pa_n; }
        return 1;
    }
    int32_t S[BAND]; start_column(I, S);
    const int n0 = g->n;
    for (int k = 0; k < n0; ++k) {
        int v = g->order[k];
        int np = g->npred[v];
        int32_t plo[MAXPRED]; const int32_t *pM[MAXPRED];
        int ulo, ubr;
        if (np == 0) { np = 1; plo[0] = 0; pM[0] = S; ulo = 0; ubr = 0; }
        else {
            int32_t bestcm = NEG - 1; ulo = 0; ubr = 0;
            for (int q = 0; q < np; ++q) {
                int u = g->pred[v * MAXPRED + q];
                plo[q] = g->lo[u]; pM[q] = g->M + (size_t)u * BAND;
                if (g->colmax[u] > bestcm) { bestcm = g->colmax[u]; ulo = g->lo[u]; ubr = g->bestrow[u]; }
            }
        }
        int lo = band_lo(ulo, ubr, I);
        g->lo[v] = lo;
        dp_column(g->base[v], r, I, lo, np, plo, pM, g->M + (size_t)v * BAND, g->mv + (size_t)v * BAND, &g->colmax[v], &g->bestrow[v]);
    }
    /* end vertex: best M[v][I], first in topological order on ties */
    int vend = -1; int32_t bs = NEG;
    for (int k = 0; k < n0; ++k) {
        int v = g->order[k]; int o = I - g->lo[v];
        if (o >= 0 && o < g_bw) { int32_t x = g->M[(size_t)v * BAND + o]; if (x > NEG / 2 && x > bs) { bs = x; vend = v; } }
    }
    if (g_poa_nscores < 64) g_poa_scores[g_poa_nscores++] = bs;
    /* SPEC "POA gate": a pass is threaded only if its alignment reaches the read's last row with a score of at least 1.0 per base
     * (the gate of step 3): a pass the band has lost, or junk, adds nothing to the graph                                        */
    if (vend < 0 || bs < I) return 0;   /* EXPERIMENT: lost band -> retry wide */
    /* traceback: pathv[i] = matched vertex of read base i, or -1 (new vertex) */
    int v = vend, i = I;
    while (v >= 0) {
        uint8_t m = g->mv[(size_t)v * BAND + (i - g->lo[v])];
        int t = m & 3, slot = m >> 2;
        if (t == MV_INS) { pathv[i - 1] = -1; --i; continue; }
        int u = (g->npred[v] == 0) ? -1 : g->pred[v * MAXPRED + slot];
        if (t == MV_DIAG) { pathv[i - 1] = (g->base[v] == r[i - 1]) ? v : -1; --i; }
        v = u;
    }
    while (i > 0) { pathv[i - 1] = -1; --i; }               /* leading insertions at START */
    /* forward replay: thread the read */
    int prevp = -1;
    for (i = 0; i < I; ++i) {
        int w = pathv[i];
        if (w >= 0) g->nreads[w] += 1;
        else { w = poa_new_vertex(g, r[i], prevp); if (w < 0) return -1; }
        if (prevp >= 0) poa_add_edge(g, prevp, w);
        prevp = w;
        pathv[i] = w;
    }
    g->nadded += 1; poa_renumber(g);
    if (g_path_align && pa_n < PA_MAXREADS) { pa_path[pa_n] = (int32_t *)malloc(sizeof(int32_t) * (I + 1)); memcpy(pa_path[pa_n], pathv, sizeof(int32_t) * I); pa_len[pa_n] = I; pa_read[pa_n] = pa_cur_read; ++pa_n; }
    return 1;
}

/* heaviest path: score(v) = 2*nreads(v) - nadded ; best(v) = score(v) + max(0, max_pred best) */
static int poa_consensus(poa_t *g, uint8_t *draft, int cap)
{
    int n = g->n; if (n == 0) return 0;
    int32_t *best = (int32_t *)malloc(sizeof(int32_t) * n), *bp = (int32_t *)malloc(sizeof(int32_t) * n);
    int vbest = -1; int32_t sb = NEG;
    for (int k = 0; k < n; ++k) {
        int v = g->order[k];
        int32_t b = 0, p = -1;
        for (int q = 0; q < g->npred[v]; ++q) { int u = g->pred[v * MAXPRED + q]; if (best[u] > b) { b = best[u]; p = u; } }
        best[v] = b + 2 * g->nreads[v] - g->nadded; bp[v] = p;
        if (best[v] > sb) { sb = best[v]; vbest = v; }
    }
    int len = 0;
    for (int v = vbest; v >= 0; v = bp[v]) ++len;
    if (len > cap) { free(best); free(bp); return -1; }
    int k = len;
    if (g_path_align) { free(pa_posv); pa_posv = (int32_t *)malloc(sizeof(int32_t) * n); pa_nv = n; for (int v = 0; v < n; ++v) pa_posv[v] = -1; }
    for (int v = vbest; v >= 0; v = bp[v]) { draft[--k] = g->base[v]; if (g_path_align) pa_posv[v] = k; }
    free(best); free(bp);
    return len;
}

static void orient(const uint8_t *b, const uint8_t *pw, int L, int rev, uint8_t *ob, uint8_t *opw)
{
    if (!rev) { for (int i = 0; i < L; ++i) ob[i] = (uint8_t)(b[i] & 3); if (opw) memcpy(opw, pw, L); }
    else for (int i = 0; i < L; ++i) { ob[i] = (uint8_t)((3 - b[L - 1 - i]) & 3); if (opw) opw[i] = pw[L - 1 - i]; }
}

/* step 2: draft from the first min(nreads, max_poa_cov) reads.  Orientation = that of read 0.
 * returns draft length, 0 = failure                                                                          */
/* backbone = pass bb; passes bb, bb+1, ... (wrapping) are threaded, max_poa_cov of them; orientation = that of pass bb */
int orc_poa_draft_bb(int nreads, const int64_t *base_off, const uint8_t *bases, const uint8_t *flags, int max_poa_cov,
                     int vcap, uint8_t *draft, int draft_cap, int bb)
{
    int npoa = nreads < max_poa_cov ? nreads : max_poa_cov;
    if (npoa <= 0) return 0;
    int maxL = 0;
    for (int r = 0; r < nreads; ++r) { int L = (int)(base_off[r + 1] - base_off[r]); if (L > maxL) maxL = L; }
    poa_t *g = poa_new(vcap);
    uint8_t *ob = (uint8_t *)malloc(maxL + 1);
    int32_t *pathv = (int32_t *)malloc(sizeof(int32_t) * (maxL + 1));
    int rev0 = flags[bb] & 1, ok = 1;
    g_poa_nscores = 0;
    if (g_path_align) pa_reset();
    for (int rr = 0; rr < npoa && ok; ++rr) {
        int r = bb + rr < nreads ? bb + rr : bb + rr - nreads;
        pa_cur_read = r;
        int L = (int)(base_off[r + 1] - base_off[r]);
        orient(bases + base_off[r], NULL, L, (flags[r] & 1) != rev0, ob, NULL);
        g_bw = g_poa_band; g_cells_kind = CNT_CELLS_POA;            /* SPEC: the POA runs in a POA_BAND-row band */
        if (poa_add_read(g, ob, L, pathv, rr == 0) < 0) ok = 0;
        g_bw = BAND; g_cells_kind = CNT_CELLS_ALIGN;
    }
    int len = ok ? poa_consensus(g, draft, draft_cap) : 0;
    if (len < 0) len = 0;
    free(ob); free(pathv); poa_free(g);
    return len;
}
int orc_poa_draft(int nreads, const int64_t *base_off, const uint8_t *bases, const uint8_t *flags, int max_poa_cov,
                  int vcap, uint8_t *draft, int draft_cap)
{
    return orc_poa_draft_bb(nreads, base_off, bases, flags, max_poa_cov, vcap, draft, draft_cap, 0);
}

/* step 3: read (draft orientation) vs draft, global, adaptive band.  rstart[0..Ld]; returns 1 if valid.
 * dirty (optional, [Ld]): the pile-up evidence of the candidate filter (docs/how-does-ccs-work.md:80-83) —
 * dirty[p] = 1 iff the optimal path does not pass draft position p by a plain matching DIAG step: a mismatch or a
 * deletion marks p; a read base inserted between positions p-1 and p marks both neighbours.                        */
static int align_ev_band(const uint8_t *r, int I, const uint8_t *d, int Ld, int32_t *rstart, int32_t *score_out, uint8_t *dirty,
                         const int32_t *need, int nneed, int *sat_out);
/* SPEC "alignment cascade": the banded global alignment is first tried with ALIGN_BAND1 rows (the band follows the best row, so
 * this finds the same path as the wide band unless an indel run of more than ~ALIGN_BAND1/2 rows occurs); a pass that is not valid
 * in the narrow band is aligned again with BAND rows (and, failing that, by the split alignment).                             */
int orc_windows(const uint8_t *d, int Ld, int32_t *b, int cap);
/* SPEC v5 "band saturation": a pass whose narrow-band alignment is valid is STILL aligned again with BAND rows when the narrow band
 * shows that it could not hold the path — (a) in some column the best row sits in the band's last SAT_ROWS rows while the band has
 * not reached the read's end (the path wants to leave downwards: an insertion run the band cannot take in one step), or (b) between
 * two window-edge columns of the same kind (need[k-2] -> need[k], one window apart) the column maximum did not grow by SAT_GAIN
 * (a window's worth of columns without net score: the band sits on the wrong phase of a repeat).  On low-complexity templates the
 * narrow band otherwise locks onto a wrong repeat phase and still passes the 1.0-per-base gate (profiles/r04_lowcx_band.txt);
 * wide = 1 (opts.disable_heuristics) skips the narrow attempt altogether.  need[0] = 0 < ... < need[nneed-1] = Ld.               */
int orc_align_ev_w(const uint8_t *r, int I, const uint8_t *d, int Ld, const int32_t *need, int nneed, int wide,
                   int32_t *rstart, int32_t *score_out, uint8_t *dirty)
{
    if (!wide && g_align_band1 < BAND) {
        int sat = 0;
        g_bw = g_align_band1;
        int v = align_ev_band(r, I, d, Ld, rstart, score_out, dirty, need, nneed, &sat);
        g_bw = BAND;
        if (v && !sat) return 1;
        orc_cnt[CNT_RETRY64] += 1;
        if (v) orc_cnt[CNT_SATURATED] += 1;
    }
    g_bw = BAND;
    return align_ev_band(r, I, d, Ld, rstart, score_out, dirty, NULL, 0, NULL);
}
/* the same with the window-edge columns derived from the draft (step 4 depends on the draft only) */
int orc_align_ev(const uint8_t *r, int I, const uint8_t *d, int Ld, int32_t *rstart, int32_t *score_out, uint8_t *dirty)
{
    int wcap0 = Ld / (WIN_CORE - 3) + 4, nneed = 0;
    int32_t *wb0 = (int32_t *)malloc(sizeof(int32_t) * wcap0), *need = (int32_t *)malloc(sizeof(int32_t) * 2 * wcap0);
    int nw0 = orc_windows(d, Ld, wb0, wcap0);
    need[nneed++] = 0;
    for (int w = 1; w < nw0; ++w) { need[nneed++] = wb0[w] - WIN_OVH; need[nneed++] = wb0[w] + WIN_OVH; }
    need[nneed++] = Ld;
    int v = orc_align_ev_w(r, I, d, Ld, need, nneed, 0, rstart, score_out, dirty);
    free(wb0); free(need);
    return v;
}
static int align_ev_band(const uint8_t *r, int I, const uint8_t *d, int Ld, int32_t *rstart, int32_t *score_out, uint8_t *dirty,
                         const int32_t *need, int nneed, int *sat_out)
{
    int32_t *lo = (int32_t *)malloc(sizeof(int32_t) * (Ld + 1));
    uint8_t *mv = (uint8_t *)malloc((size_t)(Ld + 1) * BAND);
    int32_t A[BAND], B[BAND], *prevM = A, *curM = B;
    start_column(I, prevM);
    lo[0] = 0; int32_t cm = 0, br = 0; int plo = 0;
    int sat = 0, kn = 1; int32_t cmE[2] = { 0, 0 };        /* column maxima at the last two window-edge columns (edge 0 = column 0: 0) */
    for (int j = 1; j <= Ld; ++j) {
        int l0 = band_lo(plo, br, I);
        int32_t plos[1] = { plo }; const int32_t *pMs[1] = { prevM };
        dp_column(d[j - 1], r, I, l0, 1, plos, pMs, curM, mv + (size_t)j * BAND, &cm, &br);
        lo[j] = l0; plo = l0;
        if (sat_out) {                                      /* SPEC v5 "band saturation" (narrow band only) */
            if (br - l0 >= g_bw - g_sat_rows && l0 + g_bw - 1 < I) sat = 1;
            if (kn < nneed && j == need[kn]) {
                if (kn >= 2 && cm - cmE[kn & 1] < g_sat_gain) sat = 1;
                cmE[kn & 1] = cm; ++kn;
            }
        }
        int32_t *t = prevM; prevM = curM; curM = t;
    }
    if (sat_out) *sat_out = sat;
    int o = I - lo[Ld];
    int valid = (o >= 0 && o < g_bw && prevM[o] > NEG / 2);
    int32_t sc = valid ? prevM[o] : NEG;
    if (valid && sc < Ld) valid = 0;                        /* SPEC: alignment score must reach 1.0 per draft base */
    if (score_out) *score_out = sc;
    if (valid) {
        /* rstart[j] = row at which the optimal path ENTERS column j (read bases consumed when draft
         * position j becomes the next one): insertions emitted while waiting at state j follow it. */
        if (dirty) memset(dirty, 0, Ld);
        int i = I, j = Ld;
        while (j > 0) {
            int t = mv[(size_t)j * BAND + (i - lo[j])] & 3;
            if (t == MV_INS) {                              /* read base i-1 emitted while waiting in column j */
                if (dirty) { dirty[j - 1] = 1; if (j < Ld) dirty[j] = 1; }
                --i; continue;
            }
            rstart[j] = i;
            if (t == MV_DIAG) { if (dirty && d[j - 1] != r[i - 1]) dirty[j - 1] = 1; --i; }
            else if (dirty) dirty[j - 1] = 1;               /* deletion of draft position j-1 */
            --j;
        }
        if (i > 0 && dirty) dirty[0] = 1;                   /* leading insertions */
        rstart[0] = 0;
    }
    free(lo); free(mv);
    return valid;
}
int orc_align(const uint8_t *r, int I, const uint8_t *d, int Ld, int32_t *rstart, int32_t *score_out)
{
    return orc_align_ev(r, I, d, Ld, rstart, score_out, NULL);
}

/* SPEC "split alignment" (the rescue of a pass that carries an insertion the 64-row band cannot follow: "spurious sequencing
 * activity", docs/how-does-ccs-work.md:74-78).  The same banded recurrence runs forward (read prefix against draft prefix) and on
 * the reversed read and draft (suffix against suffix); the pass is split at the interior window-edge column s that maximises
 * colmax_F(s) + colmax_R(Ld - s) with bestrow_F(s) + bestrow_R(Ld - s) <= I (ties: the smallest s; s = 0 and s = Ld, where one
 * half is empty with score 0, stand for a block before the first / after the last aligned base); the read rows in between are
 * the insertion.  Valid iff that sum reaches Ld.  Entry rows of the window-edge columns <= s come from the forward path that
 * ends in (s, bestrow_F(s)), those > s from the reverse path: rstart[c] = I - rstartR[Ld - c].  Every draft position counts as
 * dirty for such a pass (the candidate filter gets no evidence from it).  need[0] = 0 < ... < need[nneed-1] = Ld.            */
static void dp_all_columns(const uint8_t *r, int I, const uint8_t *d, int Ld, int32_t *lo, uint8_t *mv, int32_t *cmc, int32_t *brc)
{
    int32_t A[BAND], B[BAND], *prevM = A, *curM = B;
    start_column(I, prevM);
    lo[0] = 0; cmc[0] = 0; brc[0] = 0;
    int32_t cm = 0, br = 0; int plo = 0;
    for (int j = 1; j <= Ld; ++j) {
        int l0 = band_lo(plo, br, I);
        int32_t plos[1] = { plo }; const int32_t *pMs[1] = { prevM };
        dp_column(d[j - 1], r, I, l0, 1, plos, pMs, curM, mv + (size_t)j * BAND, &cm, &br);
        lo[j] = l0; plo = l0; cmc[j] = cm; brc[j] = br;
        int32_t *t = prevM; prevM = curM; curM = t;
    }
}
static void trace_entries(const uint8_t *mv, const int32_t *lo, int j, int i, int32_t *rstart)
{
    while (j > 0) {
        int t = mv[(size_t)j * BAND + (i - lo[j])] & 3;
        if (t == MV_INS) { --i; continue; }
        rstart[j] = i;
        if (t == MV_DIAG) --i;
        --j;
    }
    rstart[0] = 0;
}
int orc_align_rescue(const uint8_t *r, int I, const uint8_t *d, int Ld, const int32_t *need, int nneed,
                     int32_t *rstart, int32_t *score_out, uint8_t *dirty)
{
    int valid = 0;
    uint8_t *rr = (uint8_t *)malloc(I + 1), *dr = (uint8_t *)malloc(Ld + 1);
    for (int i = 0; i < I; ++i) rr[i] = r[I - 1 - i];
    for (int j = 0; j < Ld; ++j) dr[j] = d[Ld - 1 - j];
    int32_t *loF = (int32_t *)malloc(sizeof(int32_t) * (Ld + 1) * 6), *cmF = loF + (Ld + 1), *brF = cmF + (Ld + 1);
    int32_t *loR = brF + (Ld + 1), *cmR = loR + (Ld + 1), *brR = cmR + (Ld + 1);
    uint8_t *mvF = (uint8_t *)malloc((size_t)(Ld + 1) * BAND * 2), *mvR = mvF + (size_t)(Ld + 1) * BAND;
    int32_t *rsR = (int32_t *)malloc(sizeof(int32_t) * (Ld + 1));
    dp_all_columns(r, I, d, Ld, loF, mvF, cmF, brF);
    dp_all_columns(rr, I, dr, Ld, loR, mvR, cmR, brR);
    int32_t best = NEG; int ks = -1;
    for (int k = 0; k < nneed; ++k) {                       /* k = 0 / nneed-1: the whole pass is suffix / prefix (a block at its very start / end) */
        int s = need[k];
        if (cmF[s] < NEG / 2 || cmR[Ld - s] < NEG / 2 || brF[s] + brR[Ld - s] > I) continue;
        int32_t tot = cmF[s] + cmR[Ld - s];
        if (tot > best) { best = tot; ks = k; }
    }
    if (score_out) *score_out = best;
    if (ks >= 0 && best >= Ld) {
        valid = 1;
        int s = need[ks];
        orc_cnt[CNT_SPLIT] += 1; if (s == 0) orc_cnt[CNT_SPLIT_S0] += 1; if (s == Ld) orc_cnt[CNT_SPLIT_SLD] += 1;
        for (int j = 0; j <= Ld; ++j) rstart[j] = -1;
        trace_entries(mvF, loF, s, brF[s], rstart);
        for (int j = 0; j <= Ld - s; ++j) rsR[j] = -1;
        trace_entries(mvR, loR, Ld - s, brR[Ld - s], rsR);
        for (int c = s + 1; c <= Ld; ++c) rstart[c] = I - rsR[Ld - c];
        if (dirty) memset(dirty, 1, Ld);
    } else {
        /* SPEC "double split" (v4; docs/how-does-ccs-work.md:74-78 speaks of large insertionS): no single split column carries the
         * pass — e.g. two insertions the band cannot follow.  The forward alignment is good up to its first large insertion, the
         * reverse one back to the last: the pass is used as a PREFIX up to the window-edge column s1 with the largest forward column
         * maximum and a SUFFIX from the edge column s2 with the largest reverse column maximum (first on ties in the direction of
         * each DP, as for partial passes), iff s1 < s2, the two parts do not share read rows (bestrow_F(s1) + bestrow_R(Ld - s2) <= I)
         * and each scores at least 1.0 per covered draft base.  The edge columns in between get entry rows that make every window
         * touching them unusable for this pass (a negative segment, or one longer than the pass); every position counts as dirty. */
        int32_t bF = NEG, bR = NEG; int s1 = -1, s2r = -1;
        for (int k = 1; k < nneed; ++k) { int c = need[k]; if (cmF[c] > NEG / 2 && cmF[c] > bF) { bF = cmF[c]; s1 = c; } }
        for (int k = 1; k < nneed; ++k) { int c = Ld - need[nneed - 1 - k]; if (cmR[c] > NEG / 2 && cmR[c] > bR) { bR = cmR[c]; s2r = c; } }
        if (s1 > 0 && s2r > 0 && s1 < Ld - s2r && bF >= s1 && bR >= s2r && brF[s1] + brR[s2r] <= I) {
            valid = 1;
            const int s2 = Ld - s2r;
            if (score_out) *score_out = bF + bR;
            orc_cnt[CNT_SPLIT2] += 1;
            for (int j = 0; j <= Ld; ++j) rstart[j] = -1;
            trace_entries(mvF, loF, s1, brF[s1], rstart);
            for (int j = 0; j <= s2r; ++j) rsR[j] = -1;
            trace_entries(mvR, loR, s2r, brR[s2r], rsR);
            for (int c = s2; c <= Ld; ++c) rstart[c] = I - rsR[Ld - c];
            for (int c = s1 + 1; c < s2; ++c) rstart[c] = -(1 << 20) - c;
            if (dirty) memset(dirty, 1, Ld);
        }
    }
    free(rr); free(dr); free(loF); free(mvF); free(rsR);
    return valid;
}

/* SPEC "partial passes" (docs/faq/accuracy-vs-passes.md:26-29: the first and last subread of a ZMW are not flanked by adapters on both
 * sides; they are not used for the draft and do not count as passes, but the polish uses them where they reach: ec ~ np + 1).  A partial
 * pass is anchored at ONE end of the draft — from_end = 0: it starts where the draft starts and stops somewhere, from_end = 1: it ends
 * where the draft ends.  The banded recurrence of step 3 runs from the anchored end (on the reversed read and draft for from_end); the
 * pass covers the draft up to the window-edge column with the largest column maximum (first on ties: the score rises while the pass
 * lasts and falls by one deletion per column after its last base); valid iff that score reaches 1.0 per covered draft base.  Entry
 * rows of the covered edge columns come from the path that ends in that column's best cell; uncovered columns get entry rows that
 * make every window touching them unusable (a negative segment length); like a split pass it gives the candidate filter no evidence. */
int orc_align_partial(const uint8_t *r, int I, const uint8_t *d, int Ld, const int32_t *need, int nneed, int from_end,
                      int32_t *rstart, int32_t *score_out, uint8_t *dirty)
{
    uint8_t *rr = (uint8_t *)malloc(I + 1), *dr = (uint8_t *)malloc(Ld + 1);
    for (int i = 0; i < I; ++i) rr[i] = from_end ? r[I - 1 - i] : r[i];
    for (int j = 0; j < Ld; ++j) dr[j] = from_end ? d[Ld - 1 - j] : d[j];
    int32_t *lo = (int32_t *)malloc(sizeof(int32_t) * (Ld + 1) * 4), *cm = lo + (Ld + 1), *br = cm + (Ld + 1), *rs = br + (Ld + 1);
    uint8_t *mv = (uint8_t *)malloc((size_t)(Ld + 1) * BAND);
    dp_all_columns(rr, I, dr, Ld, lo, mv, cm, br);
    int32_t best = NEG; int sb = -1;
    for (int k = 1; k < nneed; ++k) {                        /* edge columns in the direction of the DP */
        int s = from_end ? Ld - need[nneed - 1 - k] : need[k];
        if (cm[s] > NEG / 2 && cm[s] > best) { best = cm[s]; sb = s; }
    }
    if (score_out) *score_out = best;
    int valid = (sb > 0 && best >= sb);
    if (valid) {
        for (int j = 0; j <= Ld; ++j) rs[j] = -1;
        trace_entries(mv, lo, sb, br[sb], rs);
        if (!from_end) {
            for (int c = 0; c <= sb; ++c) rstart[c] = rs[c];
            for (int c = sb + 1; c <= Ld; ++c) rstart[c] = -(1 << 20) - c;
        } else {
            for (int c = Ld - sb; c <= Ld; ++c) rstart[c] = I - rs[Ld - c];
            for (int c = 0; c < Ld - sb; ++c) rstart[c] = (1 << 20) + (Ld - c);
        }
        if (dirty) memset(dirty, 1, Ld);
        orc_cnt[CNT_PARTIAL_USED] += 1;
    }
    free(rr); free(dr); free(lo); free(mv);
    return valid;
}

/* step 4: window core boundaries b[0]=0 < ... < b[n]=Ld ; returns n (docs/how-does-ccs-work.md:57-61).
 * "Avoid breaking windows at simple repeats (homopolymers to 4-mer repeats)": a boundary nb is bad when for some
 * period p in 1..4 the p-mer before it equals the p-mer after it; the target boundary cur+22 is moved by
 * 0,+1,-1,+2,-2,+3,-3 to the first good position (all bad: unmoved).                                              */
static int win_bad_break(const uint8_t *d, int Ld, int cur, int nb)
{
    for (int p = 1; p <= 4; ++p) {
        if (nb - p < cur || nb + p > Ld) continue;
        int eq = 1;
        for (int k = 0; k < p && eq; ++k) eq = (d[nb - p + k] == d[nb + k]);
        if (eq) return 1;
    }
    return 0;
}
int orc_windows(const uint8_t *d, int Ld, int32_t *b, int cap)
{
    static const int off[7] = { 0, 1, -1, 2, -2, 3, -3 };
    int n = 0, cur = 0;
    b[0] = 0;
    while (cur < Ld) {
        int nb;
        if (Ld - cur <= WIN_CORE + 6) nb = Ld;
        else {
            nb = cur + WIN_CORE;
            for (int k = 0; k < 7; ++k) if (!win_bad_break(d, Ld, cur, cur + WIN_CORE + off[k])) { nb = cur + WIN_CORE + off[k]; break; }
        }
        if (n + 1 >= cap) return -1;
        b[++n] = nb; cur = nb;
    }
    return n;
}

/* ---------------- steps 8+9: Arrow polish of one window -------------------------------------------------- */
typedef struct { int J, cs, ce, lf, rf; uint8_t t[JMAX + 1]; } wtpl_t;

#define GS (JMAX + 1)                       /* row stride of gamma/beta in the oracle */

static void tpl_ctx(const uint8_t *t, int J, int lf, int *k) { for (int j = 0; j < J; ++j) k[j] = ctx_of(j > 0 ? t[j - 1] : lf, t[j]); }

/* A1/A2: fill gamma (non-stay part of alpha) and beta.  returns alpha(I,J) and beta(0,0) */
static void fill(const float *ME, const float *INS, const float *DL, const uint8_t *t, int J, int lf,
                 const uint8_t *o, int I, float *gam, float *bet, float *aIJ, float *b00)
{
    int k[JMAX + 1]; tpl_ctx(t, J, lf, k);
    orc_cnt[CNT_CELLS_FILL] += 2 * (int64_t)(I + 1) * (J + 1);
    float acol[IMAX + 2], pcol[IMAX + 2];
    memset(pcol, 0, sizeof(pcol));
    int dlo, dhi; fill_band_of(I, J, g_score_band, &dlo, &dhi);   /* SPEC v6: the band of diagonals j - i that exists */
    for (int j = 0; j <= J; ++j) {
        for (int i = 0; i <= I; ++i) {
            float g;
            if (j - i < dlo || j - i > dhi) { gam[i * GS + j] = 0.0f; acol[i] = 0.0f; continue; }
            if (j == 0) g = (i == 0) ? 1.0f : 0.0f;
            else {
                float m = (i > 0) ? pcol[i - 1] * ME[k[j - 1] * NOBS + o[i - 1]] : 0.0f;
                g = mad(pcol[i], DL[k[j - 1]], m);            /* v7: m + pcol[i] * DL */
            }
            gam[i * GS + j] = g;
            acol[i] = (i > 0 && j < J) ? mad(acol[i - 1], INS[k[j] * NOBS + o[i - 1]], g) : g;   /* v7: g + acol[i-1] * INS */
        }
        memcpy(pcol, acol, sizeof(float) * (I + 1));
    }
    *aIJ = pcol[I];
    for (int i = 0; i <= I + 1; ++i) bet[i * GS + J] = (i == I) ? 1.0f : 0.0f;             /* (cell (I, J) lies on diagonal J - I: inside every band) */
    for (int j = J - 1; j >= 0; --j) {
        bet[(I + 1) * GS + j] = 0.0f;
        for (int i = I; i >= 0; --i) {
            if (j - i < dlo || j - i > dhi) { bet[i * GS + j] = 0.0f; continue; }
            float t1 = (i < I) ? ME[k[j] * NOBS + o[i]] * bet[(i + 1) * GS + j + 1] : 0.0f;
            float t12 = (i < I) ? mad(INS[k[j] * NOBS + o[i]], bet[(i + 1) * GS + j], t1) : t1;   /* v7: t1 + INS * beta(i+1, j) */
            bet[i * GS + j] = mad(DL[k[j]], bet[i * GS + j + 1], t12);                               /* v7: (t1 + t2) + DL * beta(i, j+1) */
        }
    }
    *b00 = bet[0];
}

enum { MT_SUB = 0, MT_DEL = 1, MT_INS = 2 };
/* lane index m = slot*32 + c ; slots 0..2 = SUB (+1,+2,+3 mod 4), 3 = DEL, 4..7 = INS A,C,G,T */
static inline int mut_decode(int m, const uint8_t *t, int J, int *type, int *c, int *x)
{
    int slot = m >> 5; *c = m & 31;
    if (slot < 3) { *type = MT_SUB; if (*c >= J) return 0; *x = (t[*c] + 1 + slot) & 3; return 1; }
    if (slot == 3) { *type = MT_DEL; *x = 0; if (*c >= J) return 0; if (*c > 0 && t[*c - 1] == t[*c]) return 0; return 1; }
    *type = MT_INS; *x = slot - 4; if (*c > J) return 0; if (*c > 0 && t[*c - 1] == *x) return 0; return 1;
}

/* A3/A4: likelihood of the read under a virtually mutated template: extend <= 2 alpha columns, link with beta */
static float score_mut(const float *ME, const float *INS, const float *DL, const uint8_t *t, int J, int lf,
                       const uint8_t *o, int I, const float *gam, const float *bet, int type, int c, int x)
{
    int P = (c > 0) ? t[c - 1] : lf;
    int kA = 0, kB = 0, q = 0, fin = 0;
    if (type == MT_SUB)      { kA = ctx_of(P, x); fin = (c + 1 == J); if (!fin) kB = ctx_of(x, t[c + 1]); q = c + 2; }
    else if (type == MT_INS) { kA = ctx_of(P, x); fin = (c == J);     if (!fin) kB = ctx_of(x, t[c]);     q = c + 1; }
    else                     { fin = (c + 1 == J); if (!fin) kA = ctx_of(P, t[c + 1]); kB = kA; q = c + 2; }
    float ap = 0.0f, bp = 0.0f, acc = 0.0f, res = 0.0f;
    /* SPEC "banded link": the extension runs over the read rows around the straight line from (0,0) to (I,J) only (a
     * build-defined approximation of "marginalizing over all possible alignments", docs/how-does-ccs-work.md:94-96: the
     * probability mass off the band is below float resolution for every read the alpha/beta check accepts): half width SCORE_BAND, widened by
     * the part of the length difference a single indel could move the path off that line; rows outside contribute 0.   */
    int dIJ = I > J ? I - J : J - I;
    int Wr = g_score_band + (dIJ > 2 ? dIJ - 2 : 0);
    int nrows = (I < 2 * Wr ? I : 2 * Wr) + 1;
    int rc = (J > 0) ? (2 * c * I + J) / (2 * J) : 0;
    int i0 = rc - Wr; if (i0 < 0) i0 = 0; if (i0 > I + 1 - nrows) i0 = I + 1 - nrows;
    orc_cnt[CNT_CELLS_SCORE] += (type == MT_DEL ? 1 : 2) * (int64_t)nrows;
    /* (gamma / beta outside the band of SPEC v6 are zeros as fill() leaves them: a scoring band clamped into a corner of the window reads such cells) */
#define BANDED(mat, ii, jj) ((mat)[(ii) * GS + (jj)])
    int fdlo, fdhi; fill_band_of(I, J, g_score_band, &fdlo, &fdhi);
    float bq = fin ? 0.0f : BANDED(bet, i0, q);              /* beta(i, q) of the row at hand: the first row's own cell (a zero off the band), then what the row before read */
    const int dmax = fdhi - (q - c - 1);                     /* SPEC v8 "joint band test": the largest diagonal c - i at which gamma(i, c) AND beta(i+1, q) are on the band */
    for (int i = i0; i < i0 + nrows; ++i) {
        const int off = g_clip && (c - i < fdlo || c - i > dmax);   /* (v7: fill() left zeros in the off-band cells, each tested by itself) */
        float insA = 0.0f, meA = 0.0f, insB = 0.0f;
        if (i > 0) {
            if (!(type == MT_DEL && fin)) insA = INS[kA * NOBS + o[i - 1]];
            meA = ME[kA * NOBS + o[i - 1]];
            if (!fin) insB = INS[kB * NOBS + o[i - 1]];
        }
        float a = mad(ap, insA, off ? 0.0f : BANDED(gam, i, c));         /* v7: gamma + ap * insA */
        float b;
        if (type == MT_DEL) b = a;
        else b = mad(bp, insB, mad(a, DL[kA], ap * meA));                /* v7: ((ap * meA) + (a * DL)) + bp * insB */
        if (fin) res = b;
        else {
            float bqn = off ? 0.0f : BANDED(bet, i + 1, q);                /* (row I + 1 of beta is zeros) */
            float t1 = (i < I) ? ME[kB * NOBS + o[i]] * bqn : 0.0f;
            acc = mad(b, mad(DL[kB], bq, t1), acc);                      /* v7: acc + b * (t1 + DL * beta(i, q)) */
            bq = bqn;
        }
        ap = a; bp = b;
    }
#undef BANDED
    return fin ? res : acc;
}

static void tpl_apply(wtpl_t *w, int type, int c, int x)
{
    if (type == MT_SUB) w->t[c] = (uint8_t)x;
    else if (type == MT_INS) {
        memmove(w->t + c + 1, w->t + c, w->J - c); w->t[c] = (uint8_t)x; w->J++;
        if (c < w->cs) { w->cs++; w->ce++; } else if (c < w->ce) w->ce++;
    } else {
        memmove(w->t + c, w->t + c + 1, w->J - c - 1); w->J--;
        if (c < w->cs) { w->cs--; w->ce--; } else if (c < w->ce) w->ce--;
    }
}

static void revcomp_tpl(const wtpl_t *w, uint8_t *tr, int *lfr)
{
    for (int j = 0; j < w->J; ++j) tr[j] = (uint8_t)(3 - w->t[w->J - 1 - j]);
    *lfr = (w->rf < 4) ? 3 - w->rf : 4;
}

/* fixed-point image of one read's log2-likelihood gain (SPEC: the per-mutation sum over reads is an integer sum,
 * so it does not depend on the order in which reads are visited)                                                   */
static inline int32_t dq_fix(float d)
{
    if (d < -DQ_CLAMP) d = -DQ_CLAMP;
    if (d > DQ_CLAMP) d = DQ_CLAMP;
    return (int32_t)floorf(d * DQ_SCALE + 0.5f);
}

/* candidate filter state of a window: bit c of ev = "the pile-up allows skipping template position c"; the bit
 * travels with its base when insertions / deletions are applied (an inserted base is always a candidate)            */
static inline uint32_t ev_insert(uint32_t ev, int c) { uint32_t lowm = (c >= 32) ? 0xffffffffu : ((1u << c) - 1u); return (ev & lowm) | ((ev & ~lowm) << 1); }
static inline uint32_t ev_delete(uint32_t ev, int c) { uint32_t lowm = (1u << c) - 1u; return (ev & lowm) | ((ev >> 1) & ~lowm); }
/* positions skipped this round: evidence bit set and not part of a homopolymer run of the CURRENT template
 * ("homopolymers are always polished", docs/how-does-ccs-work.md:82)                                               */
static uint32_t skip_mask(const wtpl_t *w, uint32_t ev)
{
    uint32_t sk = 0;
    for (int c = 0; c < w->J; ++c) {
        int prev = c > 0 ? w->t[c - 1] : w->lf, next = c + 1 < w->J ? w->t[c + 1] : w->rf;
        int hp = (prev == w->t[c]) || (next == w->t[c]);
        if (((ev >> c) & 1u) && !hp) sk |= 1u << c;
    }
    return sk;
}

static float tract_floor(const uint8_t *v, int n, int x, int np);   /* SPEC v7 "repeat-count floor", defined with skip_perr below */
/* statistics / calibration hooks of the tests (never used by the product, which cannot link this file) */
struct orc_dbg_s {
    int32_t stats, calib;                 /* calib: run unfiltered and record the true p_err of skippable positions */
    int64_t n_scored, n_windows, n_rounds, n_pos, n_evok;
    int64_t cal_cnt[64]; double cal_sum[64]; float cal_max[64];
} orc_dbg;

static __thread uint32_t orc_dbg_calib_ev; static __thread int orc_dbg_margin[JMAX + 1];
/* side channel of polish_window_impl for the whole-ZMW driver: reads [0, pw_nfull) are full-length passes; pw_nvalid_full = how many of
 * them the window used (np counts full-length passes only, ec counts the partial ones too) */
static __thread int pw_nfull = 1 << 30, pw_nvalid_full = 0;
/* the polish seam (include/ccsx.h ccsx_polish_batch; docs/img/ccs-impl.png, docs/faq/revio.md:35-53): a caller-supplied draft replaces the draft cascade
 * (orc_polish_zmw sets these for the duration of one ZMW); qv_only = one scoring round, no mutation applied */
static __thread const uint8_t *g_given_draft = NULL; static __thread int g_given_len = -1, g_given_bb = 0, g_qv_only = 0;
/* Polish one window.  obs[r] = native-orientation observation codes of read r's segment, I[r] its length
 * (I[r] < 0 or > IMAX: read unusable in this window), strand[r] = 1 if the read is reverse to the draft.
 * ev0 = candidate-filter evidence of the draft window (bit c: position c may be skipped), skip_p = error
 * probability reported for a skipped position.
 * Outputs the core sequence, per-base error probability and raw QV.  Returns number of scoring rounds.     */
static int polish_window_impl(const float *ME, const float *INS, const float *DL,
                      const uint8_t *tpl, int J0, int cs, int ce, int lf, int rf,
                      int nreads, const uint8_t *const *obs, const int32_t *I, const uint8_t *strand,
                      uint32_t ev0, const float *skip_p /* [J0] p_err of position c if it is skipped */,
                      const float *MU, const float *VAR, float zmin /* z-score gate; MU == NULL or zmin == 0: off */,
                      uint8_t *out_seq, float *out_perr, float *out_qv, int32_t *out_len,
                      int32_t *out_nvalid, int32_t *out_nonconv, float *out_delta /* [256] optional */,
                      wtpl_t *wfinal /* optional: the converged window template incl. overhangs */,
                      int64_t *out_nscored /* optional: (mutation, read) evaluations */)
{
    wtpl_t w; w.J = J0; w.cs = cs; w.ce = ce; w.lf = lf; w.rf = rf; memcpy(w.t, tpl, J0);
    float *gam = (float *)malloc(sizeof(float) * (size_t)nreads * (IMAX + 2) * GS);
    float *bet = (float *)malloc(sizeof(float) * (size_t)nreads * (IMAX + 2) * GS);
    float *base = (float *)malloc(sizeof(float) * nreads);
    uint8_t *valid = (uint8_t *)malloc(nreads);
    uint8_t *zdrop = (uint8_t *)calloc(nreads > 0 ? nreads : 1, 1);   /* z-score gate: decided on the draft window (round 0), then kept */
    float delta[256]; uint8_t mvalid[256];
    float pskip[JMAX + 2];                                   /* travels with the bases like ev */
    for (int c = 0; c <= JMAX; ++c) pskip[c] = (skip_p && c < J0) ? skip_p[c] : 0.0f;
    uint32_t ev = ev0, sk = 0;
    int iters = 0, nonconv = 0, nvalid = 0;
    int64_t nscored = 0;
    for (int it = 0; it < MAX_ITER; ++it) {
        uint8_t tr[JMAX + 1]; int lfr; revcomp_tpl(&w, tr, &lfr);
        sk = skip_mask(&w, ev);
        nvalid = 0;
        for (int r = 0; r < nreads; ++r) {
            valid[r] = 0;
            if (I[r] < 0 || I[r] > IMAX) continue;
            float a, b;
            float *g = gam + (size_t)r * (IMAX + 2) * GS, *be = bet + (size_t)r * (IMAX + 2) * GS;
            if (strand[r]) fill(ME, INS, DL, tr, w.J, lfr, obs[r], I[r], g, be, &a, &b);
            else           fill(ME, INS, DL, w.t, w.J, w.lf, obs[r], I[r], g, be, &a, &b);
            if (!(a > TINY_P) || !(b > TINY_P)) continue;
            float la = orc_log2f(a), lb = orc_log2f(b);
            if (fabsf(la - lb) > AB_TOL) continue;
            if (zdrop[r]) continue;
            if (it == 0 && MU && zmin != 0.0f) {             /* z-score gate (round 0 only: a changing read set between rounds can make
                                                                the polish oscillate); the x4 per emitted base is 2 bits per read base */
                int kk[JMAX + 1]; tpl_ctx(strand[r] ? tr : w.t, w.J, strand[r] ? lfr : w.lf, kk);
                float M = 0.0f, V = 0.0f;
                for (int j = 0; j < w.J; ++j) { M = M + MU[kk[j]]; V = V + VAR[kk[j]]; }
                float d = (la - (float)(2 * I[r])) - M;
                if (orc_dbg.stats == 3) { float z = d / sqrtf(V); int bin = (int)floorf(z * 2.0f) + 32; if (bin < 0) bin = 0; if (bin > 63) bin = 63;
                    _Pragma("omp atomic") orc_dbg.cal_cnt[bin] += 1; }
                if (d < 0.0f && d * d > (zmin * zmin) * V) { zdrop[r] = 1; orc_cnt[CNT_ZDROP] += 1; continue; }
            }
            base[r] = la; valid[r] = 1; ++nvalid;
        }
        for (int m = 0; m < 256; ++m) {
            int type, c, x; delta[m] = 0.0f;
            mvalid[m] = (uint8_t)mut_decode(m, w.t, w.J, &type, &c, &x);
            if (mvalid[m]) {                                 /* candidate filter: insertions after the last column follow J-1 */
                int cc = (c < w.J) ? c : w.J - 1;
                if ((sk >> cc) & 1u) mvalid[m] = 0;
                /* a quiet position inside a homopolymer keeps only the mutations that change the run's LENGTH (deletion of the
                 * run's first base, insertion of the run's base before it: mut_decode admits them at run starts only)          */
                else if (((ev >> cc) & 1u) && !(c < w.J && (type == MT_DEL || (type == MT_INS && x == w.t[c])))) mvalid[m] = 0;
            }
            if (!mvalid[m]) continue;
            int32_t dsum = 0;
            for (int r = 0; r < nreads; ++r) {
                if (!valid[r]) continue;
                const float *g = gam + (size_t)r * (IMAX + 2) * GS, *be = bet + (size_t)r * (IMAX + 2) * GS;
                float res;
                if (strand[r]) {
                    int cr = (type == MT_INS) ? w.J - c : w.J - 1 - c;
                    res = score_mut(ME, INS, DL, tr, w.J, lfr, obs[r], I[r], g, be, type, cr, 3 - x);
                } else res = score_mut(ME, INS, DL, w.t, w.J, w.lf, obs[r], I[r], g, be, type, c, x);
                float d = orc_log2f(res) - base[r];
                dsum += dq_fix(d);
                ++nscored;
            }
            delta[m] = (float)dsum * (1.0f / DQ_SCALE);
        }
        ++iters;
        if (g_qv_only) break;                                 /* CCSX_QV_ONLY: the gains of the sequence as given are all that is wanted */
        if (orc_dbg.stats == 4) { int nm = 0; for (int m = 0; m < 256; ++m) nm += mvalid[m]; if (nm > 255) nm = 255;
            _Pragma("omp atomic") orc_dbg.cal_cnt[nm >> 2] += 1; }
        /* A5: greedy selection of favourable, well-separated mutations */
        int acc_m[32], nacc = 0, Jn = w.J, nfav = 0;
        uint8_t cand[256];
        for (int m = 0; m < 256; ++m) { cand[m] = (uint8_t)(mvalid[m] && delta[m] > MUT_EPS); nfav += cand[m]; }
        if (it < MAX_ITER - 1) {
            for (;;) {
                int bm = -1; float bd = 0.0f;
                for (int m = 0; m < 256; ++m) if (cand[m] && (bm < 0 || delta[m] > bd)) { bm = m; bd = delta[m]; }
                if (bm < 0) break;
                int slot = bm >> 5, c = bm & 31;
                cand[bm] = 0;
                if (slot >= 4 && Jn >= JMAX) continue;
                if (slot == 3 && Jn <= JMIN_DEL + 1) continue;
                if (slot >= 4) ++Jn; else if (slot == 3) --Jn;
                acc_m[nacc++] = bm;
                if (it >= MULTI_ROUNDS) break;
                for (int m = 0; m < 256; ++m) if (cand[m]) { int d = (m & 31) - c; if (d < 0) d = -d; if (d < MUT_SEP) cand[m] = 0; }
            }
        } else nonconv = (nfav > 0);
        if (nacc == 0) break;
        /* apply in descending position order */
        for (int a = 0; a < nacc; ++a) for (int b = a + 1; b < nacc; ++b)
            if ((acc_m[b] & 31) > (acc_m[a] & 31)) { int t = acc_m[a]; acc_m[a] = acc_m[b]; acc_m[b] = t; }
        for (int a = 0; a < nacc; ++a) {
            int type = 0, c = 0, x = 0; mut_decode(acc_m[a], w.t, w.J, &type, &c, &x);
            if (orc_dbg.stats == 2) {
                fprintf(stderr, "  APPLY it %d type %d c %d x %d delta %.2f cs %d ce %d sk %08x tpl ", it, type, c, x, delta[acc_m[a]], w.cs, w.ce, sk);
                for (int q = 0; q < w.J; ++q) fputc("ACGT"[w.t[q]], stderr);
                fputc('\n', stderr);
            }
            if (orc_dbg_calib_ev && it == 0) {
                wtpl_t w0 = w; uint32_t skc = skip_mask(&w0, orc_dbg_calib_ev);
                int cc = c < w.J ? c : w.J - 1;
                if ((skc >> cc) & 1u) {
                    fprintf(stderr, "MISS type %d c %d x %d delta %.2f J %d tpl ", type, c, x, delta[acc_m[a]], w.J);
                    for (int q = 0; q < w.J; ++q) fputc("ACGT"[w.t[q]], stderr);
                    fprintf(stderr, " margins:");
                    for (int q = 0; q < w.J; ++q) fprintf(stderr, " %d", orc_dbg_margin[q]);
                    fputc('\n', stderr);
                }
            }
            if (type == MT_INS) { ev = ev_insert(ev, c); memmove(pskip + c + 1, pskip + c, sizeof(float) * (JMAX - c)); pskip[c] = 0.0f; }
            else if (type == MT_DEL) { ev = ev_delete(ev, c); memmove(pskip + c, pskip + c + 1, sizeof(float) * (JMAX - c)); }
            tpl_apply(&w, type, c, x);
            for (int q = c - SKIP_SPREAD; q <= c + SKIP_SPREAD; ++q) if (q >= 0 && q < 32) ev &= ~(1u << q);   /* re-open the neighbourhood */
        }
    }
    /* A6: QVs from the last scoring round */
    int len = 0;
    uint8_t vis[JMAX + 3]; int nvis = 0, voff = 0;           /* SPEC v7: the window template with its flanks, for the repeat-count floor */
    if (w.lf < 4) { vis[nvis++] = (uint8_t)w.lf; voff = 1; }
    for (int j = 0; j < w.J; ++j) vis[nvis++] = w.t[j];
    if (w.rf < 4) vis[nvis++] = (uint8_t)w.rf;
    for (int c = w.cs; c < w.ce; ++c) {
        float p;
        if ((sk >> c) & 1u) p = pskip[c];                    /* skipped: error probability from the pile-up margin */
        else {
            float s = ((ev >> c) & 1u) ? pskip[c] : 0.0f;   /* quiet homopolymer position: the untested mutations */
            for (int slot = 0; slot < 8; ++slot) {
                int m = slot * 32 + c;
                if (mvalid[m]) { float d = delta[m]; if (d > 20.0f) d = 20.0f; s = s + orc_exp2f(d); }
            }
            if (c == w.J - 1) for (int slot = 4; slot < 8; ++slot) {
                int m = slot * 32 + w.J;
                if (mvalid[m]) { float d = delta[m]; if (d > 20.0f) d = 20.0f; s = s + orc_exp2f(d); }
            }
            p = s / (1.0f + s);
        }
        if (orc_dbg.calib) { if (p < 1e-10f) p = 1e-10f; }   /* (the calibration hook of tests/test_oracle_filter.py measures the HMM's own value) */
        else { float fl = tract_floor(vis, nvis, c + voff, nvalid); if (p < fl) p = fl; }
        if (!orc_dbg.calib && p < PERR_FLOOR) p = PERR_FLOOR;                /* SPEC v7: no base claims more than Q50 — nothing measured supports a higher claim (profiles/r05_qv_calibration.txt) */
        float qv = -3.01029996f * orc_log2f(p);
        if (qv < 0.0f) qv = 0.0f;
        if (qv > 93.0f) qv = 93.0f;
        out_seq[len] = w.t[c]; out_perr[len] = p; out_qv[len] = qv; ++len;
    }
    *out_len = len; *out_nvalid = nvalid; *out_nonconv = nonconv;
    pw_nvalid_full = 0;
    for (int r = 0; r < nreads && r < pw_nfull; ++r) pw_nvalid_full += valid[r];
    orc_cnt[CNT_NONCONV_WIN] += nonconv;
    if (out_delta) memcpy(out_delta, delta, sizeof(delta));
    if (wfinal) *wfinal = w;
    if (out_nscored) *out_nscored = nscored;
    free(gam); free(bet); free(base); free(valid); free(zdrop);
    return iters;
}

int orc_polish_window(const float *ME, const float *INS, const float *DL,
                      const uint8_t *tpl, int J0, int cs, int ce, int lf, int rf,
                      int nreads, const uint8_t *const *obs, const int32_t *I, const uint8_t *strand,
                      uint8_t *out_seq, float *out_perr, float *out_qv, int32_t *out_len,
                      int32_t *out_nvalid, int32_t *out_nonconv, float *out_delta /* [256] optional */)
{
    return polish_window_impl(ME, INS, DL, tpl, J0, cs, ce, lf, rf, nreads, obs, I, strand, 0u, NULL, NULL, NULL, 0.0f, out_seq, out_perr, out_qv,
                              out_len, out_nvalid, out_nonconv, out_delta, NULL, NULL);
}
/* the same with the candidate filter: ev0 bit c = position c may be skipped, skip_p[c] its reported p_err */
int orc_polish_window_ev(const float *ME, const float *INS, const float *DL,
                      const uint8_t *tpl, int J0, int cs, int ce, int lf, int rf,
                      int nreads, const uint8_t *const *obs, const int32_t *I, const uint8_t *strand,
                      uint32_t ev0, const float *skip_p,
                      uint8_t *out_seq, float *out_perr, float *out_qv, int32_t *out_len,
                      int32_t *out_nvalid, int32_t *out_nonconv, float *out_delta /* [256] optional */)
{
    return polish_window_impl(ME, INS, DL, tpl, J0, cs, ce, lf, rf, nreads, obs, I, strand, ev0, skip_p, NULL, NULL, 0.0f, out_seq, out_perr, out_qv,
                              out_len, out_nvalid, out_nonconv, out_delta, NULL, NULL);
}

/* ---------------- N4: HiFi kinetics (docs/faq/kinetics.md:8-18, tags docs/faq/bam-output.md:13-23) ------------
 * SPEC.  pw / ipd inputs are CodecV1 codes (the u8 stored in the BAM ip/pw tags).  For every
 * window, every read with a usable segment (0 <= I <= IMAX) is aligned globally to the CONVERGED window template in
 * the read's own orientation (reverse-strand reads against the reverse complement) with the integer scores of the
 * draft stage; a DIAG move whose bases agree attributes that read base's decoded IPD / PW frames to the template
 * column.  Per strand and per core position the frames are averaged (integer, round half up) and re-encoded.      */
int orc_codec_v1_decode(int c)
{
    return c < 64 ? c : (c < 128 ? 64 + (c - 64) * 2 : (c < 192 ? 192 + (c - 128) * 4 : 448 + (c - 192) * 8));
}
int orc_codec_v1_encode(int f)          /* nearest representable value, ties up, clamp at 952 */
{
    if (f < 0) f = 0;
    if (f < 64) return f;
    if (f < 192) return 64 + (f - 64 + 1) / 2;
    if (f < 448) return 128 + (f - 192 + 2) / 4;
    int c = 192 + (f - 448 + 4) / 8;
    return c > 255 ? 255 : c;
}

/* one read on one window.  t = template in the READ's orientation (J columns), rb/ipd/pwc = the read segment
 * (native orientation, I bases).  sums are indexed by FORWARD window column: jf = strand ? J-1-j : j.
 * Alignment: H[i][0] = i*INS, H[0][j] = j*DEL;  h = max(diag, left), cell = max(h, up);
 * move = UP iff up > h, else DIAG iff diag >= left, else LEFT.  Traceback from (I,J).                              */
void orc_kinetics_read(const uint8_t *t, int J, const uint8_t *rb, const uint8_t *ipd, const uint8_t *pwc, int I,
                       int strand, uint32_t *sum_ipd /* [JMAX+1] */, uint32_t *sum_pw, uint32_t *cnt)
{
    static const int MV_D = 0, MV_L = 1, MV_U = 2;
    int32_t H[IMAX + 1][JMAX + 1]; uint8_t mv[IMAX + 1][JMAX + 1];
    for (int i = 0; i <= I; ++i) { H[i][0] = i * SC_INS; mv[i][0] = (uint8_t)MV_U; }
    for (int j = 1; j <= J; ++j) {
        H[0][j] = H[0][j - 1] + SC_DEL; mv[0][j] = (uint8_t)MV_L;
        for (int i = 1; i <= I; ++i) {
            int diag = H[i - 1][j - 1] + ((rb[i - 1] & 3) == t[j - 1] ? SC_MATCH : SC_MISMATCH);
            int left = H[i][j - 1] + SC_DEL;
            int h = diag >= left ? diag : left;
            int up = H[i - 1][j] + SC_INS;
            if (up > h) { H[i][j] = up; mv[i][j] = (uint8_t)MV_U; }
            else { H[i][j] = h; mv[i][j] = (uint8_t)(diag >= left ? MV_D : MV_L); }
        }
    }
    int i = I, j = J;
    while (i > 0 || j > 0) {
        int m = mv[i][j];
        if (m == MV_D) {
            if ((rb[i - 1] & 3) == t[j - 1]) {
                int jf = strand ? J - j : j - 1;
                sum_ipd[jf] += (uint32_t)orc_codec_v1_decode(ipd[i - 1]);
                sum_pw[jf] += (uint32_t)orc_codec_v1_decode(pwc[i - 1]);
                cnt[jf] += 1;
            }
            --i; --j;
        } else if (m == MV_L) --j;
        else --i;
    }
}

static inline uint8_t kin_mean_code(uint32_t sum, uint32_t cnt)
{
    if (!cnt) return 0;
    return (uint8_t)orc_codec_v1_encode((int)((2u * sum + cnt) / (2u * cnt)));
}

/* banded unit-cost edit distance (accuracy tests: consensus vs the synthetic truth); band = max |i - j| explored */
int orc_edit_distance(const uint8_t *a, int la, const uint8_t *b, int lb, int band)
{
    if (abs(la - lb) > band) return abs(la - lb) > band ? (la > lb ? la : lb) : 0;
    int W = 2 * band + 1, INF = 1 << 29;
    int *prev = (int *)malloc(sizeof(int) * W), *cur = (int *)malloc(sizeof(int) * W);
    for (int k = 0; k < W; ++k) { int j = k - band; prev[k] = (j >= 0 && j <= lb) ? j : INF; }   /* row 0: j = i + k - band */
    for (int i = 1; i <= la; ++i) {
        for (int k = 0; k < W; ++k) {
            int j = i + k - band, v = INF;
            if (j >= 0 && j <= lb) {
                if (j == 0) v = i;
                else {
                    int dg = prev[k] + (a[i - 1] != b[j - 1]);            /* (i-1, j-1) */
                    int up = (k + 1 < W) ? prev[k + 1] + 1 : INF;          /* (i-1, j)   */
                    int lf = (k > 0) ? cur[k - 1] + 1 : INF;               /* (i, j-1)   */
                    v = dg < up ? dg : up; if (lf < v) v = lf;
                }
            }
            cur[k] = v;
        }
        int *t = prev; prev = cur; cur = t;
    }
    int k = lb - la + band, r = (k >= 0 && k < W) ? prev[k] : INF;
    free(prev); free(cur);
    return r;
}

/* Test / evaluation aid (tools/qv_calibration.py: predicted vs empirical QV): the same banded edit-distance alignment with a traceback that
 * marks the positions of `a` (a consensus) that take part in an error against `b` (the truth): a mismatch or an extra base marks that base,
 * a missing base marks the base of `a` that follows the gap (the last base when the gap is at the end).  Returns the edit distance, -1 when
 * the band cannot hold the alignment.  err[la] is written.                                                                              */
int orc_error_positions(const uint8_t *a, int la, const uint8_t *b, int lb, int band, uint8_t *err)
{
    memset(err, 0, la > 0 ? la : 0);
    if (abs(la - lb) > band || la <= 0) return -1;
    int W = 2 * band + 1, INF = 1 << 29;
    int *D = (int *)malloc(sizeof(int) * (size_t)(la + 1) * W);
    for (int k = 0; k < W; ++k) { int j = k - band; D[k] = (j >= 0 && j <= lb) ? j : INF; }
    for (int i = 1; i <= la; ++i) {
        int *prev = D + (size_t)(i - 1) * W, *cur = D + (size_t)i * W;
        for (int k = 0; k < W; ++k) {
            int j = i + k - band, v = INF;
            if (j >= 0 && j <= lb) {
                if (j == 0) v = i;
                else {
                    int dg = prev[k] + (a[i - 1] != b[j - 1]);
                    int up = (k + 1 < W) ? prev[k + 1] + 1 : INF;
                    int lf = (k > 0) ? cur[k - 1] + 1 : INF;
                    v = dg < up ? dg : up; if (lf < v) v = lf;
                }
            }
            cur[k] = v;
        }
    }
    int i = la, j = lb, k = lb - la + band;
    int dist = (k >= 0 && k < W) ? D[(size_t)la * W + k] : INF;
    if (dist >= INF) { free(D); return -1; }
    while (i > 0 || j > 0) {
        k = j - i + band;
        int v = D[(size_t)i * W + k];
        if (i > 0 && j > 0 && D[(size_t)(i - 1) * W + k] + (a[i - 1] != b[j - 1]) == v) { if (a[i - 1] != b[j - 1]) err[i - 1] = 1; --i; --j; }
        else if (i > 0 && k + 1 < W && D[(size_t)(i - 1) * W + k + 1] + 1 == v) { err[i - 1] = 1; --i; }      /* extra base in a */
        else { err[i < la ? i : la - 1] = 1; --j; }                                                               /* base of b missing in a */
    }
    free(D);
    return dist;
}

/* ---------------- first-principles helpers for tests/test_oracle_hmm.py ---------------------------------- */
/* full refill likelihood of an explicit template: returns alpha(I,J) (scaled by 4^I), and beta(0,0) */
void orc_window_likelihood(const float *ME, const float *INS, const float *DL, const uint8_t *t, int J, int lf,
                           const uint8_t *o, int I, float *aIJ, float *b00)
{
    float *gam = (float *)malloc(sizeof(float) * (IMAX + 2) * GS), *bet = (float *)malloc(sizeof(float) * (IMAX + 2) * GS);
    fill(ME, INS, DL, t, J, lf, o, I, gam, bet, aIJ, b00);
    free(gam); free(bet);
}
/* extend+link likelihood of lane m on template t (forward strand) */
float orc_window_mutation_likelihood(const float *ME, const float *INS, const float *DL, const uint8_t *t, int J, int lf,
                                     const uint8_t *o, int I, int m, int32_t *valid, int32_t *type, int32_t *c, int32_t *x)
{
    float *gam = (float *)malloc(sizeof(float) * (IMAX + 2) * GS), *bet = (float *)malloc(sizeof(float) * (IMAX + 2) * GS);
    float a, b, res = 0.0f; int ty = 0, cc = 0, xx = 0;
    fill(ME, INS, DL, t, J, lf, o, I, gam, bet, &a, &b);
    *valid = mut_decode(m, t, J, &ty, &cc, &xx); *type = ty; *c = cc; *x = xx;
    if (*valid) res = score_mut(ME, INS, DL, t, J, lf, o, I, gam, bet, ty, cc, xx);
    free(gam); free(bet);
    return res;
}
/* brute force: double-precision sum over ALL alignment paths by plain recursion on probabilities (unscaled model) */
double orc_bruteforce_likelihood(const float *ME, const float *INS, const float *DL, const uint8_t *t, int J, int lf,
                                 const uint8_t *o, int I)
{
    int k[JMAX + 1]; tpl_ctx(t, J, lf, k);
    double *A = (double *)calloc((size_t)(I + 1) * (J + 1), sizeof(double));
    for (int j = 0; j <= J; ++j) for (int i = 0; i <= I; ++i) {
        double v = (i == 0 && j == 0) ? 1.0 : 0.0;
        if (j > 0 && i > 0) v += A[(i - 1) * (J + 1) + j - 1] * (double)ME[k[j - 1] * NOBS + o[i - 1]];
        if (j > 0) v += A[i * (J + 1) + j - 1] * (double)DL[k[j - 1]];
        if (i > 0 && j < J) v += A[(i - 1) * (J + 1) + j] * (double)INS[k[j] * NOBS + o[i - 1]];
        A[i * (J + 1) + j] = v;
    }
    double r = A[I * (J + 1) + J];
    free(A);
    return r;
}

/* error probability reported for a position the candidate filter skipped, from its pile-up margin g = clean - dirty:
 * every agreeing pass multiplies the odds of each of the position's 8 mutations by well under 2^-3 (calibrated on
 * the unfiltered path: tests/test_oracle_filter.py), so p = 8 * 2^(-3 g), at least the reporting floor.            */
static inline float skip_perr(int g)
{
    if (g < 0) g = 0;
    if (g > 12) g = 12;
    float p = 8.0f * orc_exp2f(-3.0f * (float)g);
    return p < SKIP_PERR_FLOOR ? SKIP_PERR_FLOOR : p;          /* SPEC v7: the pile-up supports no claim beyond Q50 (profiles/r05_qv_calibration.txt) */
}

/* SPEC v7 "repeat-count floor" (docs/faq/low-complexity.md:11-18; docs/how-does-ccs-work.md:103-106: the predicted accuracy is the mean of the per-base QVs, so
 * a base whose error the polish cannot see must not claim a high QV).  Windowed single-base polishing cannot tell n from n +- 1 copies of a tandem repeat's unit
 * when every pass places its own count error somewhere else in the tract; measured on low-complexity templates the consensus carries ~ 0.2 wrong bases per visible
 * tract whatever its length (predicted 4.2 x too few errors before, 1.1 x with the floor; on-model data is unaffected: such tracts do not occur by chance).
 * v[0..n) = the converged window template with its flanking draft bases; a core base inside a period-p tract (p = 1..4: v[i] == v[i+p] on a maximal run) of
 * L >= REP_MINLEN[p-1] visible bases reports p_err >= REP_ERRS / L. */
#define REP_ERRS 0.2f
static const int REP_MINLEN[4] = { 8, 10, 12, 16 };
/* SPEC v8: the floor was calibrated at 10 passes.  Measured by pass count (profiles/r06_tract_floor_by_passes*.txt): a tract that runs into an END of the
 * visible template (an "open" tract: the window cannot see where it stops) carries 0.65 - 0.77 count errors whatever the coverage (3 .. 30 passes) — the floor stays;
 * a tract that begins and ends INSIDE the visible template ("closed": every pass shows the whole run to one window) is resolved by the likelihood, and its
 * errors fall with the square of the coverage (0.135 per tract at 10 passes, 0.06 at 15, 0.015 at 30): its floor is scaled by (REP_NP0 / np)^2 for np > REP_NP0
 * passes used in the window.  orc_set_rep_np0(0) restores the v7 rule. */
#define REP_NP0 10
static int g_rep_np0 = REP_NP0;
void orc_set_rep_np0(int n) { g_rep_np0 = n; }
static float tract_floor(const uint8_t *v, int n, int x, int np)
{
    float fl = 0.0f;
    for (int p = 1; p <= 4; ++p) {
        int best = 0, open = 0;
        for (int k = x - p; k <= x; ++k) {
            if (k < 0 || k + p >= n || v[k] != v[k + p]) continue;
            int i = k, j = k + 1;
            while (i > 0 && v[i - 1] == v[i - 1 + p]) --i;
            while (j + p < n && v[j] == v[j + p]) ++j;
            int L = j - i + p, op = (i == 0) || (j + p >= n);
            if (L > best) { best = L; open = op; } else if (L == best) open |= op;
        }
        if (best >= REP_MINLEN[p - 1]) {
            float f = REP_ERRS / (float)best;
            if (!open && g_rep_np0 > 0 && np > g_rep_np0) { f = f * ((float)(g_rep_np0 * g_rep_np0) / (float)(np * np)); orc_cnt[CNT_CLOSED_TRACT] += 1; }
            if (f > fl) fl = f;
        }
    }
    return fl;
}

/* STUDY HOOK: the step-3 alignment of a threaded pass from its path through the POA graph.  vp[i] = the vertex of read base i, posv[v] = the vertex's position in the
 * draft or -1.  A base on a consensus vertex is a match at that position (vertices carry one base); every other base is an insertion waiting at the column after the
 * last match; consensus positions the pass skips are deletions.  Entry rows and dirty bits as the aligner defines them (DESIGN.md §2 "Alignment"); valid iff the
 * alignment score (+3 / -4 / -4) reaches the draft's length, the aligner's gate. */
static int path_align(const int32_t *vp, int L, const int32_t *posv, int Ld, int32_t *rstart, uint8_t *dirty, int32_t *score_out)
{
    memset(dirty, 1, Ld + 1);
    int lastp = -1, ins = 0, nm = 0, nins = 0;
    rstart[0] = 0;
    for (int i = 0; i < L; ++i) {
        int p = vp[i] >= 0 ? posv[vp[i]] : -1;
        if (p < 0) { ++ins; ++nins; continue; }
        if (p <= lastp) return 0;                            /* (cannot happen: both are paths of one DAG) */
        for (int q = lastp + 2; q <= p; ++q) rstart[q] = i;  /* the skipped columns and column p are entered with base i next; column lastp + 1 was entered right after its match */
        if (lastp < 0 && p >= 1) rstart[1 <= p ? 1 : 0] = rstart[1];   /* (no-op: kept for symmetry) */
        if (lastp < 0) for (int q = 1; q <= p; ++q) rstart[q] = i;
        dirty[p] = 0;
        if (ins) { if (lastp >= 0) dirty[lastp] = 1; if (lastp + 1 <= Ld) dirty[lastp + 1 < Ld ? lastp + 1 : Ld - 1] = 1; if (lastp < 0) dirty[0] = 1; }
        if (p + 1 <= Ld) rstart[p + 1] = i + 1;
        lastp = p; ins = 0; ++nm;
    }
    for (int q = lastp + 2; q <= Ld; ++q) rstart[q] = L;
    if (lastp < 0) for (int q = 1; q <= Ld; ++q) rstart[q] = L;
    if (ins && lastp >= 0) { dirty[lastp] = 1; if (lastp + 1 < Ld) dirty[lastp + 1] = 1; }
    rstart[Ld] = L;
    int ndel = Ld - nm;
    int32_t sc = 3 * nm - 4 * nins - 4 * ndel;
    if (score_out) *score_out = sc;
    return sc >= Ld;
}

/* ---------------- whole-ZMW driver (steps 2,3,4,8,9,10) --------------------------------------------------- */
typedef struct {
    int32_t status, seq_len, np, iters, n_windows;
    float rq, ec;
    int32_t fn, rn;             /* passes used on the strand of SEQ / on the other strand (fn + rn = np) */
} orc_zmw_out;

enum { ST_SUCCESS = 0, ST_TOO_FEW = 1, ST_DRAFT_FAIL = 2, ST_UNUSABLE = 3, ST_NONCONV = 4, ST_SHORT = 5, ST_LONG = 6, ST_LOWRQ = 7, ST_EMPTY = 8, ST_CAPACITY = 9 };

int orc_consensus_zmw_kin(const orc_model *model, const orc_opts *opts, const float *snr, int nreads_in,
                      const int64_t *base_off /* [nreads+1], relative to bases */, const uint8_t *bases, const uint8_t *pw,
                      const uint8_t *flags, uint8_t *seq, uint8_t *qual, float *raw_qv, int64_t cap, orc_zmw_out *out,
                      uint8_t *draft_out, int32_t *draft_len_out,
                      const uint8_t *ipd /* NULL = no kinetics */, uint8_t *fi, uint8_t *fp, uint8_t *ri, uint8_t *rp)
{
    memset(out, 0, sizeof(*out));
    g_perr_floor = perr_floor_of(opts->max_qv);
    int nreads = nreads_in;
    {   /* SPEC v5: at most MAX_PASSES = 255 passes are used (--top-passes 0 = "all"; SPEC v4 stopped at 64) */
        int top = (opts->top_passes <= 0 || opts->top_passes > MAX_PASSES) ? MAX_PASSES : opts->top_passes;
        if (nreads > top) nreads = top;
    }
    /* SPEC "partial passes": flag bit 1; a ZMW's partial passes follow its full-length passes (the batch is validated for that) */
    int nfull = 0;
    while (nfull < nreads && !(flags[nfull] & 2)) ++nfull;
    if (nfull < opts->min_passes || nfull < 1) { out->status = ST_TOO_FEW; return 0; }
    int maxL = 0;
    for (int r = 0; r < nreads; ++r) { int L = (int)(base_off[r + 1] - base_off[r]); if (L > maxL) maxL = L; }
    int dcap = maxL + maxL / 4 + 64;
    uint8_t *draft = (uint8_t *)malloc(dcap);
    int vcap = (5 * maxL) / 2 + 256;
    /* SPEC "fallback draft" (docs/faq/accuracy-vs-passes.md:41-46, a cascade from fast to robust draft generators): when the first
     * draft fails or at most half of the passes map to it, ONE more draft is made with the pass whose length is closest to the
     * median as backbone (ties: the first) and twice as many passes threaded, starting at the backbone and wrapping around.
     * Draft, alignments and the consensus then have the orientation of that backbone pass.                                      */
    int bb = 0, attempt = 0, Ld, rev0, np;
    int32_t **rstart = (int32_t **)calloc(nreads, sizeof(int32_t *));
    uint8_t **dirty = (uint8_t **)calloc(nreads, sizeof(uint8_t *));
    uint8_t *strand = (uint8_t *)malloc(nreads), *avalid = (uint8_t *)malloc(nreads);
    int ret = 0;
    for (;;) {
        if (g_given_len >= 0) {                             /* the polish seam: the caller's draft, in the orientation of pass g_given_bb; no cascade */
            bb = (g_given_bb >= 0 && g_given_bb < nreads) ? g_given_bb : 0;
            Ld = g_given_len > dcap ? 0 : g_given_len;
            for (int q = 0; q < Ld; ++q) draft[q] = g_given_draft[q] & 3;
        } else
        if (attempt < 2) Ld = orc_poa_draft_bb(nfull, base_off, bases, flags, attempt ? 2 * opts->max_poa_cov : opts->max_poa_cov, vcap, draft, dcap, bb);
        else {                                              /* SPEC "draft cascade", last resort: the backbone pass itself is the draft */
            Ld = (int)(base_off[bb + 1] - base_off[bb]);
            if (Ld > dcap) Ld = 0;
            else orient(bases + base_off[bb], NULL, Ld, 0, draft, NULL);
        }
        if (draft_len_out) *draft_len_out = Ld;
        if (draft_out && Ld > 0) memcpy(draft_out, draft, Ld);
        int want_retry = 0;
        np = 0; out->np = 0; out->fn = out->rn = 0;
        if (Ld <= 0) { out->status = ST_DRAFT_FAIL; want_retry = 1; }
        else if (Ld < opts->min_length) { out->status = ST_SHORT; goto done; }
        else if (Ld > opts->max_length) { out->status = ST_LONG; goto done; }
        else {
            /* step 3 */
            rev0 = flags[bb] & 1;
            uint8_t *ob = (uint8_t *)malloc(maxL + 1);
            /* window-edge columns (step 4 depends on the draft only): the split alignment may split a pass at one of them */
            int wcap0 = Ld / (WIN_CORE - 3) + 4, nneed = 0;
            int32_t *wb0 = (int32_t *)malloc(sizeof(int32_t) * wcap0), *need = (int32_t *)malloc(sizeof(int32_t) * 2 * wcap0);
            {
                int nw0 = orc_windows(draft, Ld, wb0, wcap0);
                need[nneed++] = 0;
                for (int w = 1; w < nw0; ++w) { need[nneed++] = wb0[w] - WIN_OVH; need[nneed++] = wb0[w] + WIN_OVH; }
                need[nneed++] = Ld;
            }
            for (int r = 0; r < nreads; ++r) {
                int L = (int)(base_off[r + 1] - base_off[r]);
                strand[r] = (uint8_t)(((flags[r] & 1) != rev0) ? 1 : 0);
                orient(bases + base_off[r], NULL, L, strand[r], ob, NULL);
                free(rstart[r]); free(dirty[r]);
                rstart[r] = (int32_t *)malloc(sizeof(int32_t) * (Ld + 1));
                dirty[r] = (uint8_t *)malloc(Ld + 1);
                int32_t sc;
                if (r >= nfull) {                           /* partial pass: anchored at the draft's start or end (flag bit 2 = the adapter is at
                                                               the pass's END; in draft orientation that end is the draft's end iff same strand) */
                    int from_end = ((flags[r] >> 2) & 1) ^ strand[r];
                    avalid[r] = (uint8_t)(nneed >= 2 ? orc_align_partial(ob, L, draft, Ld, need, nneed, from_end, rstart[r], &sc, dirty[r]) : 0);
                    continue;                               /* not a pass: np / fn / rn count full-length passes */
                }
                int from_path = 0;
                if (g_path_align && attempt < 2 && pa_posv) {    /* STUDY HOOK: a pass that was threaded takes its alignment from its graph path */
                    for (int k = 0; k < pa_n; ++k) if (pa_read[k] == r && pa_len[k] == L) { avalid[r] = (uint8_t)path_align(pa_path[k], L, pa_posv, Ld, rstart[r], dirty[r], &sc); from_path = 1; break; }
                }
                if (!from_path)
                avalid[r] = (uint8_t)orc_align_ev_w(ob, L, draft, Ld, need, nneed, opts->disable_heuristics != 0, rstart[r], &sc, dirty[r]);
                /* a pass much longer than the draft that failed: look for ONE large insertion (SPEC "split alignment") */
                if (!avalid[r] && L - Ld > RESCUE_MIN_EXCESS && nneed >= 3)
                    avalid[r] = (uint8_t)orc_align_rescue(ob, L, draft, Ld, need, nneed, rstart[r], &sc, dirty[r]);
                np += avalid[r];
                if (avalid[r]) { if (strand[r]) out->rn += 1; else out->fn += 1; }
            }
            free(ob); free(wb0); free(need);
            out->np = np;
            if (2 * np <= nfull) { out->status = ST_UNUSABLE; want_retry = 1; }
        }
        if (!want_retry) break;
        if (attempt >= 2 || opts->no_fallback_draft || g_given_len >= 0) { if (out->status == ST_DRAFT_FAIL) { out->np = 0; out->fn = out->rn = 0; } goto done; }
        {   /* backbone of the fallback draft: the pass whose length is closest to the median (ties: the first); the last resort takes
             * the closest among the passes that have NOT been a backbone yet (pass 0 and the fallback's backbone may be the problem) */
            const int bb1 = attempt ? bb : -1;
            int med = 0, best = -1;
            for (int r = 0; r < nfull; ++r) {
                int len = (int)(base_off[r + 1] - base_off[r]), rank = 0;
                for (int q = 0; q < nfull; ++q) { int lq = (int)(base_off[q + 1] - base_off[q]); rank += (lq < len || (lq == len && q < r)) ? 1 : 0; }
                if (rank == nfull / 2) med = len;
            }
            for (int r = 0; r < nfull; ++r) {
                int d = (int)(base_off[r + 1] - base_off[r]) - med; if (d < 0) d = -d;
                if (attempt && nfull > 1 && r == bb1) continue;
                if (attempt && nfull > 2 && r == 0) continue;     /* ... and pass 0, the backbone of the first draft, failed as well */
                if (best < 0 || d < best) { best = d; bb = r; }
            }
        }
        attempt += 1; orc_cnt[attempt == 1 ? CNT_FALLBACK : CNT_THIRD_DRAFT] += 1;
    }
    {
        /* step 4 */
        int wcap = Ld / (WIN_CORE - 3) + 4;
        int32_t *wb = (int32_t *)malloc(sizeof(int32_t) * wcap);
        int nw = orc_windows(draft, Ld, wb, wcap);
        out->n_windows = nw;
        float ME[NCTX * NOBS], INS[NCTX * NOBS], DL[NCTX];
        orc_tables(model, snr, ME, INS, DL);
        float MU[NCTX], VAR[NCTX];
        orc_zparams(model, snr, MU, VAR);
        uint8_t *obuf = (uint8_t *)malloc((size_t)nreads * (IMAX + 1));
        const uint8_t **obs = (const uint8_t **)malloc(sizeof(uint8_t *) * nreads);
        int32_t *Iw = (int32_t *)malloc(sizeof(int32_t) * nreads), *Ikin = (int32_t *)malloc(sizeof(int32_t) * nreads);
        int64_t len = 0; double perr_sum = 0.0; int64_t nvalid_sum = 0; int nonconv_any = 0, overflow = 0;
        int32_t nv_hist[MAX_PASSES + 1]; memset(nv_hist, 0, sizeof(nv_hist));
        for (int w = 0; w < nw; ++w) {
            int ws = wb[w] - WIN_OVH; if (ws < 0) ws = 0;
            int we = wb[w + 1] + WIN_OVH; if (we > Ld) we = Ld;
            int J = we - ws, cs = wb[w] - ws, ce = wb[w + 1] - ws;
            int lf = ws > 0 ? draft[ws - 1] : 4, rf = we < Ld ? draft[we] : 4;
            const int maxins = opts->max_insertion_size == 0 ? 30 : opts->max_insertion_size;
            for (int r = 0; r < nreads; ++r) {
                Iw[r] = -1; Ikin[r] = -1; obs[r] = obuf + (size_t)r * (IMAX + 1);
                if (!avalid[r]) continue;
                int a = rstart[r][ws], b = rstart[r][we], L = (int)(base_off[r + 1] - base_off[r]);
                int n = b - a;
                if (n < 0 || n > L) continue;                /* (entry rows are rows of this pass: anything else is not a segment) */
                if (n <= IMAX) Ikin[r] = n;                  /* the kinetics always see the untrimmed segment */
                int na = strand[r] ? L - b : a;              /* native start of the segment */
                const uint8_t *bb = bases + base_off[r] + na, *pp = pw + base_off[r] + na;
                uint8_t *oo = obuf + (size_t)r * (IMAX + 1);
                if (maxins > 0 && n > J + maxins) {
                    /* SPEC "trim large insertions" (docs/how-does-ccs-work.md:74-78): the segment is cut down to the window's
                     * length: its first s and last J - s bases, s = the split with the most diagonal matches of prefix and
                     * suffix against the window (read orientation; ties: the smallest s)                                    */
                    uint8_t T[JMAX + 1];
                    for (int j = 0; j < J; ++j) T[j] = strand[r] ? (uint8_t)(3 - draft[ws + J - 1 - j]) : draft[ws + j];
                    int best = -1, sb = 0;
                    for (int s = 0; s <= J; ++s) {
                        int m = 0;
                        for (int i = 0; i < s; ++i) m += ((bb[i] & 3) == T[i]);
                        for (int j = s; j < J; ++j) m += ((bb[n - J + j] & 3) == T[j]);
                        if (m > best) { best = m; sb = s; }
                    }
                    for (int i = 0; i < J; ++i) { int src = i < sb ? i : n - J + i; oo[i] = (uint8_t)obs_of(bb[src], pp[src]); }
                    Iw[r] = J; orc_cnt[CNT_TRIM] += 1;
                    continue;
                }
                if (n > IMAX) continue;
                Iw[r] = n;
                for (int i = 0; i < n; ++i) oo[i] = (uint8_t)obs_of(bb[i], pp[i]);
            }
            /* step 7, candidate filter (docs/how-does-ccs-work.md:80-83): pile-up of the step-3 alignments over the
             * reads with a usable segment; position c may be skipped iff clean - dirty >= SKIP_MARGIN           */
            uint32_t ev0 = 0; float skp[JMAX + 1]; int margin[JMAX + 1];
            {
                int nuse = 0;
                for (int r = 0; r < nreads; ++r) if (Iw[r] >= 0) ++nuse;
                for (int c = 0; c < J; ++c) {
                    int nd = 0;
                    for (int r = 0; r < nreads; ++r) if (Iw[r] >= 0) nd += dirty[r][ws + c];
                    margin[c] = nuse - 2 * nd;
                    skp[c] = skip_perr(margin[c]);
                }
                for (int c = 0; c < J; ++c) {
                    int ok = !opts->disable_heuristics && margin[c] >= g_skip_margin;
                    for (int q = c - SKIP_SPREAD; ok && q <= c + SKIP_SPREAD; ++q) if (q >= 0 && q < J && margin[q] < 0) ok = 0;
                    if (ok) ev0 |= 1u << c;
                }
            }
            uint8_t wseq[JMAX + 1]; float wperr[JMAX + 1], wqv[JMAX + 1]; int32_t wlen, wnv, wnc;
            wtpl_t wf; int64_t nsc = 0;
            orc_dbg_calib_ev = (orc_dbg.calib > 2) ? ev0 : 0u; memcpy(orc_dbg_margin, margin, sizeof(int) * J);
            pw_nfull = nfull;
            int it = polish_window_impl(ME, INS, DL, draft + ws, J, cs, ce, lf, rf, nreads, obs, Iw, strand,
                                        orc_dbg.calib ? 0u : ev0, skp, MU, VAR, opts->min_zscore, wseq, wperr, wqv, &wlen, &wnv, &wnc, NULL, &wf, &nsc);
            out->iters += it; nvalid_sum += wnv; nonconv_any |= wnc;
            pw_nfull = 1 << 30;
            if (pw_nvalid_full >= 0 && pw_nvalid_full <= MAX_PASSES) nv_hist[pw_nvalid_full] += 1;
            if (orc_dbg.stats == 2) {
                fprintf(stderr, "WIN %d ws %d we %d cs %d ce %d it %d nv %d ev %08x draft ", w, ws, we, cs, ce, it, wnv, ev0);
                for (int q = 0; q < J; ++q) fputc("ACGT"[draft[ws + q]], stderr);
                fprintf(stderr, " core ");
                for (int q = 0; q < wlen; ++q) fputc("ACGT"[wseq[q]], stderr);
                fprintf(stderr, " margins");
                for (int q = 0; q < J; ++q) fprintf(stderr, " %d", margin[q]);
                fputc('\n', stderr);
            }
            if (orc_dbg.stats) {
#pragma omp critical
                {
                    orc_dbg.n_scored += nsc; orc_dbg.n_windows += 1; orc_dbg.n_rounds += it;
                    for (int c = cs; c < ce; ++c) { orc_dbg.n_pos += 1; if ((ev0 >> c) & 1u) orc_dbg.n_evok += 1; }
                    if (orc_dbg.calib && it == 1 && wlen == ce - cs) {   /* unchanged window: true p_err of skippable positions by margin */
                        wtpl_t w0; w0.J = J; w0.lf = lf; w0.rf = rf; memcpy(w0.t, draft + ws, J);
                        uint32_t sk = skip_mask(&w0, 0xffffffffu);
                        for (int c = cs; c < ce; ++c) if ((sk >> c) & 1u) {
                            int g = margin[c]; if (g < 0) g = 0; if (g > 63) g = 63;
                            orc_dbg.cal_cnt[g] += 1; orc_dbg.cal_sum[g] += wperr[c - cs];
                            if (wperr[c - cs] > orc_dbg.cal_max[g]) orc_dbg.cal_max[g] = wperr[c - cs];
                            if (orc_dbg.calib > 1 && g >= 6 && wperr[c - cs] > 1e-3f) {
                                int nuse = 0; for (int r = 0; r < nreads; ++r) if (Iw[r] >= 0) ++nuse;
                                fprintf(stderr, "CAL margin %d p %.3e c %d J %d nuse %d nvalid %d tpl ", g, wperr[c - cs], c, J, nuse, wnv);
                                for (int q = 0; q < J; ++q) fputc("ACGT"[draft[ws + q]], stderr);
                                fprintf(stderr, "  dirty:");
                                for (int r = 0; r < nreads; ++r) if (Iw[r] >= 0) { fputc(' ', stderr); for (int q = 0; q < J; ++q) fputc(dirty[r][ws + q] ? 'x' : '.', stderr); }
                                fputc('\n', stderr);
                            }
                        }
                    }
                }
            }
            uint32_t ks[2][3][JMAX + 1];                          /* [strand][ipd, pw, count][forward column] */
            if (ipd) {
                memset(ks, 0, sizeof(ks));
                uint8_t tr[JMAX + 1]; int lfr; revcomp_tpl(&wf, tr, &lfr);
                for (int r = 0; r < nreads; ++r) {
                    if (Ikin[r] < 0) continue;
                    int a = rstart[r][ws], b = rstart[r][we], L = (int)(base_off[r + 1] - base_off[r]);
                    int na = strand[r] ? L - b : a;
                    int64_t p0 = base_off[r] + na;
                    (void)a;
                    orc_kinetics_read(strand[r] ? tr : wf.t, wf.J, bases + p0, ipd + p0, pw + p0, Ikin[r], strand[r],
                                      ks[strand[r]][0], ks[strand[r]][1], ks[strand[r]][2]);
                }
            }
            float wsum = 0.0f;
            for (int i = 0; i < wlen; ++i) {
                if (len < cap) {
                    seq[len] = wseq[i]; qual[len] = (uint8_t)(wqv[i] + 0.5f); if (raw_qv) raw_qv[len] = wqv[i];
                    if (ipd) {
                        const int c = wf.cs + i;
                        fi[len] = kin_mean_code(ks[0][0][c], ks[0][2][c]); fp[len] = kin_mean_code(ks[0][1][c], ks[0][2][c]);
                        ri[len] = kin_mean_code(ks[1][0][c], ks[1][2][c]); rp[len] = kin_mean_code(ks[1][1][c], ks[1][2][c]);
                    }
                } else overflow = 1;
                wsum = wsum + wperr[i]; ++len;
            }
            perr_sum += (double)wsum;
        }
        out->rq = len > 0 ? (float)(1.0 - perr_sum / (double)len) : 0.0f;
        if (overflow) len = 0;                                   /* never a silently truncated read: status CAPACITY */
        out->seq_len = (int32_t)len;
        out->ec = nw > 0 ? (float)((double)nvalid_sum / (double)nw) : 0.0f;
        {   /* np = mode over windows of the passes used for polishing (docs/faq/accuracy-vs-passes.md:18-24); ties: the smaller count */
            int best = 0;
            for (int v = 1; v <= MAX_PASSES; ++v) if (nv_hist[v] > nv_hist[best]) best = v;
            out->np = best;
        }
        if (overflow) out->status = ST_CAPACITY;
        else if (len == 0) out->status = ST_EMPTY;
        else if (nonconv_any) out->status = ST_NONCONV;
        else if (out->rq < opts->min_rq) out->status = ST_LOWRQ;
        else out->status = ST_SUCCESS;
        ret = 1;
        free(wb); free(obuf); free(obs); free(Iw); free(Ikin);
    }
done:
    orc_cnt[CNT_ZMWS] += 1;
    orc_counts_flush();
    for (int r = 0; r < nreads; ++r) { free(rstart[r]); free(dirty[r]); }
    free(rstart); free(dirty); free(strand); free(avalid); free(draft);
    return ret;
}

/* the polish seam for one ZMW: alignment cascade + windows + polish + QVs on a caller-supplied draft (flags bit 0 = CCSX_QV_ONLY) */
int orc_polish_zmw(const orc_model *model, const orc_opts *opts, const float *snr, int nreads_in,
                   const int64_t *base_off, const uint8_t *bases, const uint8_t *pw, const uint8_t *flags,
                   const uint8_t *draft, int draft_len, int backbone, int qflags,
                   uint8_t *seq, uint8_t *qual, float *raw_qv, int64_t cap, orc_zmw_out *out)
{
    g_given_draft = draft; g_given_len = draft_len < 0 ? 0 : draft_len; g_given_bb = backbone; g_qv_only = qflags & 1;
    int rc = orc_consensus_zmw_kin(model, opts, snr, nreads_in, base_off, bases, pw, flags, seq, qual, raw_qv, cap, out, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    g_given_draft = NULL; g_given_len = -1; g_qv_only = 0;
    orc_counts_flush();
    return rc;
}

int orc_consensus_zmw(const orc_model *model, const orc_opts *opts, const float *snr, int nreads_in,
                      const int64_t *base_off /* [nreads+1], relative to bases */, const uint8_t *bases, const uint8_t *pw,
                      const uint8_t *flags, uint8_t *seq, uint8_t *qual, float *raw_qv, int64_t cap, orc_zmw_out *out,
                      uint8_t *draft_out, int32_t *draft_len_out)
{
    return orc_consensus_zmw_kin(model, opts, snr, nreads_in, base_off, bases, pw, flags, seq, qual, raw_qv, cap, out,
                                 draft_out, draft_len_out, NULL, NULL, NULL, NULL, NULL);
}

/* batch driver over the ccsx SoA/CSR layout; nthreads > 1 uses OpenMP over ZMWs (cpu_baseline leg of bench.py) */
int orc_consensus_batch_kin(const orc_model *model, const orc_opts *opts, int n_zmw, const float *snr, const int32_t *read_off,
                        const int64_t *base_off, const uint8_t *bases, const uint8_t *pw, const uint8_t *flags,
                        const int64_t *seq_off, int32_t *status, int32_t *seq_len, uint8_t *seq, uint8_t *qual, float *raw_qv,
                        float *rq, int32_t *np, float *ec, int32_t *iters, int32_t *n_windows, int nthreads,
                        const uint8_t *ipd, uint8_t *fi, uint8_t *fp, uint8_t *ri, uint8_t *rp, int32_t *fn, int32_t *rn)
{
    /* keep per-window / per-read scratch on the (per-thread) malloc arenas instead of mmap/munmap per call */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int z = 0; z < n_zmw; ++z) {
        int r0 = read_off[z], nr = read_off[z + 1] - r0;
        int64_t b0 = base_off[r0];
        int64_t *rel = (int64_t *)malloc(sizeof(int64_t) * (nr + 1));
        for (int r = 0; r <= nr; ++r) rel[r] = base_off[r0 + r] - b0;
        orc_zmw_out o;
        orc_consensus_zmw_kin(model, opts, snr + 4 * z, nr, rel, bases + b0, pw + b0, flags + r0,
                          seq + seq_off[z], qual + seq_off[z], raw_qv ? raw_qv + seq_off[z] : NULL,
                          seq_off[z + 1] - seq_off[z], &o, NULL, NULL,
                          ipd ? ipd + b0 : NULL, ipd ? fi + seq_off[z] : NULL, ipd ? fp + seq_off[z] : NULL,
                          ipd ? ri + seq_off[z] : NULL, ipd ? rp + seq_off[z] : NULL);
        status[z] = o.status; seq_len[z] = o.seq_len; rq[z] = o.rq; np[z] = o.np; ec[z] = o.ec; iters[z] = o.iters; n_windows[z] = o.n_windows;
        if (fn) fn[z] = o.fn;
        if (rn) rn[z] = o.rn;
        free(rel);
    }
    return 0;
}

int orc_consensus_batch(const orc_model *model, const orc_opts *opts, int n_zmw, const float *snr, const int32_t *read_off,
                        const int64_t *base_off, const uint8_t *bases, const uint8_t *pw, const uint8_t *flags,
                        const int64_t *seq_off, int32_t *status, int32_t *seq_len, uint8_t *seq, uint8_t *qual, float *raw_qv,
                        float *rq, int32_t *np, float *ec, int32_t *iters, int32_t *n_windows, int nthreads)
{
    return orc_consensus_batch_kin(model, opts, n_zmw, snr, read_off, base_off, bases, pw, flags, seq_off, status, seq_len, seq,
                                   qual, raw_qv, rq, np, ec, iters, n_windows, nthreads, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
}
