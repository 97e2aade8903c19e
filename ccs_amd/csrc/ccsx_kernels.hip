// ccsx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the CCS per-ZMW consensus hot path.
//
//   k_setup       A0    per-ZMW Arrow parameter tables + z-score parameters              (docs/how-does-ccs-work.md:90-94)
//   k_poa_init    D3    pass selection, backbone chain (cascade pass 2: the draft itself)   (docs/how-does-ccs-work.md:34-51,
//   k_poa_dp      D1/D2 banded POA DP of one pass, FOUR graphs per wave64 (16-lane DPP rows)   docs/faq/accuracy-vs-passes.md:41-46)
//   k_poa_thread  D3    gate, traceback, threading of the pass, column records of the next DP
//   k_poa_finish  D3    heaviest path, draft, window bounds (step 4)
//   k_align16     step3 subread -> draft, 16-row band, FOUR passes per wave; band-saturation trigger (SPEC v5)
//   k_align       step3 the 64-row retry of the alignment cascade                            (docs/how-does-ccs-work.md:53-55)
//   k_rescue      step3/6 split / double-split alignment, partial passes                     (docs/how-does-ccs-work.md:74-78)
//   k_post        A7    per-ZMW usable-pass accounting, draft-cascade marks                  (docs/faq/accuracy-vs-passes.md:37-39)
//   k_wmap(_fill) step4 the batch's windows in compact order (grid map of the polish stage)
//   k_polish      A1-A6 Arrow alpha/beta fill, candidate filter, mutation scoring, polish loop, QVs; one workgroup per window
//                                                                                            (docs/how-does-ccs-work.md:57-61,80-106)
//   k_kinetics    N4    HiFi kinetics of the converged windows                               (docs/faq/kinetics.md:8-18)
//   k_stitch      step10 concatenate window cores, rq / np / ec, status                      (docs/how-does-ccs-work.md:108-112)
//
// The arithmetic follows DESIGN.md §SPEC operation by operation (compiled with -ffp-contract=off, no
// fast-math) so that sequences are bit-identical to the CPU restatement.  No MFMA: the recurrences are
// 3-term stencils with data-dependent coefficients.  DP matrices live in LDS (polish) or registers
// (alignment); only the POA keeps score columns in HBM scratch because its DAG is irregular.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

#include "ccsx.h"
#include "ccsx_kernels.h"
#include "wave_ops.h"

// gfx950 counts global STORES in vmcnt like loads, and the counter retires in order: a wait for one load drains every store issued before it.  The compiler
// waits for a load at its first use; when a RARE branch loads a value that is used after the join, that wait sits on the common path and every
// iteration stalls until the stores of the previous iteration have been acknowledged (k_poa_dp: two such waits per column).  LANDED(x) makes x
// "used" right where it is written, so the wait stays inside the rare branch and the common path keeps its stores in flight.
#define LANDED(x) asm volatile("" : "+v"(x))

// ---- experiment switches (timing studies of DESIGN.md §8; every one of them computes WRONG results).  They compile only together with -DCCSX_EXPERIMENT,
// which the library reports through ccsx_build_flags() and which turns ccsx_spec_version() negative, so that such a build cannot pass for the product
// (tests/test_abi.py; tools/gpu_ab.sh adds the define itself).
#if (defined(CCSX_EXP_NO_FILL) || defined(CCSX_EXP_NO_SCORE) || defined(CCSX_EXP_ONE_ROUND) || defined(CCSX_EXP_ALL_VALID) || defined(CCSX_EXP_CHEAP_VALIDITY) || \
     defined(CCSX_EXP_NO_ROWS) || defined(CCSX_EXP_NO_SCORE_LOG) || defined(CCSX_EXP_SKIP_ROUND2_SCORE) || defined(CCSX_EXP_NO_QV_EXP) || defined(CCSX_EXP_REPEAT) || defined(CCSX_EXP_NO_BANDMASK) || \
     defined(CCSX_EXP_HOT_PROLOGUE) || defined(CCSX_EXIT_AFTER_PROLOGUE)) && !defined(CCSX_EXPERIMENT)
#error "CCSX_EXP_* / CCSX_EXIT_AFTER_PROLOGUE switches produce wrong results: build them with -DCCSX_EXPERIMENT (never ship such a library)"
#endif
#define CCSX_STR2(x) #x
#define CCSX_STR(x) CCSX_STR2(x)
int ccsx_kernel_is_experiment()
{
#ifdef CCSX_EXPERIMENT
    return 1;
#else
    return 0;
#endif
}
const char *ccsx_kernel_build_flags()
{
    return ""
#ifdef CCSX_EXPERIMENT
        " CCSX_EXPERIMENT"
#endif
#ifdef CCSX_EXP_NO_FILL
        " CCSX_EXP_NO_FILL"
#endif
#ifdef CCSX_EXP_NO_SCORE
        " CCSX_EXP_NO_SCORE"
#endif
#ifdef CCSX_EXP_ONE_ROUND
        " CCSX_EXP_ONE_ROUND"
#endif
#ifdef CCSX_EXP_ALL_VALID
        " CCSX_EXP_ALL_VALID"
#endif
#ifdef CCSX_EXP_CHEAP_VALIDITY
        " CCSX_EXP_CHEAP_VALIDITY"
#endif
#ifdef CCSX_EXP_NO_ROWS
        " CCSX_EXP_NO_ROWS"
#endif
#ifdef CCSX_EXP_NO_SCORE_LOG
        " CCSX_EXP_NO_SCORE_LOG"
#endif
#ifdef CCSX_EXP_SKIP_ROUND2_SCORE
        " CCSX_EXP_SKIP_ROUND2_SCORE"
#endif
#ifdef CCSX_EXP_NO_QV_EXP
        " CCSX_EXP_NO_QV_EXP"
#endif
#ifdef CCSX_EXP_NO_BANDMASK
        " CCSX_EXP_NO_BANDMASK"
#endif
#ifdef CCSX_EXP_HOT_PROLOGUE
        " CCSX_EXP_HOT_PROLOGUE"
#endif
#ifdef CCSX_EXP_REPEAT
        " CCSX_EXP_REPEAT=" CCSX_STR(CCSX_EXP_REPEAT)
#endif
#ifdef CCSX_EXIT_AFTER_PROLOGUE
        " CCSX_EXIT_AFTER_PROLOGUE"
#endif
#ifdef CCSX_PROFILE_PHASES
        " CCSX_PROFILE_PHASES"
#endif
#ifdef CCSX_DEBUG_CHECKS
        " CCSX_DEBUG_CHECKS"
#endif
#ifdef CCSX_EXTRA_FLAGS_STR                                  // (__graft_entry__.build(): whatever $CCSX_EXTRA_FLAGS held, e.g. tuning overrides of the PW_* defaults)
        " extra: " CCSX_EXTRA_FLAGS_STR
#endif
        ;
}
#define LANES 64
#define PW_MAXREADS_SPEC CCSX_MAX_PASSES      // passes of a ZMW the engine uses (k_polish / k_kinetics take them in groups of PW_MAXREADS)
#ifdef CCSX_PROFILE_PHASES
// (round 6: the phase times of wave 0 are summed in LDS (thread 0) and flushed ONCE per workgroup — the first version added a barrier and a global atomic per phase
// boundary, which made the instrumented kernel four times slower and every phase look alike)
#define PHASE_T0() __shared__ unsigned long long sPh[8]; unsigned long long ph_t = __builtin_readcyclecounter(); if (threadIdx.x < 8) sPh[threadIdx.x] = 0ull; (void)ph_t
#define PHASE(idx) do { unsigned long long n_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) sPh[(idx)] += n_ - ph_t; ph_t = n_; } while (0)
#define PHASE_FLUSH() do { if (threadIdx.x == 0 && (blockIdx.x & 63u) == 0u) for (int q_ = 0; q_ < 8; ++q_) atomicAdd((unsigned long long *)P.phase + q_, sPh[q_]); } while (0)   /* one workgroup in 64: millions of atomics on eight addresses are a workload of their own */
#else
#define PHASE_T0() do { } while (0)
#define PHASE(idx) do { } while (0)
#define PHASE_FLUSH() do { } while (0)
#endif
#ifdef CCSX_PROFILE_PHASES                  // one-wave kernels: lane 0's cycle counter between phases, summed over the graphs
#define TPH_T0() unsigned long long tph_t = __builtin_readcyclecounter(); (void)tph_t
#define TPH(idx) do { if (threadIdx.x == 0) { unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd((unsigned long long *)P.phase + (idx), n_ - tph_t); tph_t = n_; } } while (0)
#else
#define TPH_T0() do { } while (0)
#define TPH(idx) do { } while (0)
#endif
#ifdef CCSX_DEBUG_CHECKS
#ifndef CCSX_CHK_MASK
#define CCSX_CHK_MASK 0xff
#endif
#define CHK(cond, code) do { if (((CCSX_CHK_MASK >> ((code) - 101)) & 1) && !(cond)) { if (atomicCAS(P.debug, 0, (code)) == 0) { P.debug[1] = __LINE__; } return; } } while (0)
#else
#define CHK(cond, code) do { } while (0)
#endif
#define NEGV (-(1 << 28))
#define SC_MATCH 3
#define SC_MISMATCH (-5)
#define SC_INS (-4)
#define SC_DEL (-4)
#define MV_DIAG 0
#define MV_DEL 1
#define MV_INS 2
#define MUT_EPS 0.01f
#define MUT_SEP 5
#define MULTI_ROUNDS 2
#define JMIN_DEL 4
#define AB_TOL 0.01f
#define TINY_P 1e-30f
#define SKIP_MARGIN 6
#define SKIP_SPREAD 3
#define SCORE_BAND 5                  // SPEC: half width (read rows) of the mutation scoring band around the window diagonal
// (SPEC v7 / ABI v6: the smallest per-base error probability that is reported — also for a position the candidate filter skips — is P.perr_floor = 10^(-opts.max_qv/10),
// 1e-5 = Q50 by default: nothing measured supports a higher claim; max_qv 93 = the reference's documented range)
#define REP_NP0 10                    // SPEC v8: a CLOSED tract's floor (both ends inside the visible template) is scaled by (REP_NP0 / passes used)^2 beyond REP_NP0 passes
#define REP_ERRS 0.2f                 // SPEC v7 "repeat-count floor": a core base inside a period-p tandem tract of L >= REP_MINLEN(p) visible bases reports p_err >= REP_ERRS / L
#define FILL_MARGIN 2                 // SPEC v6 "banded fill": alpha / beta exist on the diagonals j - i in [min(0, J - I) - (Wr + 2), max(0, J - I) + (Wr + 2)] only
#define DQ_SCALE 65536.0f
#define DQ_CLAMP 100.0f

// ------------------------------------------------------------------------------------------------
// deterministic log2 / exp2 (DESIGN.md §SPEC "det math"): only +, *, / (IEEE, correctly rounded) and bit ops
__device__ __forceinline__ float det_log2f(float x)
{
    uint32_t u = __float_as_uint(x);
    if ((int32_t)u < 0x00800000) return -127.0f;
    int e = (int)(u >> 23) - 127;
    float f = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (f > 1.41421356f) { f = f * 0.5f; e = e + 1; }
    float t = f - 1.0f;
    float s = __fdiv_rn(t, 2.0f + t);
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

__device__ __forceinline__ float det_exp2f(float x)
{
    if (x < -125.0f) x = -125.0f;
    if (x > 60.0f) x = 60.0f;
    float n = floorf(x + 0.5f);
    float f = (x - n) * 0.693147181f;
    float p = f * 1.98412698e-4f;
    p = p + 1.38888889e-3f;
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
    return p * __uint_as_float((uint32_t)(ni + 127) << 23);
}

__device__ __forceinline__ int ctx_of(int prev, int cur) { if (prev > 3) prev = (cur + 2) & 3; return prev * 4 + cur; }
__device__ __forceinline__ int obs_of(int base, int pw) { int b = pw < 1 ? 1 : (pw > 3 ? 3 : pw); return (base & 3) * 3 + (b - 1); }   // only the low two bits of a base code count

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { int o = __shfl_xor(v, s); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { int o = __shfl_xor(v, s); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { float o = __shfl_xor(v, s); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int bcast_i32(int v, int lane) { return __shfl(v, lane); }

// scalar min / max of wave-uniform values (the compiler otherwise moves such chains to the VALU for v_min3 and reads them back)
__device__ __forceinline__ int smin(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int smax(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
// band placement of a column from scalars (k_align, k_poa's chain step): same arithmetic as band_lo
__device__ __forceinline__ int band_lo_s(int lo_u, int bestrow_u, int hi /* max(I - 63, 0) */)
{
    return smax(smin(smin(smax(bestrow_u + 1 - CCSX_BAND / 2, lo_u), lo_u + 2), hi), 0);
}
__device__ __forceinline__ int band_lo(int lo_u, int bestrow_u, int I)
{
    int lo = bestrow_u + 1 - CCSX_BAND / 2;
    if (lo < lo_u) lo = lo_u;
    if (lo > lo_u + 2) lo = lo_u + 2;
    int hi = I - (CCSX_BAND - 1); if (hi < 0) hi = 0;
    if (lo > hi) lo = hi;
    if (lo < 0) lo = 0;
    return lo;
}

// ------------------------------------------------------------------------------------------------
// A0: per-ZMW parameter tables.  One thread per (zmw, ctx).
__global__ void k_setup(KParams P)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P.n_zmw * CCSX_NCTX) return;
    int z = gid / CCSX_NCTX, k = gid % CCSX_NCTX;
    const ccsx_model *m = P.model;
    int cur = k & 3;
    float s = P.snr[z * 4 + cur];
    if (s < m->snr_lo) s = m->snr_lo;
    if (s > m->snr_hi) s = m->snr_hi;
    float w[3];
#pragma unroll
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
    float pM = __fdiv_rn(1.0f, den), pB = __fdiv_rn(w[0], den), pS = __fdiv_rn(w[1], den), pD = __fdiv_rn(w[2], den);
    float *ME = P.tabME + (size_t)z * 192, *INS = P.tabINS + (size_t)z * 192;
    for (int o = 0; o < CCSX_NOBS; ++o) {
        int b = o / 3, pwb = o % 3;
        ME[k * CCSX_NOBS + o] = (pM * m->em_match[k][o]) * 4.0f;
        if (b == cur) INS[k * CCSX_NOBS + o] = (pB * m->em_branch[k][pwb]) * 4.0f;
        else          INS[k * CCSX_NOBS + o] = ((pS * m->em_stick[k][pwb]) * 0.333333333f) * 4.0f;
    }
    P.tabDL[(size_t)z * 16 + k] = pD;
    // A7 z-score parameters (SPEC "z-score gate"; oracle orc_zparams, same operation order)
    {
        const float pA = pM + pD, pI = pB + pS;
        const float lM = det_log2f(pM), lD = det_log2f(pD), lB = det_log2f(pB), lS = det_log2f(pS * 0.333333333f);
        float e1m = 0.0f, e2m = 0.0f, e1b = 0.0f, e2b = 0.0f, e1s = 0.0f, e2s = 0.0f;
        for (int o = 0; o < CCSX_NOBS; ++o) { const float p = m->em_match[k][o], l = det_log2f(p), t = p * l; e1m = e1m + t; e2m = e2m + t * l; }
        for (int b = 0; b < 3; ++b) { const float p = m->em_branch[k][b], l = det_log2f(p), t = p * l; e1b = e1b + t; e2b = e2b + t * l; }
        for (int b = 0; b < 3; ++b) { const float p = m->em_stick[k][b], l = det_log2f(p), t = p * l; e1s = e1s + t; e2s = e2s + t * l; }
        const float a1 = __fdiv_rn(pM * (lM + e1m) + pD * lD, pA);
        const float a2 = __fdiv_rn(pM * ((lM * lM + (2.0f * lM) * e1m) + e2m) + pD * (lD * lD), pA);
        const float s1 = __fdiv_rn(pB * (lB + e1b) + pS * (lS + e1s), pI);
        const float s2 = __fdiv_rn(pB * ((lB * lB + (2.0f * lB) * e1b) + e2b) + pS * ((lS * lS + (2.0f * lS) * e1s) + e2s), pI);
        const float vA = a2 - a1 * a1, vS = s2 - s1 * s1;
        const float EN = __fdiv_rn(pI, pA), VN = __fdiv_rn(pI, pA * pA);
        P.tabZ[(size_t)z * 32 + k] = EN * s1 + a1;
        P.tabZ[(size_t)z * 32 + 16 + k] = (EN * vS + VN * (s1 * s1)) + vA;
    }
}

// ------------------------------------------------------------------------------------------------
// packed 2-bit oriented read in LDS (one wave owns it)
// mode bit 0: walk the bytes backwards, bit 1: complement.  0 = as stored, 3 = reverse complement (rev), 1 / 2 = those two read
// from their last base to their first (the reversed half of the split alignment)
// 16 consecutive bases of a read as one packed word.  A load under `if (i < L)` is waited for on the spot, so the plain form (still used for the words
// at the two ends of a read) costs sixteen DEPENDENT memory round trips per word (~ 1.5 us each under load: 0.2 ms per 2048-base chunk of k_poa_dp, with
// the other three graphs of the wave waiting at the barrier).  A word that lies inside the read issues its sixteen byte loads off ONE address
// (immediate offsets) before the first is used: one round trip.
// (NB = byte loads in flight at a time: 16, or 8 where the caller's loop is short of registers)
template <int NB>
__device__ __forceinline__ uint32_t pack16_bases(const uint8_t *bases, int L, int i0, bool backwards, bool complement)
{
    uint32_t v = 0;
    if (NB > 0 && i0 >= 0 && i0 + 15 < L) {
#pragma unroll
        for (int h = 0; h < 16; h += (NB > 0 ? NB : 16)) {
            uint32_t b[NB > 0 ? NB : 1];
            if (backwards) {
                const uint8_t *p = bases + (L - 1 - i0 - h);
#pragma unroll
                for (int k = 0; k < NB; ++k) b[k] = (uint32_t)p[-k];
            } else {
                const uint8_t *p = bases + i0 + h;
#pragma unroll
                for (int k = 0; k < NB; ++k) b[k] = (uint32_t)p[k];
            }
            if (NB < 16) asm volatile("" : "+v"(b[NB - 1]));    // (keeps the second batch from being hoisted above the first batch's use)
#pragma unroll
            for (int k = 0; k < NB; ++k) v |= (b[k] & 3u) << (2 * (h + k));
        }
        if (complement) v = ~v;                                 // 3 - b on every 2-bit field
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = i0 + k;
            if (i >= 0 && i < L) { uint32_t b = (uint32_t)bases[backwards ? L - 1 - i : i]; if (complement) b = 3u - b; v |= (b & 3u) << (2 * k); }
        }
    }
    return v;
}
__device__ __forceinline__ void load_read_packed_mode(uint32_t *sread, const uint8_t *bases, int L, int mode, int lane)
{
    int nw = (L + 15) >> 4;
    for (int w = lane; w < nw; w += LANES) sread[w] = pack16_bases<16>(bases, L, w << 4, (mode & 1) != 0, (mode & 2) != 0);
}
__device__ __forceinline__ void load_read_packed(uint32_t *sread, const uint8_t *bases, int L, int rev, int lane)
{
    load_read_packed_mode(sread, bases, L, rev ? 3 : 0, lane);
}
// a CH16-base chunk [c0, c0 + CH16) of the oriented read, packed 2 bits per base into sread[0 .. CH16/16] (k_align16 keeps only the
// part of each pass its band can reach in LDS: 0.5 KB instead of the whole pass, so the kernel's occupancy is not LDS-bound)
#define CH16 2048
__device__ __forceinline__ void load_read_chunk(uint32_t *sread, const uint8_t *bases, int L, int rev, int c0, int lane)
{
    for (int w = lane; w <= CH16 / 16; w += LANES) sread[w] = pack16_bases<0>(bases, L, c0 + (w << 4), rev != 0, rev != 0);   // (plain form: k_align16 needs <= 64 VGPRs for its 8 waves per SIMD, which cover the latency)
}
__device__ __forceinline__ int read_base_packed(const uint32_t *sread, int i) { return (int)((sread[i >> 4] >> (2 * (i & 15))) & 3u); }
// the band's next read base, wave-uniform index: 64 packed words (1024 bases) of the read sit in one VGPR, lane = word, loaded
// at a point from which the band cannot advance past the window before the next load (<= 2 rows per column); a column takes
// its word with v_readlane into an SGPR (scalar shift / mask): no LDS access and no vector arithmetic in the column
struct BaseCursor { uint32_t vw; int w0; };
__device__ __forceinline__ void cursor_load(BaseCursor &c, const uint32_t *sread, int nwords, int t0, int lane)
{
    c.w0 = (t0 < 0 ? 0 : t0) >> 4;
    const int wi = c.w0 + lane;
    c.vw = sread[wi < nwords ? wi : (nwords > 0 ? nwords - 1 : 0)];
}
__device__ __forceinline__ int base_at(const BaseCursor &c, int t)
{
    return (int)(((uint32_t)__builtin_amdgcn_readlane((int)c.vw, ((t >> 4) - c.w0) & 63) >> (2 * (t & 15))) & 3u);
}

// ------------------------------------------------------------------------------------------------
// D2/D3: sparse POA (docs/how-does-ccs-work.md:34-51; SPEC: POA_BAND = 32 rows, a pass is threaded only if it passes the gate).
//
// Round 3: the POA is a pipeline of four kernels per threaded pass instead of one wave per graph doing everything —
//   k_poa_init    one wave per graph: pass selection, the backbone chain, the column records of the first DP
//   k_poa_dp      ONE WAVE = FOUR GRAPHS: the banded DP of one pass over four graphs in lockstep.  Each graph owns a 16-lane DPP
//                 row; a lane holds TWO band rows (2l, 2l+1), so the 32-row band's insertion chain is a lane-local step plus a
//                 4-step row scan, and the column maximum / best row one packed 4-step row scan.  Every column a graph finishes
//                 goes into an LDS ring (the last PRING columns + the START column, guard cells on both sides): in-edges read
//                 their source column from there at any band offset with plain ds_reads — no register shifting, no per-column
//                 branch on the edge type; only sources more than PRING positions back come from HBM (flagged by the prepass).
//                 Per column 32 move bytes + one 16-byte record go to HBM (round 2: 64 move bytes + the 256-byte score column
//                 of every column an in-edge skipped to + the record).
//   k_poa_thread  one wave per graph: the gate, traceback, threading of the pass into the graph (wave-parallel list insertion),
//                 and the prepass for the next DP: column records by topological position {base, in-edge count, positions of
//                 in-edges 0..2, "a far in-edge reads this column back"} — round 6: the threading itself produces them (see the layout note below)
//   k_poa_finish  heaviest path, draft, window bounds
// Round 6: a vertex IS its topological position.  Everything is laid out by position: crec {base | in-edges << 8 | flags | passes << 20, positions of
// in-edges 0..2} and px {positions of in-edges 3..6} (both ping-pong: threading a pass writes the next numbering while it reads the current one), kinfo {lo,
// colmax, bestrow, position of in-edge 0} and the move row mvK[32] of the current DP pass, M[32] for the far-read columns.  Until round 5 the graph lived by
// vertex ID (record, overflow in-edges, rank, passes-through) beside two order arrays, and every pass gathered / scattered all of them by id to produce the
// column records of the next DP: 8.5 MB of 4- and 16-byte gathers per ZMW.  Now a pass is threaded by ONE streaming merge: the records of the vertices that
// are not on the pass's path shift to their new positions, the path's elements write theirs (existing: in-edges remapped, passes + 1, the new edge; new: a
// fresh record), and the result IS the next DP's column records.
#define PB CCSX_POA_BAND              // rows of the POA band
#define PRING 8                       // columns of a graph the DP keeps in LDS (+ the START column in slot PRING)
#define PGS 40                        // words per ring column: 4 guards, 32 rows, 4 guards
#define CREC_NEED (1 << 16)            // column record: a far in-edge reads this column back from HBM
#define CREC_FAR0 (1 << 17)            //                in-edge 0 comes from more than PRING positions back
#define CREC_NREADS_SHIFT 20           //                bits 20..26: passes that go through the vertex (k_poa_finish)
enum { ST_N, ST_NADDED, ST_OK, ST_PAR, ST_KEND, ST_BS, ST_NPOA, ST_BB, ST_NREADS, ST_REV0, ST_LIVE, ST_WORDS = 16 };
struct PoaSlot {
    int32_t *st;                      // [ST_WORDS] per-graph state that travels between the kernels
    int4 *kinfo, *crec0, *crec1, *px0, *px1;
    int32_t *M, *bestK, *bpK, *pathv, *loK;
    uint8_t *mvK, *needK, *onpK;
};
#define POA_MV_BYTES (PB / 2)         // a column's moves: one nibble per band row (0..13 = in-edge slot * 2 + [deletion], 15 = insertion)
#define POA_BYTES_PER_VERTEX (PB * 4 + 16 * 5 + POA_MV_BYTES + 3 * 4 + 2)       // = 238 (the host sizes a slot with it: ccsx_kernels.h CCSX_POA_BYTES_PER_VERTEX)
static_assert(POA_BYTES_PER_VERTEX == CCSX_POA_BYTES_PER_VERTEX, "the host's slot size and the kernels' layout disagree");

__device__ __forceinline__ PoaSlot poa_slot(const KParams &P, int slot)
{
    PoaSlot s;
    const size_t vc = (size_t)P.vcap_max + 64;
    uint8_t *p = P.poa_scratch + (size_t)slot * P.poa_slot_bytes;
    s.st = (int32_t *)p;      p += 256;
    s.M = (int32_t *)p;       p += vc * PB * 4;
    s.kinfo = (int4 *)p;      p += vc * 16;
    s.crec0 = (int4 *)p;      p += vc * 16;     // column records, ping-pong (ST_PAR names the current one)
    s.crec1 = (int4 *)p;      p += vc * 16;
    s.px0 = (int4 *)p;        p += vc * 16;     // positions of in-edges 3..6, ping-pong like the records; an entry is meaningful only where the record counts more than three
    s.px1 = (int4 *)p;        p += vc * 16;
    s.mvK = p;                p += vc * POA_MV_BYTES;
    s.bestK = (int32_t *)p;   p += vc * 4;      // also the run-count / shift array while threading a read
    s.bpK = (int32_t *)p;     p += vc * 4;
    s.loK = (int32_t *)p;     p += vc * 4;      // band start of the column in the current DP pass (the traceback's half of kinfo)
    s.needK = p;              p += (vc + 3) & ~(size_t)3;   // 1 = some in-edge reaches this column from more than PRING positions ahead
    s.onpK = p;               p += (vc + 3) & ~(size_t)3;   // (while threading) 1 = the vertex is on the pass's path
    s.pathv = (int32_t *)p;
    return s;
}

__device__ __forceinline__ int imed3(int x, int lo, int hi) { x = x > lo ? x : lo; return x < hi ? x : hi; }   // clamp (v_med3_i32)
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int int4_get(const int4 &v, int q) { return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w)); }
__device__ __forceinline__ void int4_set(int4 &v, int q, int x) { if (q == 0) v.x = x; else if (q == 1) v.y = x; else if (q == 2) v.z = x; else v.w = x; }

extern __shared__ uint32_t dyn_lds[];
#define TB_BLOCK 64                   // positions cached per traceback block (64 move rows of 32 bytes)

// zref[z]: bits 0-7 = backbone pass of the current draft, plus the state of the draft cascade (SPEC "fallback draft", "last resort")
#define ZREF_RETRY 256                  // k_post: this ZMW's first draft failed or most passes do not map to it
#define ZREF_DONE 512                   // the fallback draft (pass 1) has been made
#define ZREF_RETRY2 1024                // k_post: the fallback draft failed as well (bits 0-5: its backbone)
#define ZREF_DONE2 2048                 // the last-resort draft (pass 2: the backbone pass itself) has been made
#define ZREF_PASSBIT(pass) ((pass) == 1 ? ZREF_DONE : ZREF_DONE2)

// step 4: window bounds of a draft of Ld bases (one wave); returns the number of windows
__device__ __forceinline__ int poa_windows(const KParams &P, int z, int Ld, int lane)
{
    int nw = 0;
    {
        // step 4 windows.  SPEC: a break nb is bad when for some period p in 1..4 the p-mer before it equals the p-mer
        // after it ("avoid breaking windows at simple repeats", docs/how-does-ccs-work.md:58-60); the target cur+22 moves by
        // 0,+1,-1,+2,-2,+3,-3 to the first good position.  Lane l holds draft[cur+14+l]; E_p = ballot(d[i] == d[i+p]).
        const uint8_t *d = P.draft + P.seq_off[z];
        int32_t *b = P.wbounds + P.wb_off[z];
        int cur = 0;
        if (lane == 0) b[0] = 0;
        while (cur < Ld) {
            int nb;
            if (Ld - cur <= CCSX_WIN_CORE + 6) nb = Ld;
            else {
                const int base = cur + CCSX_WIN_CORE - 8;                 // positions base .. base+15 cover every p-mer examined
                const int pos = base + lane;
                const int x = (lane < 16 && pos < Ld) ? (int)d[pos] : 8 + lane;
                unsigned e[4];
#pragma unroll
                for (int p = 1; p <= 4; ++p) e[p - 1] = (unsigned)__ballot(x == __shfl(x, (lane + p) & 63)) & 0xffffu;
                nb = cur + CCSX_WIN_CORE;
                const int offs[7] = {0, 1, -1, 2, -2, 3, -3};
#pragma unroll
                for (int k = 6; k >= 0; --k) {                            // last assignment wins: scan the preference order backwards
                    const int c = 8 + offs[k];                            // bit index of the candidate break
                    bool bad = false;
#pragma unroll
                    for (int p = 1; p <= 4; ++p) bad |= ((e[p - 1] >> (c - p)) & ((1u << p) - 1u)) == ((1u << p) - 1u);
                    if (!bad) nb = cur + CCSX_WIN_CORE + offs[k];
                }
            }
            ++nw;
            if (lane == 0) b[nw] = nb;
            cur = nb;
        }
    }
    return nw;
}

// ---- k_poa_init: which passes, the backbone chain, the first column records.  One wave per graph (slot = block).
__global__ __launch_bounds__(64) void k_poa_init(KParams P, int z0, int pass, int b0)
{
    const int bx = (int)blockIdx.x + b0;                // (b0: the launch's first workgroup — the POA stage runs as two half-batches on two streams)
    const int lane = threadIdx.x;
    uint32_t *sread = dyn_lds;
    PoaSlot g = poa_slot(P, bx);
    if (z0 + bx >= P.n_zmw) return;
    const int z = rfl(P.zmw_perm[z0 + bx]);     // longest ZMWs first
    if (lane < ST_WORDS) g.st[lane] = 0;                // (ST_LIVE = 0: the other kernels skip this graph unless it is set below)
    __threadfence_block();
    const int r0 = rfl(P.read_off[z]);
    int nreads = rfl(P.read_off[z + 1]) - r0;
    {   // SPEC v5: at most CCSX_MAX_PASSES = 255 passes are used (--top-passes 0 = all of them)
        const int top = (P.opts.top_passes <= 0 || P.opts.top_passes > PW_MAXREADS_SPEC) ? PW_MAXREADS_SPEC : P.opts.top_passes;
        if (nreads > top) nreads = top;
    }
    // SPEC "partial passes" (flag bit 1; they follow the ZMW's full-length passes): not in the draft, not counted as passes
    const int nall = nreads;
    {
        int nf = 0;
        for (int b0 = 0; b0 < nall; b0 += LANES) nf += __popcll(__ballot(b0 + lane < nall && !(P.flags[r0 + (b0 + lane < nall ? b0 + lane : 0)] & 2)));
        nreads = rfl(nf);
    }
    // SPEC "fallback draft" (docs/faq/accuracy-vs-passes.md:41-46: a cascade from fast to robust draft generators): pass 1
    // (only for ZMWs k_post marked) takes the pass whose length is closest to the median as backbone and threads twice as
    // many passes, starting at the backbone and wrapping around
    // SPEC "last resort" (pass 2, only for ZMWs whose fallback draft failed too): the pass closest to the median among the passes
    // that have not been a backbone yet is the draft itself, no POA; the polish repairs its errors
    int bb = 0;
    if (pass >= 1) {
        const int zr = rfl(P.zref[z]);
        if (!(zr & (pass == 1 ? ZREF_RETRY : ZREF_RETRY2))) return;
        const int bb1 = pass == 2 ? (zr & 255) : -1;
        // up to CCSX_MAX_PASSES = 255 full-length passes: their lengths go to LDS, a lane ranks the passes lane, lane + 64, ...
        int32_t *slen = (int32_t *)sread;
        for (int q = lane; q < nreads; q += LANES) slen[q] = (int)(P.base_off[r0 + q + 1] - P.base_off[r0 + q]);
        __syncthreads();
        int medl = 0x7fffffff;                          // the length whose rank (ties by index) is nreads / 2
        for (int i = lane; i < nreads; i += LANES) {
            const int len = slen[i];
            int rank = 0;
            for (int q = 0; q < nreads; ++q) { const int lq = slen[q]; rank += (lq < len || (lq == len && q < i)) ? 1 : 0; }
            if (rank == nreads / 2) medl = len;
        }
        const int med = rfl(wave_min_i32(medl));
        int key = 0x7fffffff;
        for (int i = lane; i < nreads; i += LANES) {
            int dist = slen[i] - med; dist = dist < 0 ? -dist : dist;
            const bool cand = !(pass == 2 && ((i == bb1 && nreads > 1) || (i == 0 && nreads > 2)));   // not a backbone that failed
            const int k = ((dist > 0x3fffff ? 0x3fffff : dist) << 8) | i;
            if (cand && k < key) key = k;
        }
        bb = rfl(wave_min_i32(key)) & 255;
        __syncthreads();                                // (sread is loaded with the backbone below)
    }
    if (lane == 0) { P.nreads_used[z] = nall; P.nfull[z] = nreads; P.draft_len[z] = 0; P.nwin[z] = 0; P.zref[z] = bb | (pass ? ZREF_PASSBIT(pass) : 0); }
    // a new draft invalidates the partial passes' alignments of the previous generator (k_align16 resets the full-length passes it
    // realigns; the partial ones are aligned by k_rescue, which only looks at passes that are not valid)
    for (int q = nreads + lane; q < nall; q += LANES) { P.avalid[r0 + q] = 0; P.ascore[r0 + q] = NEGV; }
    const bool enough = !(nreads < P.opts.min_passes || nreads < 1);
    if (!enough) { if (lane == 0) P.zstat[z] = CCSX_TOO_FEW_PASSES; return; }
    const int cov = pass ? 2 * P.opts.max_poa_cov : P.opts.max_poa_cov;
    const int npoa = nreads < cov ? nreads : cov;
    const int vcap = rfl(P.vcap[z]);
    const int rev0 = rfl(P.flags[r0 + bb] & 1);
    const int r = r0 + bb;
    const uint8_t *rb = P.bases + P.base_off[r];
    const int I = rfl((int)(P.base_off[r + 1] - P.base_off[r]));
    if (pass == 2) {                                    // the backbone pass is the draft
        int Ld = I <= P.dcap[z] ? I : 0, nw = 0, stat = -1;
        uint8_t *draft = P.draft + P.seq_off[z];
        for (int q = lane; q < Ld; q += LANES) draft[q] = rb[q] & 3;
        __threadfence_block();
        if (Ld <= 0) stat = CCSX_DRAFT_FAILURE;
        else if (Ld < P.opts.min_length) stat = CCSX_TOO_SHORT;
        else if (Ld > P.opts.max_length) stat = CCSX_TOO_LONG;
        else nw = poa_windows(P, z, Ld, lane);
        if (lane == 0) { P.draft_len[z] = Ld; P.nwin[z] = (stat < 0) ? nw : 0; P.zstat[z] = (stat < 0) ? CCSX_SUCCESS : stat; }
        return;
    }
    load_read_packed(sread, rb, I, 0, lane);
    __syncthreads();
    int ok = 1, n = 0;
    if (I > vcap) ok = 0;
    else {                                              // first read: backbone chain; its column records are trivial
        for (int i = lane; i < I; i += LANES) {
            const int b = read_base_packed(sread, i);
            g.crec0[i] = make_int4(b | ((i > 0 ? 1 : 0) << 8) | (1 << CREC_NREADS_SHIFT), i - 1, -1, -1);
            g.needK[i] = 0;                               // (a chain has no far in-edge)
        }
        n = I;
    }
    __threadfence_block();
    if (lane == 0) {
        g.st[ST_N] = n; g.st[ST_NADDED] = 1; g.st[ST_OK] = ok; g.st[ST_PAR] = 0; g.st[ST_KEND] = -1; g.st[ST_BS] = NEGV;
        g.st[ST_NPOA] = npoa; g.st[ST_BB] = bb; g.st[ST_NREADS] = nreads; g.st[ST_REV0] = rev0; g.st[ST_LIVE] = 1;
    }
}

// ---- k_draft_in: the polish seam (ccsx_polish_batch, docs/img/ccs-impl.png "Polish Stage" fed by a host-side draft generator; docs/faq/revio.md:35-53: Arrow
// run again for QVs on a sequence made elsewhere).  The caller's drafts are in P.draft already; one wave per ZMW sets what the draft generators leave behind:
// passes used / full-length passes, the orientation reference, status by length, window bounds.  No cascade follows: the alignment's outcome is final.
__global__ __launch_bounds__(64) void k_draft_in(KParams P)
{
    const int lane = threadIdx.x, z = blockIdx.x;
    if (z >= P.n_zmw) return;
    const int r0 = rfl(P.read_off[z]);
    int nreads = rfl(P.read_off[z + 1]) - r0;
    {
        const int top = (P.opts.top_passes <= 0 || P.opts.top_passes > PW_MAXREADS_SPEC) ? PW_MAXREADS_SPEC : P.opts.top_passes;
        if (nreads > top) nreads = top;
    }
    const int nall = nreads;
    {
        int nf = 0;
        for (int b0 = 0; b0 < nall; b0 += LANES) nf += __popcll(__ballot(b0 + lane < nall && !(P.flags[r0 + (b0 + lane < nall ? b0 + lane : 0)] & 2)));
        nreads = rfl(nf);
    }
    int bb = rfl(P.din_bb[z]);
    if (bb < 0 || bb >= (nall > 0 ? nall : 1)) bb = 0;
    int Ld = rfl(P.din_len[z]);
    if (Ld < 0 || Ld > P.dcap[z]) Ld = 0;                  // (a draft that does not fit its slot is no draft)
    int stat = -1, nw = 0;
    if (nreads < P.opts.min_passes || nreads < 1) stat = CCSX_TOO_FEW_PASSES;
    else if (Ld <= 0) stat = CCSX_DRAFT_FAILURE;
    else if (Ld < P.opts.min_length) stat = CCSX_TOO_SHORT;
    else if (Ld > P.opts.max_length) stat = CCSX_TOO_LONG;
    uint8_t *draft = P.draft + P.seq_off[z];
    for (int q = lane; q < Ld; q += LANES) draft[q] = draft[q] & 3;     // only the low two bits of a base code count, as for the subreads
    __threadfence_block();
    if (stat < 0) nw = poa_windows(P, z, Ld, lane);
    for (int q = lane; q < nall; q += LANES) { P.avalid[r0 + q] = 0; P.ascore[r0 + q] = NEGV; }
    if (lane == 0) {
        P.nreads_used[z] = nall; P.nfull[z] = nreads; P.zref[z] = bb;
        P.draft_len[z] = (stat == CCSX_TOO_FEW_PASSES) ? 0 : Ld; P.nwin[z] = (stat < 0) ? nw : 0; P.zstat[z] = (stat < 0) ? CCSX_SUCCESS : stat;
    }
}

// a CH16-base chunk of the oriented read for one 16-lane group, packed 2 bits per base, shifted by ONE base: word w holds the bases
// c0 - 1 + 16 w .. (index -1 = a dummy), so that the base of row i - 1 for row i = c0 + x sits at packed position x
__device__ __forceinline__ void load_read_chunk_m1(uint32_t *sread, const uint8_t *bases, int L, int rev, int c0, int l16)
{
    for (int w = l16; w <= CH16 / 16 + 1; w += 16) sread[w] = pack16_bases<8>(bases, L, c0 - 1 + (w << 4), rev != 0, rev != 0);
}

// ---- k_poa_dp: the banded DP of pass rr over FOUR graphs per wave (see the header of this section)
// (branch layout: a taken branch costs the wave ~ 20 cycles of instruction fetch and this kernel runs two waves per SIMD, so the conditions of the rare blocks
// below carry __builtin_expect(., 0) — the blocks move out of line and the common column falls through.  The "two or more in-edges" blocks (43 % of the
// wave's columns) carry none: either hint measured slower, profiles/r04_poa_dp_vmcnt.txt)
__global__ __launch_bounds__(64) void k_poa_dp(KParams P, int z0, int pass, int rr, int b0)
{
    const int bx = (int)blockIdx.x + b0;                // (b0: the launch's first workgroup — the POA stage runs as two half-batches on two streams)
    __shared__ __attribute__((aligned(16))) int32_t sRing[4][PRING + 1][PGS];
    __shared__ __attribute__((aligned(16))) int4 sKin[4][PRING + 1];
    __shared__ __attribute__((aligned(16))) int4 sCrec[4][16];
    __shared__ uint32_t sRead[4][CH16 / 16 + 2];
    const int lane = threadIdx.x, gq = lane >> 4, l = lane & 15;
    const int zi = z0 + 4 * bx + gq;
    const bool have = zi < P.n_zmw && 4 * bx + gq < P.poa_slots;
    const PoaSlot G = poa_slot(P, have ? 4 * bx + gq : 4 * bx);   // per lane, uniform inside a 16-lane group
    int32_t *st = G.st;
    int32_t *Mcol = G.M;
    int4 *kinfo = G.kinfo;
    const int par_ = have ? st[ST_PAR] : 0;
    const int4 *crec = par_ ? G.crec1 : G.crec0;        // the current numbering's column records / overflow in-edges (ST_PAR)
    const int32_t *pxw = (const int32_t *)(par_ ? G.px1 : G.px0);
    const uint8_t *needK = G.needK;
    uint8_t *mvK = G.mvK;
    bool live = have && st[ST_LIVE] && st[ST_OK] && rr < st[ST_NPOA];
    if (!__any(live)) return;
    const int z = have ? P.zmw_perm[zi] : 0;
    const int r0 = P.read_off[z];
    const int n0 = live ? st[ST_N] : 0;
    int I = 0, rev = 0;
    const uint8_t *rb = P.bases;
    if (live) {
        const int bb = st[ST_BB], nr = st[ST_NREADS];
        const int r = r0 + (bb + rr < nr ? bb + rr : bb + rr - nr);
        rb = P.bases + P.base_off[r];
        I = (int)(P.base_off[r + 1] - P.base_off[r]);
        rev = ((P.flags[r] & 1) != st[ST_REV0]) ? 1 : 0;
    }
    // ---- LDS: guards and unused cells = NEGV, the START column M[l] = l * INS (l <= I), its record (lo 0, colmax 0, best row 0)
    for (int e = l; e < (PRING + 1) * PGS; e += 16) (&sRing[gq][0][0])[e] = NEGV;
    __syncthreads();
    sRing[gq][PRING][4 + 2 * l] = (2 * l <= I) ? 2 * l * SC_INS : NEGV;
    sRing[gq][PRING][5 + 2 * l] = (2 * l + 1 <= I) ? (2 * l + 1) * SC_INS : NEGV;
    if (l == 0) sKin[gq][PRING] = make_int4(0, 0, 0, 0);
    int c0 = 0;                                         // first base of the group's read chunk in LDS
    load_read_chunk_m1(sRead[gq], rb, I, rev, 0, l);
    __syncthreads();
    const int hiI = I - (PB - 1) > 0 ? I - (PB - 1) : 0;
    const int nmax = wave_max_i32(n0);
    int bs = NEGV, kend = -1;
    // per-lane constants of the column step
    const int l2 = 2 * l, l8 = 8 * l;
    const int kRow0 = 63 - 2 * l, kRow1 = 62 - 2 * l;
    typedef int32_t __attribute__((address_space(3))) *lds_i32;
    const uint32_t ringBase = (uint32_t)(uintptr_t)(lds_i32)&sRing[gq][0][0];      // 32-bit LDS addresses
    const uint32_t readBase = (uint32_t)(uintptr_t)(lds_i32)(int32_t *)&sRead[gq][0];
    const int bcastAddr = (lane | 15) << 2;                                        // ds_bpermute: the row's last lane
    // running per-lane output pointers (one 64-bit add per column instead of an index multiply)
    uint8_t *mvp = mvK + l;                             // one byte per lane and column: the nibbles of rows 2l and 2l+1
    int4 *kip = kinfo;
    int32_t *lop = G.loK;
    int32_t *Mp = Mcol + 2 * l;
    int4 recN = make_int4(0, -1, -1, -1);               // the NEXT block of 16 column records (lane l: column kb + 16 + l) and its
    int needN = 0;                                      // "a far in-edge reads this column" byte: merged at the hand-off, 16 columns after the loads were issued
    recN = crec[l < n0 ? l : 0];                        // (unconditional, clamped, masked at the hand-off: a load under a lane mask is copied into the
    needN = needK[l < n0 ? l : 0];                      //  loop-carried register at once, i.e. waited for)
#define LDS_I32(addr) (*(lds_i32)(uintptr_t)(addr))
    // wave masks of per-lane conditions straight from the compare (a bool that goes through ballot costs two extra VALU ops)
#define M_NE0(a) __builtin_amdgcn_uicmp((unsigned)(a), 0u, 33)
#define M_SLT(a, b) __builtin_amdgcn_sicmp((int)(a), (int)(b), 40)
#define M_SGT(a, b) __builtin_amdgcn_sicmp((int)(a), (int)(b), 38)
#define M_UGT(a, b) __builtin_amdgcn_uicmp((unsigned)(a), (unsigned)(b), 34)
    const uint32_t kinBase = (uint32_t)(uintptr_t)(lds_i32)(int32_t *)&sKin[gq][0];
    const unsigned long long livem = M_NE0(live ? 1 : 0);
    for (int k = 0; k < nmax; ++k, mvp += POA_MV_BYTES, ++kip, ++lop, Mp += PB) {
        if (__builtin_expect((k & 15) == 0, 0)) {       // hand the prefetched block to LDS, start fetching the one after it
            { int4 r = recN; if (needN) r.x |= CREC_NEED; sCrec[gq][l] = (k + l < n0) ? r : make_int4(0, -1, -1, -1); }
            { const int kn = k + 16 + l < n0 ? k + 16 + l : 0; recN = crec[kn]; needN = needK[kn]; }
            __syncthreads();
        }
        const unsigned long long actm = livem & M_SLT(k, n0);
        const int4 rec = sCrec[gq][k & 15];
        const int vb = rec.x & 3;
        const int p0 = rec.y;
        const unsigned long long multim = actm & M_NE0(rec.x & 0xe00);         // two or more in-edges
        const unsigned long long far0m = actm & M_NE0(rec.x & CREC_FAR0);      // in-edge 0 comes from more than PRING positions back
        // ---- band placement: from the in-edge column with the largest column maximum (first on ties); START for a source
        const uint32_t slot0 = p0 < 0 ? PRING : (uint32_t)(p0 & (PRING - 1));
        int k0x, k0y, k0z;
        { const uint32_t ka = kinBase + slot0 * 16; k0x = LDS_I32(ka); k0y = LDS_I32(ka + 4); k0z = LDS_I32(ka + 8); }
        if (__builtin_expect(far0m != 0ull, 0)) { if (rec.x & CREC_FAR0) { const int4 t = kinfo[p0]; k0x = t.x; k0y = t.y; k0z = t.z; } LANDED(k0x); LANDED(k0y); LANDED(k0z); }
        int ulo = k0x, ubr = k0z;
        int plo1 = 0, plo2 = 0;
        if (multim) {
            const int np = (rec.x >> 8) & 15;
            const bool act = live && k < n0;
            int bestcm = k0y;
            for (int q = 1; q < CCSX_MAXPRED; ++q) {
                if ((actm & M_SGT(np, q)) == 0ull) break;
                int pq = q == 1 ? rec.z : rec.w;
                if (q >= 3) {                           // in-edges 3..6 (rare): the overflow record of the column
                    if (act && np > q) pq = pxw[(size_t)k * 4 + (q - 3)];
                    LANDED(pq);
                }
                const bool on = act && np > q;
                const bool farq = on && k - pq > PRING;
                const uint32_t ka = kinBase + (on ? (uint32_t)(pq & (PRING - 1)) : (uint32_t)PRING) * 16;
                int qx = LDS_I32(ka), qy = LDS_I32(ka + 4), qz = LDS_I32(ka + 8);
                if (__builtin_expect(M_NE0(farq ? 1 : 0) != 0ull, 0)) { if (farq) { const int4 t = kinfo[pq]; qx = t.x; qy = t.y; qz = t.z; } LANDED(qx); LANDED(qy); LANDED(qz); }
                if (q == 1) plo1 = qx; else if (q == 2) plo2 = qx;
                if (on && qy > bestcm) { bestcm = qy; ulo = qx; ubr = qz; }
            }
        }
        int lo = imed3(ubr + (1 - PB / 2), ulo, ulo + 2);                        // clamp(best row + 1 - 16, ulo, ulo + 2)
        lo = lo < hiI ? lo : hiI;
        lo = lo > 0 ? lo : 0;
        // ---- the read bases of rows r0 - 1 and r0 (r0 = lo + 2l): the chunk follows the band
        if (__builtin_expect((actm & M_UGT(lo - c0, CH16 - PB - 2)) != 0ull, 0)) {
            __syncthreads();
            if (live && k < n0 && (unsigned)(lo - c0) > (unsigned)(CH16 - PB - 2)) {
                c0 = lo - 128 > 0 ? lo - 128 : 0;
                load_read_chunk_m1(sRead[gq], rb, I, rev, c0, l);
            }
            __syncthreads();
        }
        const int r0w = lo + l2;
        int s0, s1;
        {
            const uint32_t idx = (uint32_t)(r0w - c0) & (CH16 - 1);              // (inactive groups: any in-range index)
            const uint32_t wa = readBase + ((idx >> 4) << 2);
            const uint32_t wlo = (uint32_t)LDS_I32(wa), whi = (uint32_t)LDS_I32(wa + 4);
            const uint32_t t = __builtin_amdgcn_alignbit(whi, wlo, (idx & 15u) << 1);
            s0 = (vb == (int)(t & 3u)) ? SC_MATCH : SC_MISMATCH;
            s1 = (vb == (int)((t >> 2) & 3u)) ? SC_MATCH : SC_MISMATCH;
        }
        // ---- in-edge 0 (or START): diagonal then deletion; three consecutive rows of the source column at the band offset
        int b0, b1, m0, m1;
        {
            const int w = imed3(r0w - k0x + 3, 1, PGS - 3);   // word of row (r0 - plo) - 1 behind the 4 guard words
            const uint32_t src = ringBase + slot0 * (PGS * 4) + ((uint32_t)w << 2);
            int x0 = LDS_I32(src), y0 = LDS_I32(src + 4), y1 = LDS_I32(src + 8);
            if (__builtin_expect(far0m != 0ull, 0)) {
                if (rec.x & CREC_FAR0) {                // the source column comes from HBM
                    const int32_t *Mu = Mcol + (size_t)p0 * PB;
                    const int o = r0w - k0x - 1;
                    x0 = (unsigned)o < (unsigned)PB ? Mu[o] : NEGV;
                    y0 = (unsigned)(o + 1) < (unsigned)PB ? Mu[o + 1] : NEGV;
                    y1 = (unsigned)(o + 2) < (unsigned)PB ? Mu[o + 2] : NEGV;
                }
                LANDED(x0); LANDED(y0); LANDED(y1);
            }
            const int d0 = x0 + s0, e0 = y0 + SC_DEL, d1 = y0 + s1, e1 = y1 + SC_DEL;
            const bool t0 = e0 > d0, t1 = e1 > d1;              // move codes (a nibble): in-edge slot * 2 + [deletion], 15 = insertion
            b0 = t0 ? e0 : d0; m0 = t0 ? 1 : 0;
            b1 = t1 ? e1 : d1; m1 = t1 ? 1 : 0;
        }
        if (multim) {                                    // further in-edges, in list order: a later candidate wins only if strictly greater
            const int np = (rec.x >> 8) & 15;
            const bool act = live && k < n0;
            for (int q = 1; q < CCSX_MAXPRED; ++q) {
                if ((actm & M_SGT(np, q)) == 0ull) break;
                const bool on = act && np > q;
                int pq = q == 1 ? rec.z : rec.w, plo = q == 1 ? plo1 : plo2;
                if (q >= 3) {
                    if (on) {
                        pq = pxw[(size_t)k * 4 + (q - 3)];
                        if (k - pq > PRING) plo = kinfo[pq].x; else plo = LDS_I32(kinBase + (uint32_t)(pq & (PRING - 1)) * 16);
                    }
                    LANDED(pq); LANDED(plo);
                }
                const bool farq = on && k - pq > PRING;
                const int off = r0w - plo;
                const int w = imed3(off + 3, 1, PGS - 3);
                const uint32_t src = ringBase + (on ? (uint32_t)(pq & (PRING - 1)) : (uint32_t)PRING) * (PGS * 4) + ((uint32_t)w << 2);
                int x0 = LDS_I32(src), y0 = LDS_I32(src + 4), y1 = LDS_I32(src + 8);
                if (__builtin_expect(M_NE0(farq ? 1 : 0) != 0ull, 0)) {
                    if (farq) {
                        const int32_t *Mu = Mcol + (size_t)pq * PB;
                        const int o = off - 1;
                        x0 = (unsigned)o < (unsigned)PB ? Mu[o] : NEGV;
                        y0 = (unsigned)(o + 1) < (unsigned)PB ? Mu[o + 1] : NEGV;
                        y1 = (unsigned)(o + 2) < (unsigned)PB ? Mu[o + 2] : NEGV;
                    }
                    LANDED(x0); LANDED(y0); LANDED(y1);
                }
                const int mq = q << 1;
                int c;
                c = x0 + s0;     if (on && c > b0) { b0 = c; m0 = mq; }
                c = y0 + SC_DEL; if (on && c > b0) { b0 = c; m0 = mq | 1; }
                c = y0 + s1;     if (on && c > b1) { b1 = c; m1 = mq; }
                c = y1 + SC_DEL; if (on && c > b1) { b1 = c; m1 = mq | 1; }
            }
        }
        // ---- insertion chain x_i = max(c_i, x_{i-1} + INS) over the 32 rows: lane-local (row 2l -> 2l+1), an exclusive 4-step
        // row scan of the lanes' outgoing values, then the incoming value applied to both rows
        { const int c = b0 + SC_INS; if (c > b1) { b1 = c; m1 = 15; } }
        {
            const int S = row_scan_max_i32(b1 + l8);
            const int Sp = __builtin_amdgcn_update_dpp(-(1 << 30), S, DPP_ROW_SHR(1), 0xf, 0xf, false);
            const int c0v = Sp - (l8 - 8 - SC_INS);     // (final value of row 2l - 1) + INS ; lane 0: very negative
            const int c1v = Sp - (l8 - 8 - 2 * SC_INS);
            if (c0v > b0) { b0 = c0v; m0 = 15; }
            if (c1v > b1) { b1 = c1v; m1 = 15; }
        }
        if (r0w > I || b0 < NEGV / 2) b0 = NEGV;
        if (r0w >= I || b1 < NEGV / 2) b1 = NEGV;
        // ---- column maximum and the first row that attains it: one packed row scan (value << 6 | 63 - row).  A column without a
        // valid cell reports -2^24 as its maximum (it loses every comparison a valid column takes part in, like the SPEC's NEG)
        int cm, br;
        {
            const int f = -(1 << 24);
            const int k0_ = ((b0 > f ? b0 : f) * 64) | kRow0, k1_ = ((b1 > f ? b1 : f) * 64) | kRow1;
            const int key = row_allmax_i32(k0_ > k1_ ? k0_ : k1_);       // (an all-reduce by row rotations: no ds_bpermute broadcast on the chain that places the next band)
            cm = key >> 6; br = (lo + 63) - (key & 63);
        }
        // ---- the read's last row: best end cell over all columns (first in topological order)
        if (__builtin_expect((actm & M_SGT(lo + PB, I)) != 0ull, 0)) {
            const int e = r0w == I ? b0 : (r0w + 1 == I ? b1 : NEGV);
            const int ev = __builtin_amdgcn_ds_bpermute(bcastAddr, row_scan_max_i32(e));
            if (live && k < n0 && ev > NEGV / 2 && ev > bs) { bs = ev; kend = k; }
        }
        // ---- out: ring (LDS), moves + column record (HBM), the score column if a far in-edge will read it
        if (live && k < n0) {
            *(int2 *)&sRing[gq][k & (PRING - 1)][4 + l2] = make_int2(b0, b1);
            *mvp = (uint8_t)(m0 | (m1 << 4));
            if (l == 0) {
                sKin[gq][k & (PRING - 1)] = make_int4(lo, cm, br, 0);
                *lop = lo;                                  // all the traceback needs of a column (in-edge 0's position is in its record)
            }
            if (__builtin_expect((rec.x & CREC_NEED) != 0, 0)) {   // a far in-edge will read this column back: its scores and its record
                *(int2 *)Mp = make_int2(b0, b1);
                if (l == 0) *kip = make_int4(lo, cm, br, p0);
            }
        }
    }
#undef M_NE0
#undef M_SLT
#undef M_SGT
#undef M_UGT
#undef LDS_I32
    __threadfence_block();
    if (live && l == 0) { st[ST_KEND] = kend; st[ST_BS] = bs; }
}

// ---- k_poa_thread: gate, traceback and threading of pass rr, then the column records of the next DP.  One wave per graph.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_poa_thread(KParams P, int z0, int pass, int rr, int b0)
{
    const int bx = (int)blockIdx.x + b0;                // (b0: the launch's first workgroup — the POA stage runs as two half-batches on two streams)
    __shared__ __attribute__((aligned(16))) uint8_t sMv[TB_BLOCK * POA_MV_BYTES];   // move rows (a nibble per cell) of the traceback's current block
    const int lane = threadIdx.x;
    uint32_t *sread = dyn_lds;
    PoaSlot g = poa_slot(P, bx);
    if (z0 + bx >= P.n_zmw) return;
    if (!g.st[ST_LIVE] || !g.st[ST_OK] || rr >= g.st[ST_NPOA]) return;
    TPH_T0();
    const int z = rfl(P.zmw_perm[z0 + bx]);
    const int r0 = rfl(P.read_off[z]);
    const int bb = rfl(g.st[ST_BB]), nreads = rfl(g.st[ST_NREADS]), npoa = rfl(g.st[ST_NPOA]);
    const int r = r0 + (bb + rr < nreads ? bb + rr : bb + rr - nreads);
    const uint8_t *rb = P.bases + P.base_off[r];
    const int I = rfl((int)(P.base_off[r + 1] - P.base_off[r]));
    const int kend = rfl(g.st[ST_KEND]), bsc = rfl(g.st[ST_BS]);
    // SPEC "POA gate": a pass is threaded only if its alignment reaches the read's last row with a score of at least 1.0 per base
    if (kend < 0 || bsc < I) return;
    const int rev = rfl(((P.flags[r] & 1) != g.st[ST_REV0]) ? 1 : 0);
    const int vcap = rfl(P.vcap[z]);
    const int n0 = rfl(g.st[ST_N]);
    const int par = rfl(g.st[ST_PAR]);
    const int4 *crec = par ? g.crec1 : g.crec0, *pxc = par ? g.px1 : g.px0;      // the current numbering ...
    int4 *crec_nx = par ? g.crec0 : g.crec1, *px_nx = par ? g.px0 : g.px1;       // ... and the one this pass writes
    load_read_packed(sread, rb, I, rev, lane);
    __syncthreads();
    TPH(8);
    // ---- traceback: lane 0 walks, the block of TB_BLOCK positions it is in is cached in LDS.  Round 3: the per-position words come
    // from the DP's column record (base, in-edge count and the positions of in-edges 0..2 by position: no vertex-record gather), and
    // the NEXT block (the walk goes down the positions) is fetched into registers while the current one is walked.  Round 6: the path
    // holds POSITIONS (of the current numbering; -1 = a new vertex)
    {
        int k = kend, i = I;
        int4 kiN = make_int4(0, 0, 0, -1), crN = make_int4(0, -1, -1, -1);
        int kbN = -1;
        uint4 mvN0 = make_uint4(0, 0, 0, 0);
        auto fetch = [&](int kb_) {
            const int kk = kb_ + lane;
            kbN = kb_;
            kiN = make_int4(0, 0, 0, -1); crN = make_int4(0, -1, -1, -1);
            if (kb_ >= 0) {
                if (kk < n0) { crN = crec[kk]; kiN = make_int4(g.loK[kk], 0, 0, crN.y); }
                mvN0 = ((const uint4 *)(g.mvK + (size_t)kb_ * POA_MV_BYTES))[lane];     // 64 columns x 16 bytes
            }
        };
        fetch((k / TB_BLOCK) * TB_BLOCK);
        while (k >= 0) {
            const int kb = (k / TB_BLOCK) * TB_BLOCK;
            if (kbN != kb) fetch(kb);                          // (an in-edge that skipped a whole block: rare)
            __syncthreads();
            // block cache: the TB_BLOCK move rows go to LDS, the per-position words (band start, position of in-edge 0,
            // vertex id, record words) stay in lane registers and are handed out with v_readlane
            const int4 kiL = kiN, crL = crN;
            const int metaL = crN.x;
            ((uint4 *)sMv)[lane] = mvN0;
            fetch(kb - TB_BLOCK);                              // in flight while this block is walked
            __syncthreads();
            while (k >= kb) {                                  // uniform walk: every lane follows the same (k, i)
                int kl = k - kb;
                int mknown;                                    // the move of the cell the walk stands on after the run, if a lane has read it already (255: no)
                {
                    // a run of plain DIAG steps along the chain (in-edge 0 is the previous position): lane s tests
                    // step s of the run, one ballot gives its length, the path entries are stored by the lanes
                    const int kls = kl - lane;
                    const int src = kls & 63;
                    const int lo_s = __shfl(kiL.x, src), pp_s = __shfl(kiL.w, src);
                    const int off = i - lane - lo_s;
                    const bool inb = kls >= 0 && i - lane >= 1 && (unsigned)off < (unsigned)PB;
                    const int m_s = inb ? ((sMv[kls * POA_MV_BYTES + (off >> 1)] >> ((off & 1) << 2)) & 15) : 255;
                    const unsigned long long simple = __ballot(m_s == 0 && pp_s == k - lane - 1);
                    const int R = (simple == ~0ull) ? 64 : __ffsll((long long)~simple) - 1;
                    if (R > 0) {
                        const int meta_s = __shfl(metaL, src);
                        if (lane < R) {
                            const int ir = i - lane - 1;
                            g.pathv[ir] = ((meta_s & 3) == read_base_packed(sread, ir)) ? k - lane : -1;
                        }
                        k -= R; i -= R;
                        if (R >= 64 || k < kb) continue;
                        kl = k - kb;
                    }
                    // the cell that ended the run (or the very first one, R = 0) is lane R's cell: its move is in that lane's register already, so the step
                    // below needs neither another detection pass (two ds_bpermute + a byte read + a ballot per run before) nor a second LDS read
                    mknown = rl(m_s, R);
                }
                const int lo_k = rl(kiL.x, kl);
                CHK(i - lo_k >= 0 && i - lo_k < PB && i >= 0, 101);
                const int mo = i - lo_k;
                const int m = mknown != 255 ? mknown : rfl((sMv[kl * POA_MV_BYTES + (mo >> 1)] >> ((mo & 1) << 2)) & 15);
                const int t = m == 15 ? MV_INS : (m & 1), slot = m >> 1;   // (MV_DIAG = 0, MV_DEL = 1)
                CHK(i >= 1 || t == MV_DEL, 102);
                if (t == MV_INS) { if (lane == 0) g.pathv[i - 1] = -1; --i; continue; }
                const int meta = rl(metaL, kl);
                const int np = (meta >> 8) & 15;
                int up;
                if (np == 0) up = -1;
                else if (slot == 0) up = rl(kiL.w, kl);
                else if (slot == 1) up = rl(crL.z, kl);
                else if (slot == 2) up = rl(crL.w, kl);
                else up = rfl(((const int32_t *)pxc)[(size_t)k * 4 + (slot - 3)]);
                if (t == MV_DIAG) {
                    if (lane == 0) g.pathv[i - 1] = ((meta & 3) == read_base_packed(sread, i - 1)) ? k : -1;
                    --i;
                }
                CHK(up < k && up >= -1, 103);
                k = up;
            }
            k = rfl(k); i = rfl(i);
        }
        for (int q = lane; q < i; q += LANES) g.pathv[q] = -1;     // leading insertions at START
    }
    __threadfence_block();
    TPH(9);
    // ---- thread the read into the graph (wave-parallel; identical result to the serial list insertion).  Round 6: ONE streaming merge by position.
    //   A  runs of new vertices: the last element of a run records the run's length at its anchor (the path vertex before it; list head if none), path
    //      vertices are flagged;                                   B  prefix sum: S[q] = how far position q moves to the right;
    //   C  the vertices that are NOT on the path shift (in-edges remapped p -> p + S[p], far flags recomputed);
    //   D  the path's elements write their records at their new positions — an existing vertex: its record remapped, passes + 1, the edge from the path
    //      vertex before it (unless present / the in-edge cap is hit); a new vertex: a fresh record — and that IS the next DP's column records.
    int32_t *cnt = g.bestK;
    // the new position of the vertex at (current) position q; the record of the current numbering at q carried over to the next one
    auto newpos = [&](int q) -> int { return q + cnt[q]; };
    for (int q = lane; q <= n0; q += LANES) { cnt[q] = 0; g.onpK[q] = 0; }
    __threadfence_block();
    TPH(10);
    int lastEx = -1, nnew = 0;
    for (int c0 = 0; c0 < I; c0 += 2 * LANES) {                    // A: run lengths at the anchors, path flags (two blocks of loads in flight)
        int w[2], wn[2], ex[2];
        bool valid[2], isnew[2], lastOfRun[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = c0 + u * LANES + lane;
            valid[u] = i < I;
            const int ic = valid[u] ? i : I - 1;
            w[u] = g.pathv[ic];
            wn[u] = g.pathv[ic + 1 < I ? ic + 1 : ic];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = c0 + u * LANES + lane;
            isnew[u] = valid[u] && w[u] < 0;
            nnew += __popcll(__ballot(isnew[u]));
            const int incl = wave_scan_max_i32((valid[u] && !isnew[u]) ? i : -1);
            int e = wave_shr1_i32(incl, -1);
            e = e > lastEx ? e : lastEx;                           // last existing path element before i
            ex[u] = e;
            const int li = rl(incl, 63);
            lastEx = li > lastEx ? li : lastEx;
            lastOfRun[u] = isnew[u] && ((i + 1 >= I) || (wn[u] >= 0));
        }
        int av[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) av[u] = g.pathv[(lastOfRun[u] && ex[u] >= 0) ? ex[u] : 0];     // the anchor's position
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = c0 + u * LANES + lane;
            if (valid[u] && !isnew[u]) g.onpK[w[u]] = 1;           // (a vertex is on the path once: one writer)
            if (lastOfRun[u]) {
                const int ap = ex[u] >= 0 ? av[u] : -1;
                CHK(ap >= -1 && ap < n0, 106);
                cnt[ap + 1] = i - ex[u];                           // (distinct runs have distinct anchors: one writer)
            }
        }
    }
    nnew = rfl(nnew);
    if (n0 + nnew > vcap) { if (lane == 0) g.st[ST_OK] = 0; return; }
    __threadfence_block();
    TPH(11);
    int carry = 0;
    for (int c0 = 0; c0 <= n0; c0 += 4 * LANES) {                  // B: inclusive prefix sum of the run counts
        int cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = c0 + u * LANES + lane; cv[u] = cnt[q <= n0 ? q : n0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = c0 + u * LANES + lane;
            const int incl = wave_scan_add_i32(q <= n0 ? cv[u] : 0);
            if (q <= n0) cnt[q] = carry + incl;
            carry += rl(incl, 63);
        }
    }
    const int n = n0 + nnew;
    for (int q = lane; q < n; q += LANES) g.needK[q] = 0;
    __threadfence_block();
    // the record at current position q in the next numbering (position kn): in-edges remapped, the far flags of the DP recomputed
    auto carry_over = [&](int q, int kn, int4 rec, int4 &pxr) -> int4 {
        const int np = (rec.x >> 8) & 15;
        rec.x &= ~CREC_FAR0;
        if (np > 0) { rec.y = newpos(rec.y); if (kn - rec.y > PRING) { g.needK[rec.y] = 1; rec.x |= CREC_FAR0; } }
        if (np > 1) { rec.z = newpos(rec.z); if (kn - rec.z > PRING) g.needK[rec.z] = 1; }
        if (np > 2) { rec.w = newpos(rec.w); if (kn - rec.w > PRING) g.needK[rec.w] = 1; }
        if (np > 3) {
            pxr = pxc[q];
            for (int e = 3; e < np; ++e) { const int pn = newpos(int4_get(pxr, e - 3)); int4_set(pxr, e - 3, pn); if (kn - pn > PRING) g.needK[pn] = 1; }
        }
        return rec;
    };
    for (int q0 = 0; q0 < n0; q0 += 2 * LANES) {                   // C: the vertices off the path shift right
        int4 rec[2]; int kn[2]; bool go[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = q0 + u * LANES + lane;
            go[u] = q < n0 && !g.onpK[q < n0 ? q : 0];
            rec[u] = make_int4(0, -1, -1, -1); kn[u] = 0;
            if (go[u]) { rec[u] = crec[q]; kn[u] = newpos(q); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) if (go[u]) {
            const int q = q0 + u * LANES + lane;
            int4 pxr = make_int4(-1, -1, -1, -1);
            const int4 r = carry_over(q, kn[u], rec[u], pxr);
            CHK(kn[u] >= 0 && kn[u] < n, 105);
            crec_nx[kn[u]] = r;
            if (((r.x >> 8) & 15) > 3) px_nx[kn[u]] = pxr;
        }
    }
    TPH(12);
    lastEx = -1;
    int prevP = -1;                                                // new position of the path element before the block's first one
    for (int c0 = 0; c0 < I; c0 += LANES) {                        // D: the path's elements
        const int i = c0 + lane;
        const bool valid = i < I;
        const int w = g.pathv[valid ? i : I - 1];
        const bool isnew = valid && w < 0;
        const int incl = wave_scan_max_i32((valid && !isnew) ? i : -1);
        int ex = wave_shr1_i32(incl, -1);
        ex = ex > lastEx ? ex : lastEx;
        { const int li = rl(incl, 63); lastEx = li > lastEx ? li : lastEx; }
        const int av = g.pathv[(isnew && ex >= 0) ? ex : 0];       // a new vertex: the position of its run's anchor
        int4 rec = make_int4(0, -1, -1, -1);
        if (valid && !isnew) rec = crec[w];
        const int qa = valid ? (isnew ? (ex >= 0 ? av : 0) : w) : 0;
        int P = newpos(qa);                                        // existing: its own new position; new: the anchor's ...
        if (isnew) P = (ex >= 0 ? P + 1 : 0) + (i - ex - 1);       // ... + its place in the run (list head: the run starts at position 0)
        if (!valid) P = -1;
        int Pp = wave_shr1_i32(P, prevP);                          // new position of the path element before this one (-1: none)
        prevP = rl(P, 63);
        if (valid) {
            CHK(P >= 0 && P < n, 104);
            int4 pxr = make_int4(-1, -1, -1, -1);
            bool pxdirty = false;
            if (isnew) rec = make_int4(read_base_packed(sread, i) | ((Pp >= 0 ? 1 : 0) << 8) | (1 << CREC_NREADS_SHIFT), Pp, -1, -1);
            else {
                rec = carry_over(w, P, rec, pxr);
                rec.x += 1 << CREC_NREADS_SHIFT;                   // one more pass goes through the vertex
                if (Pp >= 0) {                                      // SPEC: the edge is appended unless present or the in-edge cap (7) is hit
                    const int np = (rec.x >> 8) & 15;
                    bool found = false;
                    if (np > 0) found |= rec.y == Pp;
                    if (np > 1) found |= rec.z == Pp;
                    if (np > 2) found |= rec.w == Pp;
                    for (int e = 3; e < np; ++e) found |= int4_get(pxr, e - 3) == Pp;
                    if (!found && np < CCSX_MAXPRED) {
                        if (np == 0) rec.y = Pp; else if (np == 1) rec.z = Pp; else if (np == 2) rec.w = Pp; else { int4_set(pxr, np - 3, Pp); pxdirty = true; }
                        rec.x += 1 << 8;
                        if (P - Pp > PRING) { g.needK[Pp] = 1; if (np == 0) rec.x |= CREC_FAR0; }
                    }
                }
            }
            crec_nx[P] = rec;
            if (((rec.x >> 8) & 15) > 3 || pxdirty) px_nx[P] = pxr;
        }
    }
    __threadfence_block();
    if (lane == 0) { g.st[ST_N] = n; g.st[ST_NADDED] += 1; g.st[ST_PAR] ^= 1; }
    (void)npoa;
    TPH(13);
}

// ---- k_poa_finish: consensus (heaviest path), draft, window bounds.  One wave per graph.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_poa_finish(KParams P, int z0, int pass, int b0)
{
    const int bx = (int)blockIdx.x + b0;                // (b0: the launch's first workgroup — the POA stage runs as two half-batches on two streams)
    const int lane = threadIdx.x;
    PoaSlot g = poa_slot(P, bx);
    if (z0 + bx >= P.n_zmw) return;
    if (!g.st[ST_LIVE]) return;
    const int z = rfl(P.zmw_perm[z0 + bx]);
    const int ok = rfl(g.st[ST_OK]), n = rfl(g.st[ST_N]), nadded = rfl(g.st[ST_NADDED]);
    const int4 *crec = g.st[ST_PAR] ? g.crec1 : g.crec0;
    const int32_t *pxw = (const int32_t *)(g.st[ST_PAR] ? g.px1 : g.px0);
    // ---- consensus: heaviest path (uniform walk).  The column records of the last prepass give every column's base, pass count and the
    // topological positions of its in-edges 0..2 by position: one coalesced load per 64 columns, no order -> vertex record -> rank
    // chain of dependent gathers (round 3; in-edges 3..7 still go through the vertex id)
    int Ld = 0, nw = 0, stat = -1;
    if (ok && n > 0) {
        int kbest = -1, sb = NEGV, best_prev = 0;
        // Forward pass, 64 columns at a time.  Most columns continue a chain (one in-edge, from the position before): along a chain the
        // heaviest-path weight is a prefix sum of the columns' own weights.  So a block needs ONE wave scan of the weights, a serial
        // (wave-uniform) resolution of its chain HEADS only — columns with no / several in-edges or an in-edge from elsewhere —
        // and two ds_bpermute to hand every column its head's value.  Integer arithmetic: the same numbers as the column-by-column walk.
        for (int kb = 0; kb < n; kb += LANES) {
            const int kkL = kb + lane;
            int4 cL = make_int4(0, -1, -1, -1);
            if (kkL < n) cL = crec[kkL];
            const int nblk = (n - kb) < LANES ? (n - kb) : LANES;
            const int npL = (cL.x >> 8) & 15;
            const int wL = kkL < n ? 2 * ((cL.x >> CREC_NREADS_SHIFT) & 127) - nadded : 0;
            const int Pw = wave_scan_add_i32(wL);                                  // inclusive prefix sum of the block's weights
            const bool headL = kkL < n && (lane == 0 || npL != 1 || cL.y != kkL - 1);
            unsigned long long heads = __ballot(headL);
            const unsigned long long headmask = heads;
            int hb = 0, bpv = kkL - 1;                                             // lane h: value / back pointer of head h (chain columns: the position before)
            while (heads) {
                const int j = (int)__ffsll((long long)heads) - 1;
                heads &= heads - 1;
                const int k = kb + j;
                const int np = rl(npL, j);
                int b = 0, p = -1;
                for (int q = 0; q < np; ++q) {
                    int pu;
                    if (q < 3) pu = q == 0 ? rl(cL.y, j) : (q == 1 ? rl(cL.z, j) : rl(cL.w, j));
                    else pu = rfl(pxw[(size_t)k * 4 + (q - 3)]);
                    CHK(pu >= 0 && pu < k, 108);
                    int bu;
                    if (pu >= kb) {                                                // inside the block: its head's value + the weights since
                        const int pj = pu - kb;
                        const int hd = 63 - __clzll((long long)(headmask & ((2ull << pj) - 1ull)));
                        bu = rl(hb, hd) + rl(Pw, pj) - rl(Pw, hd);
                    } else bu = pu == kb - 1 ? best_prev : rfl(g.bestK[pu]);
                    if (bu > b) { b = bu; p = pu; }
                }
                const int bv = b + rl(wL, j);
                if (lane == j) { hb = bv; bpv = p; }
            }
            const int hdL = 63 - __clzll((long long)(headmask & ((2ull << lane) - 1ull)));   // (lanes past the block: any head)
            int myBest = __shfl(hb, hdL) + Pw - __shfl(Pw, hdL);
            // a path never continues from a non-positive value (it starts anew: "b = max(0, ...)"), which breaks the prefix sum of a
            // chain: only where a graph begins or is junk.  Such a block is walked column by column.
            const unsigned long long nonpos = __ballot(kkL < n && myBest <= 0);
            if ((nonpos << 1) & ~headmask & (nblk == 64 ? ~0ull : (1ull << nblk) - 1ull)) {
                int myBp = -1; myBest = 0;
                int bprev = best_prev;
                for (int j = 0; j < nblk; ++j) {
                    const int k = kb + j;
                    const int np = rl(npL, j);
                    int b = 0, p = -1;
                    for (int q = 0; q < np; ++q) {
                        int pu;
                        if (q < 3) pu = q == 0 ? rl(cL.y, j) : (q == 1 ? rl(cL.z, j) : rl(cL.w, j));
                        else pu = rfl(pxw[(size_t)k * 4 + (q - 3)]);
                        const int bu = pu == k - 1 ? bprev : ((pu >= kb) ? rl(myBest, pu - kb) : rfl(g.bestK[pu]));
                        if (bu > b) { b = bu; p = pu; }
                    }
                    const int bv = b + rl(wL, j);
                    if (lane == j) { myBest = bv; myBp = p; }
                    bprev = bv;
                }
                bpv = myBp;
            }
            if (kkL < n) { g.bestK[kkL] = myBest; g.bpK[kkL] = bpv; }
            // the block's maximum, first position on ties (the walk's "bv > sb" in position order)
            const int key = wave_max_i32(kkL < n ? myBest * 64 + (63 - lane) : INT32_MIN);
            const int bmax = key >> 6;
            if (bmax > sb) { sb = bmax; kbest = kb + 63 - (key & 63); }
            best_prev = rl(myBest, nblk - 1);
            __threadfence_block();
        }
        // backtrack (uniform), bases collected in reverse into scratch
        uint8_t *tmp = (uint8_t *)g.pathv;
        int len = 0, k = kbest;
        while (k >= 0) {
            const int kb = (k >> 6) << 6;
            const int kk = kb + lane;
            const int bpL = kk < n ? g.bpK[kk] : -1;
            const int bL = kk < n ? (crec[kk].x & 255) : 0;
            // whole runs per step: `chain` marks the positions whose back pointer is the position before; from kl the path takes every position down to the
            // highest one at or below kl whose bit is clear (t), each lane stores its own base, and the walk continues at t's back pointer (one position per
            // step and a lane-0 store before: ~ 10 k dependent steps per graph)
            const unsigned long long chain = __ballot(kk < n && bpL == kk - 1);
            while (k >= kb) {
                const int kl = k - kb;
                const unsigned long long inv = ~chain & ((2ull << kl) - 1ull);
                const int t = inv ? 63 - __clzll((long long)inv) : 0;
                CHK(len + kl - t < n, 107);
                if (lane >= t && lane <= kl) tmp[len + (kl - lane)] = (uint8_t)bL;
                len += kl - t + 1;
                k = inv ? rl(bpL, t) : kb - 1;
            }
        }
        __threadfence_block();
        if (len <= P.dcap[z]) {
            uint8_t *draft = P.draft + P.seq_off[z];
            for (int q = lane; q < len; q += LANES) draft[q] = tmp[len - 1 - q];
            Ld = len;
        }
        __threadfence_block();
    }
    if (Ld <= 0) stat = CCSX_DRAFT_FAILURE;
    else if (Ld < P.opts.min_length) stat = CCSX_TOO_SHORT;
    else if (Ld > P.opts.max_length) stat = CCSX_TOO_LONG;
    else {
        nw = poa_windows(P, z, Ld, lane);
    }
    if (lane == 0) { P.draft_len[z] = Ld; P.nwin[z] = (stat < 0) ? nw : 0; P.zstat[z] = (stat < 0) ? CCSX_SUCCESS : stat; }
}

// ------------------------------------------------------------------------------------------------
// step 3: subread -> draft, global, adaptive 64-row band, one wave per read; the previous column lives
// in registers.  Instead of storing moves and tracing back every cell, each cell carries the row at which
// its best path ENTERED the most recent window-edge column ("origin"); at every window-edge column the
// propagated origins are saved (64 x int32), and the entry rows are recovered by hopping edge to edge.
// window-edge column of needed-column index k (k_align's list: 0, b1-2, b1+2, b2-2, ..., Ld)
__device__ __forceinline__ int need_col(const int32_t *wb, int nw, int Ld, int k)
{
    return k == 0 ? 0 : (k == 2 * nw - 1 ? Ld : wb[(k + 1) >> 1] + ((k & 1) ? -CCSX_WIN_OVERHANG : CCSX_WIN_OVERHANG));
}
struct AlignEnd { int M, O, lo; unsigned K; };
// the column loop of step 3 over one oriented read in LDS (REV: the draft is walked backwards, the window-edge columns are
// mirrored: the suffix-against-suffix half of the split alignment).  Saves (origin, dirty) of every cell, the band start and the
// best cell (score, row, entry row) at every window-edge column.
template <int REV>
__device__ __forceinline__ AlignEnd align_pass(const KParams &P, const uint32_t *sread, int I, const uint8_t *d, int Ld, const int32_t *wb, int nw,
                                               int32_t *Osave, int lane)
{
    const int nneed = 2 * nw;                           // needed columns: 0, b1-2, b1+2, ..., Ld
    int32_t *lo_need = Osave + (size_t)P.need_max * 128;
    int32_t *cm_need = lo_need + P.need_max + 64, *br_need = cm_need + P.need_max, *eb_need = br_need + P.need_max;   // best cell of every window-edge column (split alignment)
    int2 *OMsave = (int2 *)Osave;                      // per window-edge column and cell: (origin row at the previous edge, dirty bits)
    int kk = 1;                                         // next needed column index
    auto need_at = [&](int k) { return REV ? Ld - need_col(wb, nw, Ld, nneed - 1 - k) : need_col(wb, nw, Ld, k); };
    int next_need = rfl(need_at(1));
    // column 0 = START
    int Mprev = (lane <= I) ? lane * SC_INS : NEGV;
    int Oprev = 0;                                      // entry row at column 0 is 0 for every cell
    // candidate-filter pile-up (SPEC "dirty masks"): every cell carries the dirty bits of the draft positions its best path has
    // passed since the last window-edge column e: bit (j - e - 1) = position j-1 (column j) was not passed by a matching DIAG
    // step.  Leading insertions dirty position 0.  Bit 31: an insertion right at the edge column (dirties the LAST position of
    // the previous interval, patched in the epilogue).
    unsigned Kprev = (lane >= 1) ? 1u : 0u;
    int ecol = 0;                                       // last window-edge column
    int lo = 0, br = 0;
    int rbv = (lane >= 1 && lane <= I) ? read_base_packed(sread, lane - 1) : 4;   // base of row lo + lane (minus one), band at lo = 0
    const int hiI = I - (CCSX_BAND - 1) > 0 ? I - (CCSX_BAND - 1) : 0;
    BaseCursor bcur; bcur.w0 = 0; bcur.vw = 0u;
    const int nwords = (I + 15) >> 4;
    for (int jb = 0; jb < Ld; jb += LANES) {            // draft bases: one coalesced load per 64 columns
        const int dL = (jb + lane < Ld) ? d[REV ? Ld - 1 - (jb + lane) : jb + lane] : 0;
        cursor_load(bcur, sread, nwords, lo + 61, lane);  // the next 64 columns fetch their band-top bases from this window
        asm volatile("" :: "v"(dL), "v"(bcur.vw));       // wait for the block loads here, not inside the column loop
        const int nblk = (Ld - jb) < LANES ? (Ld - jb) : LANES;
        // two columns per loop iteration (round 4, as in k_align16): the carried cells rotate between two register sets
        auto column = [&](const int jj) {
            const int j = jb + jj + 1;
            const int plo = lo;
            lo = rfl(band_lo_s(plo, br, hiI));           // scalar unit; the readfirstlane tells the compiler the result is wave-uniform
            const int sh = lo - plo;                    // 0..2
            const int i = lo + lane;
            const int vb = rl(dL, jj);
            // the read base of row i-1 travels with the band: a band shift moves it one lane down and only the top lane(s)
            // fetch a new base (4 = no base: rows 0 and > I)
            int x, y, ox, oy;
            unsigned kx, ky;
            // (origins and dirty bits shift with a hardware zero fill: bound_ctrl, no register pre-loaded with the fill value)
            if (sh == 0) { x = wave_shr1_i32(Mprev, NEGV); y = Mprev; ox = wave_shr1_i32_z(Oprev); oy = Oprev; kx = (unsigned)wave_shr1_i32_z((int)Kprev); ky = Kprev; }
            else {
                const int top = lo + 62;                                // read index of lane 63's base
                if (sh == 1) {
                    const int nb = top < I ? base_at(bcur, top) : 4;
                    x = Mprev; y = wave_shl1_i32(Mprev, NEGV); ox = Oprev; oy = wave_shl1_i32_z(Oprev);
                    kx = Kprev; ky = (unsigned)wave_shl1_i32_z((int)Kprev);
                    rbv = wave_shl1_i32(rbv, nb);
                } else {
                    x = wave_shl1_i32(Mprev, NEGV); y = wave_shl1_i32(x, NEGV); ox = wave_shl1_i32_z(Oprev); oy = wave_shl1_i32_z(ox);
                    kx = (unsigned)wave_shl1_i32_z((int)Kprev); ky = (unsigned)wave_shl1_i32_z((int)kx);
                    const int nb1 = top - 1 < I ? base_at(bcur, top - 1) : 4;
                    const int nb = top < I ? base_at(bcur, top) : 4;
                    rbv = wave_shl1_i32(wave_shl1_i32(rbv, nb1), nb);
                }
            }
            // invalid cells (outside the band, row 0 for the diagonal) carry NEGV and simply lose every comparison; rows > I and
            // anything below NEGV/2 are reset to NEGV at the end of the column, so nothing accumulates
            const unsigned bitj = 1u << (j - ecol - 1);
            const bool match = (vb == rbv);
            int best = x + (match ? SC_MATCH : SC_MISMATCH), org = ox;
            unsigned kd = match ? kx : (kx | bitj);
            { const int c = y + SC_DEL; if (c > best) { best = c; org = oy; kd = ky | bitj; } }
            const bool need = (j == next_need);
            unsigned insbits = bitj | (bitj << 1);      // an insertion in column j dirties positions j-1 and j
            if (need) {
                // origin (previous edge) of the cell's entry move + dirty bits of the interval that ends here
                OMsave[(size_t)kk * 64 + lane] = make_int2(org, (int)kd);
                if (lane == 0) lo_need[kk] = lo;
                org = i;                                // reset: this column is the new edge
                kd = 0u; insbits = 0x80000001u; ecol = j;
            }
            // insertion chain with origin: the winner of lane l is the highest lane k <= l that attains the prefix maximum of
            // d = c + 4*lane (ties keep the higher lane, as the SPEC's serial chain does).  ONE max-scan finds value and lane
            // together: the key is (d << 6) | lane, with d clamped at -2^24 so that it fits (cells of valid paths are above
            // -2^19; anything below -2^22 is an invalid cell and is reset to NEGV at the end of the column).
            const int d0 = best + 4 * lane;
            const int dc = d0 > -(1 << 24) ? d0 : -(1 << 24);
            const int key = wave_scan_max_i32((dc << 6) | lane);
            const int xi = (key >> 6) - 4 * lane;
            const int ksl = key & 63;
            const int osrc = __shfl(org, ksl);                                   // origin / dirty bits of the cell the insertion run starts from
            const unsigned ksrc = (unsigned)__shfl((int)kd, ksl);
            if (xi > best) { best = xi; org = osrc; kd = ksrc | insbits; }
            if (i > I || best < -(1 << 22)) best = NEGV;
            const int cm = wave_reduce_max_i32(best);
            const unsigned long long bal = __ballot(best == cm);
            br = lo + (__ffsll((long long)bal) - 1);
            Mprev = best; Oprev = org; Kprev = kd;
            if (need) {
                const int ebest = rl(org, br - lo);          // row at which the best cell's path entered this column
                if (lane == 0) { cm_need[kk] = cm; br_need[kk] = br; eb_need[kk] = ebest; }
                ++kk;
                next_need = (kk >= nneed) ? -1 : rfl(need_at(kk));
            }
        };
        { int jj = 0; for (; jj + 2 <= nblk; jj += 2) { column(jj); column(jj + 1); } if (jj < nblk) column(jj); }
    }
    AlignEnd out; out.M = Mprev; out.O = Oprev; out.K = Kprev; out.lo = lo;
    return out;
}

// ------------------------------------------------------------------------------------------------
// step 3, first attempt of the alignment cascade (SPEC "alignment cascade"): FOUR passes of one ZMW per wave, each in a 16-row band
// that lives in one 16-lane DPP row — row shifts, the insertion-chain scan and the column maximum are native row operations
// (row_shr / row_shl with the hardware's fill, 4-step scans), the band position is a per-lane value that is uniform inside a row.
// The band follows the best row, so this finds the path of the 64-row band unless an indel run of more than ~8 rows occurs
// (bit-identical consensus on the test sets); a pass that is not valid here goes on a list for the 64-row retry (k_align), and
// from there to the split alignment.  Same cell recurrence as align_pass.
// Round 5 (second session): the cells no longer carry (origin, dirty bits) through the DP — three values to shift, select and permute per cell instead of one.
// Every cell's MOVE goes to HBM instead (2 bits: diagonal match / diagonal mismatch / deletion / insertion; a lane shifts the two predicate masks of a column into a
// register with two v_addc and stores it every 16 columns, 4 bytes per pass and column), plus the band's step per column (2 bits), and k_align16_tb — ONE LANE PER
// PASS, 64 passes per wave — walks the moves back and writes the entry rows and dirty masks of the window-edge columns: the same path (the moves are the forward
// pass's own decisions), so the same entries and masks, for a third fewer instructions in the kernel that saturates the VALU and half of its HBM writes.
#define AB16 16
#define NEG16 (-(1 << 23))           // k_align16's internal "minus infinity" (see the kernel)
#define AB16_ABOVE 6
#define AB16_SAT_ROWS 1               // SPEC v5 "band saturation": best row within the last AB16_SAT_ROWS rows of a band that can still move down
#define AB16_SAT_GAIN 1               //   ... or a window's worth of columns (edge k-2 -> edge k) without this much gain of the column maximum
#define AVALID_TB 2                   // avalid: valid in the 16-row band, entries / masks not yet traced back (k_align16 -> k_align16_tb, same stream)
// scratch of a quad (one k_align16 workgroup): per pass h [blocks of 16 columns][16 band rows] move words, then [blocks] (band steps, edge flags): 2 bits per column each,
// column j at bits 2 * (15 - ((j - 1) & 15)); edge flag of column j = "column j - 1 is a window-edge column" (the trace-back then needs no window bounds); then the four final band starts
__device__ __forceinline__ int tb_blocks(int Ld) { return ccsx_tb_blocks(Ld); }
__device__ __forceinline__ int tb_stride(int Ld) { return ccsx_tb_stride(Ld); }                  // (ccsx_kernels.h: shared with the host's slot size)
// acc * 2 + (the lane's bit of the wave mask m): one v_addc
__device__ __forceinline__ unsigned shift_in(unsigned acc, unsigned long long m)
{
    unsigned r; unsigned long long co;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(co) : "v"(acc), "s"(m));
    return r;
}
__device__ __forceinline__ int sel_i32(unsigned long long m, int a, int b)      // m ? a : b per lane, the mask already in scalar registers
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__global__ __launch_bounds__(64) void k_align16(KParams P, int qbase, int pass, int region)
{
    const int lane = threadIdx.x, h = lane >> 4, l = lane & 15, rowb = lane & 48;
    uint32_t *sread = dyn_lds;
    uint32_t *Osave = (uint32_t *)(P.align_scratch + ((size_t)region * P.align16_slots + blockIdx.x) * P.align16_slot_i32);
    if (qbase + (int)blockIdx.x >= P.n_quads) return;
    const int qd = rfl(P.quads[qbase + blockIdx.x]);    // (first pass << 2) | (passes - 1): up to four consecutive passes of one ZMW
    const int rfirst = qd >> 2, nq = (qd & 3) + 1;
    const int z = rfl(P.read_zmw[rfirst]);
    const int r0 = rfl(P.read_off[z]);
    const int zr = rfl(P.zref[z]);
    if (pass && !(zr & ZREF_PASSBIT(pass))) return;    // later passes: only the ZMWs whose draft was redone in that pass
    const bool live = h < nq;
    const int r = rfirst + (live ? h : 0);
    if (live && l == 0) { P.avalid[r] = 0; P.ascore[r] = NEGV; }
    if (P.zstat[z] != CCSX_SUCCESS) return;
    const int nused = rfl(P.nreads_used[z]);
    if (rfirst - r0 >= nused) return;
    const bool use = live && (r - r0 < nused);
    if (P.opts.disable_heuristics) {                    // SPEC v5: --disable-heuristics aligns every pass with the 64-row band at once
        if (l == 0 && use) { const int idx = atomicAdd(&P.align_retry[0], 1); P.align_retry[16 + idx] = r; }
        return;
    }
    const int Ld = rfl(P.draft_len[z]), nw = rfl(P.nwin[z]);
    const uint8_t *d = P.draft + P.seq_off[z];
    const int32_t *wb = P.wbounds + P.wb_off[z];
    const int wstride = CH16 / 16 + 3;                  // LDS words per pass: one guard word (the base "before" the chunk: row 0 of a band at 0 looks there),
                                                        // then one chunk of the pass at a time
    const int fl0 = rfl(P.flags[r0 + (zr & 255)] & 1);
    for (int hh = 0; hh < nq; ++hh) {
        const int rr = rfirst + hh;
        const int Ih = rfl((int)(P.base_off[rr + 1] - P.base_off[rr]));
        load_read_chunk(sread + hh * wstride + 1, P.bases + P.base_off[rr], Ih, rfl(((P.flags[rr] & 1) != fl0) ? 1 : 0), 0, lane);
    }
    if (lane < 4) sread[lane * wstride] = 0u;           // the guard words
    __syncthreads();
    int c0 = 0;                                         // first base of the row's chunk in LDS (per lane, uniform inside a row)
    int bi = l - 1;                                     // chunk index of the base of row i - 1 (the band starts at row 0)
    const int I = use ? (int)(P.base_off[r + 1] - P.base_off[r]) : 0;
    const uint32_t *myread = sread + h * wstride + 1;
    const int nneed = 2 * nw;
    const int tbs = tb_stride(Ld);
    uint32_t *mvL = Osave + (size_t)h * tbs + l;                        // the lane's move word of the current block of 16 columns
    uint32_t *shL = Osave + (size_t)h * tbs + tb_blocks(Ld) * 16;       // the pass's (band steps, edge flags) of the block (stored by lane 0 of the row)
    unsigned mvacc = 0u, shacc = 0u, edgeacc = 0u;
    bool prev_edge = false;                             // column j - 1 is a window-edge column other than column 0 (scalar)
    int kk = 1;
    // window-edge columns ahead: lane q holds column kkb + q of the list, refilled every 64 edges (a load + wait per edge otherwise)
    int kkb = 1;
    int needv = need_col(wb, nw, Ld, (kkb + lane < nneed) ? kkb + lane : nneed - 1);
    int next_need = rl(needv, 0);
    // (round 4) inside this kernel an invalid cell is NEG16 = -2^23 instead of NEGV = -2^28: still below every cell of a valid path (> -2^19) and below the
    // -2^22 threshold that resets a cell
    int Mprev = (l <= I) ? l * SC_INS : NEG16;
    int lo = 0, br = 0;                                  // per lane, uniform inside a row
    const int hiI = I - (AB16 - 1) > 0 ? I - (AB16 - 1) : 0;
    const int lane4 = 4 * lane;
    // the band never leaves the read (lo <= hiI = I - 15) unless the read is shorter than the band: only then can a row lie beyond the read and must be kept out of
    // the column maximum (wave-uniform: two VALU per column that ordinary passes do not need)
    const bool tiny = __any(use && I < AB16 - 1);
    // SPEC v5 "band saturation": the narrow band's answer is not trusted (-> 64-row retry) when the best row reaches the band's last row
    // before the band has reached the read's end, or when the column maximum gains less than AB16_SAT_GAIN between two window-edge
    // columns one window apart (cmE0 / cmE1: the maxima two edges / one edge back; edge 0 is column 0 with maximum 0)
    int satf = 0, cmE0 = 0, cmE1 = 0;
    unsigned long long satm = 0ull;
    for (int jb = 0; jb < Ld; jb += LANES) {
        const int dL = (jb + lane < Ld) ? d[jb + lane] : 0;
        asm volatile("" :: "v"(dL));
        // the 64 draft bases of the block as two wave masks: a column takes its base with scalar shifts (round 4: a v_readlane per column before — 6 cycles of the
        // VALU this kernel saturates)
        const unsigned long long dB0 = __ballot(dL & 1), dB1 = __ballot(dL & 2);
        const int nblk = (Ld - jb) < LANES ? (Ld - jb) : LANES;
        auto column = [&](const int jj) {
            const int j = jb + jj + 1;
            const int plo = lo;
            {   // band_lo with 16 rows, per row (the SPEC's final max(., 0) cannot bind: plo >= 0 and hiI >= plo)
                int t = br + 1 - AB16_ABOVE;              // 6 rows above the best row, 9 below: insertion bursts push the path down
                t = t > plo ? t : plo;
                t = t < plo + 2 ? t : plo + 2;
                lo = t < hiI ? t : hiI;
            }
            const int sh = lo - plo;                    // 0..2, per row
            bi += sh;                                   // index of base i - 1 in the row's chunk (no clamps: index -1 is the guard word, indices beyond the read
                                                        // hold zero bits; neither can reach a valid cell)
            shacc = (shacc << 2) | (unsigned)sh;
            edgeacc = (edgeacc << 2) | (prev_edge ? 1u : 0u);
            prev_edge = j == next_need;
            const int vb = (int)((dB0 >> jj) & 1ull) | ((int)((dB1 >> jj) & 1ull) << 1);
            // (every ~2000 columns per pass) the band reaches the end of a chunk: next chunk.  Looked at every 8th column only — the band moves at most two rows
            // per column, so 16 rows of margin cover the columns in between (the chunk holds CH16 + 32 bases)
            if ((jj & 7) == 0 && __any(lo + AB16 + 2 + 16 > c0 + CH16)) {
                for (int hh = 0; hh < nq; ++hh) {
                    const int src = hh << 4;
                    if (rl(lo, src) + AB16 + 2 + 16 > rl(c0, src) + CH16) {
                        const int rr = rfirst + hh;
                        const int Ih = rfl((int)(P.base_off[rr + 1] - P.base_off[rr]));
                        const int nc0 = rl(lo, src) - 16 > 0 ? rl(lo, src) - 16 : 0;
                        __syncthreads();
                        load_read_chunk(sread + hh * wstride + 1, P.bases + P.base_off[rr], Ih, rfl(((P.flags[rr] & 1) != fl0) ? 1 : 0), nc0, lane);
                        __syncthreads();
                        if (h == hh) { c0 = nc0; bi = lo + l - 1 - nc0; }
                    }
                }
            }
            // the read base of row i-1: an unconditional (clamped) LDS read issued here, consumed after the shifts below
            const uint32_t bw = myread[bi >> 4];
            // rows of the previous column in the new band position.  x = cell (i - 1) of the previous column is the previous column shifted by sh - 1 lanes (one
            // of three variants, selected per row — none at all when all four bands move down by one row, the common column); y = cell i is x shifted up by one
            // more lane, except for a band that did not move (its last lane keeps its own cell)
            // (no "no base" code for rows 0 and > I: row 0 has no diagonal source — its x is the shift's fill — and a row beyond the read is reset below, so
            // whatever base the clamped index fetches there cannot reach a valid cell)
            const int rbv = (int)((bw >> (2 * (bi & 15))) & 3u);
            const unsigned long long matchm = __builtin_amdgcn_uicmp((unsigned)vb, (unsigned)rbv, 32 /* eq */);
            int x, y;
            if (__all(sh == 1)) {                         // every band moves down by one row: no selects
                x = Mprev; y = row_shl1_i32(Mprev, NEG16);
            } else {
                const int mR = row_shr1_i32(Mprev, NEG16), m1 = row_shl1_i32(Mprev, NEG16);
                const bool s0 = sh == 0, s1 = sh == 1;
                x = s0 ? mR : (s1 ? Mprev : m1);
                const int yu = row_shl1_i32(x, NEG16);
                y = s0 ? Mprev : yu;
            }
            // the cell: diagonal, then deletion if strictly better, then the insertion chain inside the row (x_i = max(c_i, x_{i-1} + INS): one max-scan of
            // value + 4 * lane) if strictly better — the three decisions are the move
            const int dg = x + sel_i32(matchm, SC_MATCH, SC_MISMATCH);
            const int dl = y + SC_DEL;
            const unsigned long long delm = __builtin_amdgcn_sicmp(dl, dg, 38 /* gt */);
            int best = dl > dg ? dl : dg;
            const int xi = row_scan_max_i32(best + lane4) - lane4;
            const unsigned long long insm = __builtin_amdgcn_sicmp(xi, best, 38 /* gt */);
            best = xi > best ? xi : best;
            // move code of the cell: 0 diagonal match, 1 diagonal mismatch, 2 deletion, 3 insertion
            mvacc = shift_in(mvacc, delm | insm);
            mvacc = shift_in(mvacc, insm | ~(delm | matchm));
            if ((jj & 15) == 15) {                       // 16 columns of moves per lane, 16 band steps per row
                *mvL = mvacc; mvL += 16;
                if (l == 0) *(uint2 *)shL = make_uint2(shacc, edgeacc);
                shL += 2;
            }
            if (tiny && lo + l > I) best = NEG16;        // (a row beyond the read must not take part in the column maximum; an invalid cell inside the read may keep
                                                         // whatever it has below -2^22: it loses every comparison and cannot drift far in 65 k columns)
            // column maximum of the row and the lowest row that attains it, in every lane of the row: ONE all-reduce of (value * 16 + 15 - row) by four row rotations
            // (second session of round 5; before: a prefix scan, a ds_bpermute broadcast — an LDS round trip on the chain that places the next column's band —, a
            // ballot, a 64-bit shift and a find-first)
            const int ck = row_allmax_i32((best << 4) | (15 - l));
            const int cm = ck >> 4, brl = 15 - (ck & 15);
            br = lo + brl;
            satm |= __builtin_amdgcn_sicmp(brl, AB16 - AB16_SAT_ROWS, 39 /* >= */) & __builtin_amdgcn_sicmp(lo, hiI, 40 /* < */);   // (wave masks: two compares, the rest scalar)
            Mprev = best;
            if (j == next_need) {
                if (kk >= 2 && cm - cmE0 < AB16_SAT_GAIN) satf = 1;       // cmE0 = the maximum two edges back, cmE1 = one edge back (a rotation: an
                cmE0 = cmE1; cmE1 = cm;                                   // index kk & 1 made the compiler put the pair into scratch memory)
                ++kk;
                if (kk - kkb >= LANES) { kkb = kk; needv = need_col(wb, nw, Ld, (kkb + lane < nneed) ? kkb + lane : nneed - 1); }
                next_need = (kk >= nneed) ? -1 : rl(needv, kk - kkb);
            }
        };
        { int jj = 0; for (; jj + 2 <= nblk; jj += 2) { column(jj); column(jj + 1); } if (jj < nblk) column(jj); }
    }
    if (Ld & 15) {                                        // the last, partial block: its columns to the top of the word, like a full one
        const int s2 = 2 * (16 - (Ld & 15));
        *mvL = mvacc << s2;
        if (l == 0) *(uint2 *)shL = make_uint2(shacc << s2, edgeacc << s2);
    }
    const int oe = I - lo;
    const bool inr = oe >= 0 && oe < AB16;
    const int srcl = rowb + (inr ? oe : 0);
    const int scv = __shfl(Mprev, srcl);
    satf |= (int)((satm >> lane) & 1ull);
    const int sc = (inr && scv > -(1 << 22)) ? scv : NEGV;     // (an invalid end cell is reported as the SPEC's NEG)
    const int valid = (sc > NEGV / 2 && sc >= Ld && !satf) ? 1 : 0;
    if (l == 0 && use) {
        P.ascore[r] = sc;
        if (valid) {
            Osave[4 * (size_t)tbs + h] = (uint32_t)lo;          // where the trace-back starts: the band of the last column
            P.avalid[r] = AVALID_TB;
        } else {
            P.avalid[r] = 0;
            const int idx = atomicAdd(&P.align_retry[0], 1);        // -> the 64-row retry
            P.align_retry[16 + idx] = r;
        }
    }
}

// The trace-back of k_align16's passes: one LANE per pass, 64 passes per wave.  Walks the stored moves from (I, Ld) to (0, 0) and writes, for every window-edge
// column, the row at which the path ENTERS it (ent) and the dirty bits of the draft positions between two edge columns (dmask) — exactly what the cells used to
// carry forward (align_pass still does): bit (j - e - 1) of an interval that starts after edge column e for a mismatch or deletion at column j, bits of columns j
// and j + 1 for an insertion after column j; an insertion in an edge column itself belongs to the NEXT interval (bit 0) and dirties the last position of its own.
// The walk comes from above, so it collects an interval's bits counted from the interval's UPPER edge (bit eH - j) and reverses them when it reaches the lower one.
// The walk is a dependent chain per lane and its 64 lanes are 64 different passes, so (1) nothing in it may wait for HBM and (2) every per-lane global access is a
// fully divergent instruction — 64 cache lines, ~ 256 cycles of the address unit each (a first version with the loads in the walk took 20 ms per 16384 ZMWs, one
// with per-lane window bounds and 4-byte entry stores 12).  The wave therefore works in EPOCHS of TBE blocks of 16 draft columns: per epoch it copies every pass's
// move words of those columns to LDS with full-width loads (16 bytes per lane, a pass's 128 bytes contiguous), each lane fetches its band steps and edge flags with
// ONE 16-byte load, the walk itself touches LDS only, and the entries go out four at a time (16-byte stores at multiples of four edge columns).
#define TBE 2                          // blocks of 16 draft columns per epoch
#define TB_LSTRIDE (TBE * 16 + 2 * TBE + 1)             // words per lane: odd, so the lanes' rows sit in different banks
__global__ __launch_bounds__(64) void k_align16_tb(KParams P, int qbase, int nslots, int region)
{
    __shared__ uint32_t sS[64 * TB_LSTRIDE];
    __shared__ unsigned long long sOff[64];             // word offset of the pass's first staged move word in the alignment scratch
    __shared__ int sCnt[64];                            // move words of the pass in this epoch (0: none)
    const int lane = threadIdx.x;
    const int t = (int)blockIdx.x * 64 + lane;
    const int s = t >> 2, h = t & 3;
    bool act = s < nslots;
    int r = 0;
    if (act) {
        const int qd = P.quads[qbase + s];
        const int rfirst = qd >> 2, nq = (qd & 3) + 1;
        act = h < nq;
        r = rfirst + (act ? h : 0);
        act = act && P.avalid[r] == AVALID_TB;
    }
    if (!__any(act)) return;
    const int z = act ? P.read_zmw[r] : 0;
    const int Ld = act ? P.draft_len[z] : 0, nw = act ? P.nwin[z] : 1;
    const int I = act ? (int)(P.base_off[r + 1] - P.base_off[r]) : 0;
    const int tbs = tb_stride(Ld), nb = tb_blocks(Ld);
    const size_t sbase = ((size_t)region * P.align16_slots + s) * P.align16_slot_i32;
    const size_t moff = sbase + (size_t)h * tbs;
    const uint32_t *As = (const uint32_t *)P.align_scratch;
    const uint32_t *shw = As + moff + (size_t)nb * 16;
    int lo = act ? (int)As[sbase + 4 * (size_t)tbs + h] : 0;
    int32_t *ent = P.ent + P.ent_off[r];                // (ent_off is a multiple of 4: the 16-byte stores below)
    uint32_t *dm = P.dmask + P.ent_off[r];
    const int nneed = 2 * nw;
    uint32_t *my = sS + lane * TB_LSTRIDE;              // [0, 16 TBE) moves, then TBE (band steps, edge flags)
    uint32_t *mySh = my + TBE * 16;
    uint32_t e0 = 0u, e1 = 0u, e2r = 0u, e3 = 0u, d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;   // the last four entry rows / dirty masks, newest first
    int i = I, j = act ? Ld : 0, K = nneed - 1;         // interval K = columns (eL, eH] between the edge columns K - 1 and K
    int eH = Ld;
    unsigned mk = 0u, pend = 0u;                        // dirty bits of interval K counted from eH / of interval K + 1 (final form) until the insertions of edge column K are known
    bool atedge = true;                                 // the walk has just arrived at column eH
    int bhi = nb;                                       // the next epoch covers the blocks [b0, bhi), b0 even
    while (__any(j > 0)) {
        const bool on = j > 0;
        const int b0 = bhi > 0 ? (bhi - 1) & ~1 : 0;          // (a lane that has finished stays at j = 0 = its jstop)
        sOff[lane] = on ? (unsigned long long)(moff + (size_t)b0 * 16) : 0ull;
        sCnt[lane] = on ? (bhi - b0) * 16 : 0;
        // (every load of the epoch is issued before the first is used — unconditional, from a clamped address: a load under a branch is waited for at once, and
        // nine dependent round trips per epoch were what the second version spent its time on)
        const uint4 vs = *(const uint4 *)(on ? shw + 2 * b0 : As);
        __syncthreads();
        // the moves: 16 bytes per lane, TBE * 4 lanes per pass
        uint4 vm[TBE * 4];
#pragma unroll
        for (int q = 0; q < TBE * 4; ++q) {
            const int rd = q * (16 / TBE) + lane / (TBE * 4), part = lane % (TBE * 4);
            vm[q] = *(const uint4 *)(As + sOff[rd] + (part * 4 < sCnt[rd] ? part * 4 : 0));
        }
        mySh[0] = vs.x; mySh[1] = vs.y; mySh[2] = vs.z; mySh[3] = vs.w;
#pragma unroll
        for (int q = 0; q < TBE * 4; ++q) {
            const int rd = q * (16 / TBE) + lane / (TBE * 4), part = lane % (TBE * 4);
            if (part * 4 < sCnt[rd]) {
                uint32_t *dst = sS + rd * TB_LSTRIDE + part * 4;
                dst[0] = vm[q].x; dst[1] = vm[q].y; dst[2] = vm[q].z; dst[3] = vm[q].w;
            }
        }
        __syncthreads();
        const int jstop = b0 * 16;                      // column j lives in block (j - 1) >> 4
        // (the 64 lanes are 64 different passes, so every branch of the step is taken by some lane in nearly every iteration: the step is written with selects,
        // the last four entry rows / masks live in registers that shift at every edge column, and only the 16-byte stores are under a branch)
        while (j > jstop) {
            const int blk = ((j - 1) >> 4) - b0, o = (i - lo) & 15;
            int sft = 2 * (15 - ((j - 1) & 15));
            const uint32_t Wm = my[blk * 16 + o], Ws = mySh[2 * blk], We = mySh[2 * blk + 1];
            if (!atedge) {
                // Round 6: a RUN of plain matches per iteration.  A diagonal match with a band step of 1 keeps the band row (i - lo), so the run's moves are consecutive
                // 2-bit fields of the SAME move word, and its band steps / edge flags consecutive fields of the block's two other words: the run ends at the first
                // field that is not (move 0, step 1, no edge) — the fields shifted in above the word's last column read as "step 0", i.e. end it too.  Matches
                // touch neither the dirty bits nor the entry rows, so i, j and the band start just move by the run's length (one move per iteration before:
                // ~ 12 k iterations of a 64-way divergent walk per 10 kb pass, now ~ one per mismatch / indel / word)
                uint32_t bad = (Wm >> sft) | ((Ws >> sft) ^ 0x55555555u) | ((We >> sft) & 0x55555555u);
                bad = (bad | (bad >> 1)) & 0x55555555u;
                const int k = bad ? (__ffs((int)bad) - 1) >> 1 : 16;
                if (k > 0) {
                    i -= k; j -= k; lo -= k; sft += 2 * k;
                    if (i < 0) { j = 0; continue; }     // (cannot happen on a valid path: never spin on corrupt moves)
                    if (sft >= 32) continue;            // the run reached the word's last column: the next word at the loop's head
                }
            }
            const int code = (int)(Wm >> sft) & 3;
            const unsigned w2 = Ws >> sft, e2 = We >> sft;
            const bool ins = code == 3;
            // dirty bits: an insertion after column j = columns j and j + 1 (in an edge column: the next interval's first position and this one's last);
            // a mismatch or deletion = column j
            const unsigned bj = 1u << ((eH - j) & 31);
            mk |= ins ? (atedge ? 1u : bj | (bj >> 1)) : (code != 0 ? bj : 0u);
            if (ins && atedge) pend |= 1u;
            const bool emit = !ins && atedge;           // the path enters edge column K at row i
            if (emit) {
                e3 = e2r; e2r = e1; e1 = e0; e0 = (uint32_t)i;
                if ((K & 3) == 0) *(uint4 *)(ent + K) = make_uint4(e0, e1, e2r, e3);
                if (K + 1 < nneed) {
                    d3 = d2; d2 = d1; d1 = d0; d0 = pend & 0x7fffffffu;
                    if (((K + 1) & 3) == 0) *(uint4 *)(dm + K + 1) = make_uint4(d0, d1, d2, d3);
                }
                atedge = false;
            }
            if (code != 2) --i;
            if (ins) { if (i < 0) j = 0; continue; }    // (i < 0 cannot happen on a valid path: never spin on corrupt moves)
            lo -= (int)(w2 & 3u);                       // the band of column j - 1
            --j;
            if (e2 & 1u) {                              // column j is an edge column: next interval down, the finished one's bits counted from its lower edge
                const int len = eH - j < 32 ? eH - j : 32;
                pend = __brev(mk) >> (32 - len); mk = 0u; --K;
                eH = j;
                atedge = true;
            }
        }
        bhi = b0;
        __syncthreads();
    }
    if (act) {
        // column 0: the bits of interval 1; rows above the path's start are insertions before the first column (its bit 0).  The last edge column entered was 1.
        const int len = eH < 32 ? eH : 32;
        unsigned m1 = len > 0 ? __brev(mk) >> (32 - len) : 0u;
        if (i > 0) m1 |= 1u;
        *(uint4 *)ent = make_uint4(0u, e0, e1, e2r);
        *(uint4 *)dm = make_uint4(0u, m1 & 0x7fffffffu, d0, d1);
        P.avalid[r] = 1;
    }
}

// The 64-row retry of the alignment cascade: the passes k_align16 could not align in its 16-row band (a list it appended to),
// one wave per pass over a grid-stride loop.
__global__ __launch_bounds__(64) void k_align(KParams P, int pass)
{
    const int lane = threadIdx.x;
    uint32_t *sread = dyn_lds;
    int32_t *Osave = P.retry_scratch + (size_t)blockIdx.x * P.align_slot_i32;   // [need][64] then lo_need[need]
    const int nretry = rfl(P.align_retry[0]);
    for (int it = blockIdx.x; it < nretry; it += gridDim.x) {
    const int r = rfl(P.align_retry[16 + it]);
    const int z = rfl(P.read_zmw[r]);
    const int r0 = rfl(P.read_off[z]);
    const int zr = rfl(P.zref[z]);
    __syncthreads();
    const int Ld = rfl(P.draft_len[z]), nw = rfl(P.nwin[z]);
    const uint8_t *d = P.draft + P.seq_off[z];
    const int32_t *wb = P.wbounds + P.wb_off[z];
    const int I = rfl((int)(P.base_off[r + 1] - P.base_off[r]));
    const int rev = rfl(((P.flags[r] & 1) != (P.flags[r0 + (zr & 255)] & 1)) ? 1 : 0);
    load_read_packed(sread, P.bases + P.base_off[r], I, rev, lane);
    __syncthreads();
    const AlignEnd ae = align_pass<0>(P, sread, I, d, Ld, wb, nw, Osave, lane);
    const int lo = ae.lo, Mprev = ae.M, Oprev = ae.O; const unsigned Kprev = ae.K;
    const int nneed = 2 * nw;
    int32_t *lo_need = Osave + (size_t)P.need_max * 128;
    int2 *OMsave = (int2 *)Osave;
    const int oe = I - lo;
    int sc = NEGV, eLast = 0;
    unsigned kLast = 0u;
    if (oe >= 0 && oe < LANES) { sc = rl(Mprev, oe); eLast = rl(Oprev, oe); kLast = (unsigned)rl((int)Kprev, oe); }
    const int valid = (sc > NEGV / 2 && sc >= Ld) ? 1 : 0;
    __threadfence_block();
    if (lane == 0) {
        P.ascore[r] = sc; P.avalid[r] = (uint8_t)valid;
        if (valid) {
            int32_t *ent = P.ent + P.ent_off[r];
            uint32_t *dm = P.dmask + P.ent_off[r];
            int e = eLast;
            ent[nneed - 1] = e;
            // interval k2 = columns col(k2-1)+1 .. col(k2); a set bit 31 in interval k2+1 (insertion while sitting on edge k2)
            // dirties the last position of interval k2
            unsigned carry = kLast >> 31;               // trailing insertions after the last draft position
            for (int k2 = nneed - 1; k2 >= 1; --k2) {
                const int cell = e - lo_need[k2];
                const int2 om = OMsave[(size_t)k2 * 64 + cell];
                unsigned mk = (unsigned)om.y;
                const int cend = (k2 == nneed - 1) ? Ld : wb[(k2 + 1) >> 1] + ((k2 & 1) ? -CCSX_WIN_OVERHANG : CCSX_WIN_OVERHANG);
                const int cbeg = (k2 == 1) ? 0 : wb[k2 >> 1] + (((k2 - 1) & 1) ? -CCSX_WIN_OVERHANG : CCSX_WIN_OVERHANG);
                if (carry) mk |= 1u << (cend - cbeg - 1);
                carry = mk >> 31;
                dm[k2] = mk & 0x7fffffffu;
                if (k2 >= 2) { e = om.x; ent[k2 - 1] = e; }
            }
            ent[0] = 0; dm[0] = 0u;
        }
    }
    }
}

// SPEC "split alignment": a pass that failed step 3 and is more than RESCUE_MIN_EXCESS bases longer than the draft is tried as
// prefix + ONE large insertion + suffix ("spurious sequencing activity", docs/how-does-ccs-work.md:74-78; the 64-row band cannot
// follow an insertion run of more than ~31 rows).  The forward pass and the pass over the reversed read and draft each leave the
// best cell of every window-edge column; the split is the interior edge column with the largest sum of the two scores whose rows do
// not overlap.  Entry rows left of it come from the forward origins, right of it from the mirrored reverse origins; every position
// counts as dirty.  One wave per read over a grid-stride loop (rare reads: most iterations end at the first test); two scratch slots.
#define RESCUE_MIN_EXCESS 24
__global__ __launch_bounds__(64) void k_rescue(KParams P, int pass)
{
    const int lane = threadIdx.x;
    uint32_t *sread = dyn_lds;
    int32_t *OsF = P.retry_scratch + (size_t)(2 * blockIdx.x) * P.align_slot_i32, *OsR = OsF + P.align_slot_i32;
    for (int rbase = blockIdx.x * LANES; rbase < P.n_reads; rbase += gridDim.x * LANES) {
      // 64 passes per look: nearly all of them aligned
      unsigned long long todo = __ballot(rbase + lane < P.n_reads && !P.avalid[P.read_perm[rbase + lane < P.n_reads ? rbase + lane : 0]]);
      while (todo) {
        const int rp = rbase + (int)__ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int r = rfl(P.read_perm[rp]);
        const int z = rfl(P.read_zmw[r]);
        const int r0 = rfl(P.read_off[z]);
        const int zr = rfl(P.zref[z]);
        if (pass && !(zr & ZREF_PASSBIT(pass))) continue;
        if (P.zstat[z] != CCSX_SUCCESS || r - r0 >= P.nreads_used[z] || P.avalid[r]) continue;
        const int Ld = rfl(P.draft_len[z]), nw = rfl(P.nwin[z]);
        const int I = rfl((int)(P.base_off[r + 1] - P.base_off[r]));
        const int nneed = 2 * nw;
        const uint8_t *d = P.draft + P.seq_off[z];
        const int32_t *wb = P.wbounds + P.wb_off[z];
        const int fl = rfl((int)P.flags[r]);
        const int rev = rfl(((fl & 1) != (P.flags[r0 + (zr & 255)] & 1)) ? 1 : 0);
        if (fl & 2) {
            // SPEC "partial passes": anchored at one end of the draft (flag bit 2 = the adapter is at the pass's end; in draft orientation
            // that is the draft's end iff the pass is on the draft's strand).  The column loop runs from the anchored end; the pass covers
            // the draft up to the window-edge column with the largest column maximum (first on ties), valid iff that score reaches 1.0 per
            // covered base; uncovered edges get entry rows that give every window touching them a negative segment length
            if (nneed < 2) continue;
            const int from_end = ((fl >> 2) & 1) ^ rev;
            __syncthreads();
            load_read_packed_mode(sread, P.bases + P.base_off[r], I, from_end ? (rev ? 2 : 1) : (rev ? 3 : 0), lane);
            __syncthreads();
            if (from_end) (void)align_pass<1>(P, sread, I, d, Ld, wb, nw, OsF, lane);
            else (void)align_pass<0>(P, sread, I, d, Ld, wb, nw, OsF, lane);
            __threadfence_block();
            const int32_t *loF = OsF + (size_t)P.need_max * 128, *cmF = loF + P.need_max + 64, *ebF = cmF + 2 * P.need_max;
            int best = NEGV, bk = 1 << 30;
            for (int k = 1 + lane; k < nneed; k += LANES) {          // edge columns in the direction of the DP
                const int cf = cmF[k];
                if (cf > NEGV / 2 && cf > best) { best = cf; bk = k; }
            }
            const int wbest = rfl(wave_max_i32(best));
            const int ks = rfl(wave_min_i32(best == wbest ? bk : (1 << 30)));
            if (wbest <= NEGV / 2) continue;
            const int scol = from_end ? Ld - need_col(wb, nw, Ld, nneed - 1 - ks) : need_col(wb, nw, Ld, ks);   // covered draft columns
            if (scol <= 0 || wbest < scol) continue;
            if (lane == 0) {
                int32_t *ent = P.ent + P.ent_off[r];
                uint32_t *dm = P.dmask + P.ent_off[r];
                const int2 *OM = (const int2 *)OsF;
                int e = ebF[ks];
                if (!from_end) {
                    ent[ks] = e;
                    for (int k2 = ks; k2 >= 2; --k2) { e = OM[(size_t)k2 * 64 + (e - loF[k2])].x; ent[k2 - 1] = e; }
                    ent[0] = 0;
                    for (int k2 = ks + 1; k2 < nneed; ++k2) ent[k2] = -(1 << 20) - 64 * k2;
                } else {
                    ent[nneed - 1 - ks] = I - e;
                    for (int kq = ks; kq >= 1; --kq) {
                        e = (kq >= 2) ? OM[(size_t)kq * 64 + (e - loF[kq])].x : 0;
                        ent[nneed - kq] = I - e;
                    }
                    for (int k2 = 0; k2 < nneed - 1 - ks; ++k2) ent[k2] = (1 << 20) + 64 * (nneed - k2);
                }
                dm[0] = 0u;
                for (int k2 = 1; k2 < nneed; ++k2) dm[k2] = 0x7fffffffu;
                P.avalid[r] = 1; P.ascore[r] = wbest;
            }
            continue;
        }
        if (I - Ld <= RESCUE_MIN_EXCESS || nneed < 3) continue;
        __syncthreads();
        load_read_packed_mode(sread, P.bases + P.base_off[r], I, rev ? 3 : 0, lane);
        __syncthreads();
        (void)align_pass<0>(P, sread, I, d, Ld, wb, nw, OsF, lane);
        __syncthreads();
        load_read_packed_mode(sread, P.bases + P.base_off[r], I, rev ? 2 : 1, lane);     // the oriented read, last base first
        __syncthreads();
        (void)align_pass<1>(P, sread, I, d, Ld, wb, nw, OsR, lane);
        __threadfence_block();
        const int32_t *loF = OsF + (size_t)P.need_max * 128, *cmF = loF + P.need_max + 64, *brF = cmF + P.need_max, *ebF = brF + P.need_max;
        const int32_t *loR = OsR + (size_t)P.need_max * 128, *cmR = loR + P.need_max + 64, *brR = cmR + P.need_max, *ebR = brR + P.need_max;
        // the split: interior edge column k (mirror index nneed-1-k) with the largest score sum, rows not overlapping; ties: smallest k
        int best = NEGV, bk = 1 << 30;
        for (int k = lane; k < nneed; k += LANES) {              // k = 0 / nneed-1: an empty half (score 0, no rows): a block at the very start / end
            const int kr = nneed - 1 - k;
            const int cf = k ? cmF[k] : 0, cr = kr ? cmR[kr] : 0, a = k ? brF[k] : 0, b = kr ? brR[kr] : 0;
            if (cf > NEGV / 2 && cr > NEGV / 2 && a + b <= I) { const int tot = cf + cr; if (tot > best) { best = tot; bk = k; } }
        }
        const int wbest = rfl(wave_max_i32(best));
        const int ks = rfl(wave_min_i32(best == wbest ? bk : (1 << 30)));
        if (wbest <= NEGV / 2 || wbest < Ld) {
            // SPEC "double split" (v4): no single split column carries the pass (two insertions the band cannot follow).  It is used
            // as a prefix up to the edge column with the largest forward column maximum and a suffix from the edge column with the
            // largest reverse one (first on ties in the direction of each DP, as for partial passes), iff there are edge columns in
            // between, the parts share no read rows and each scores at least 1.0 per covered base; the edges in between get entry
            // rows that make every window touching them unusable for this pass (negative, or longer than the pass)
            int bF = NEGV, kF = 1 << 30, bR = NEGV, kR = 1 << 30;
            for (int k = 1 + lane; k < nneed; k += LANES) {
                const int cf = cmF[k], cr = cmR[k];
                if (cf > NEGV / 2 && cf > bF) { bF = cf; kF = k; }
                if (cr > NEGV / 2 && cr > bR) { bR = cr; kR = k; }
            }
            const int wF = rfl(wave_max_i32(bF)), wR = rfl(wave_max_i32(bR));
            if (wF <= NEGV / 2 || wR <= NEGV / 2) continue;
            const int k1 = rfl(wave_min_i32(bF == wF ? kF : (1 << 30))), kr = rfl(wave_min_i32(bR == wR ? kR : (1 << 30)));
            const int k2 = nneed - 1 - kr;
            const int s1 = need_col(wb, nw, Ld, k1), s2r = Ld - need_col(wb, nw, Ld, k2);
            if (!(k1 < k2 && s1 > 0 && s2r > 0 && wF >= s1 && wR >= s2r && brF[k1] + brR[kr] <= I)) continue;
            if (lane == 0) {
                int32_t *ent = P.ent + P.ent_off[r];
                uint32_t *dm = P.dmask + P.ent_off[r];
                const int2 *OMF = (const int2 *)OsF, *OMR = (const int2 *)OsR;
                int e = ebF[k1];
                ent[k1] = e;
                for (int kq = k1; kq >= 2; --kq) { e = OMF[(size_t)kq * 64 + (e - loF[kq])].x; ent[kq - 1] = e; }
                ent[0] = 0;
                int er = ebR[kr];
                ent[k2] = I - er;
                for (int kq = kr; kq >= 1; --kq) {
                    er = (kq >= 2) ? OMR[(size_t)kq * 64 + (er - loR[kq])].x : 0;
                    ent[nneed - kq] = I - er;
                }
                for (int kq = k1 + 1; kq < k2; ++kq) ent[kq] = -(1 << 20) - 64 * kq;
                dm[0] = 0u;
                for (int kq = 1; kq < nneed; ++kq) dm[kq] = 0x7fffffffu;
                P.avalid[r] = 1; P.ascore[r] = wF + wR;
            }
            continue;
        }
        if (lane == 0) {
            int32_t *ent = P.ent + P.ent_off[r];
            uint32_t *dm = P.dmask + P.ent_off[r];
            const int2 *OMF = (const int2 *)OsF, *OMR = (const int2 *)OsR;
            int e = ks ? ebF[ks] : 0;
            ent[ks] = e;
            for (int k2 = ks; k2 >= 2; --k2) { e = OMF[(size_t)k2 * 64 + (e - loF[k2])].x; ent[k2 - 1] = e; }
            ent[0] = 0;
            const int ksr = nneed - 1 - ks;                         // the same column in the mirrored list
            // reverse entries of the mirrored columns below ksr are the forward entries of the columns above ks: I - entry
            int er = ksr ? ebR[ksr] : 0;
            for (int kq = ksr; kq >= 1; --kq) {
                er = (kq >= 2) ? OMR[(size_t)kq * 64 + (er - loR[kq])].x : 0;
                ent[nneed - kq] = I - er;
            }
            dm[0] = 0u;
            for (int k2 = 1; k2 < nneed; ++k2) dm[k2] = 0x7fffffffu;
            P.avalid[r] = 1; P.ascore[r] = wbest;
        }
      }
    }
}

// per-ZMW: count usable reads, raise TOO_MANY_UNUSABLE (docs/faq/accuracy-vs-passes.md:37-39)
__global__ void k_post(KParams P, int pass)
{
    int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= P.n_zmw) return;
    const int zr = P.zref[z];
    if (pass && !(zr & ZREF_PASSBIT(pass))) return;
    // the draft cascade: a failed draft, or one most passes do not map to, is retried — pass 0 -> the fallback draft (pass 1), pass 1 ->
    // the last resort (pass 2); the last attempt's outcome is final
    const bool may_retry = !P.opts.no_fallback_draft && pass < 2;
    const int retry = pass == 0 ? ZREF_RETRY : ((zr & 255) | ZREF_RETRY2);
    if (P.zstat[z] != CCSX_SUCCESS) {
        P.np[z] = 0; P.out_fn[z] = 0; P.out_rn[z] = 0;
        if (may_retry && P.zstat[z] == CCSX_DRAFT_FAILURE) P.zref[z] = retry;
        return;
    }
    int r0 = P.read_off[z], nr = P.nfull[z], np = 0, rn = 0;     // full-length passes only: partial passes are not passes
    const int f0 = P.flags[r0 + (zr & 255)];
    for (int r = 0; r < nr; ++r) {
        const int v = P.avalid[r0 + r];
        np += v;
        if (v && ((P.flags[r0 + r] ^ f0) & 1)) ++rn;
    }
    P.np[z] = np; P.out_fn[z] = np - rn; P.out_rn[z] = rn;
    if (2 * np <= nr) {
        if (may_retry) P.zref[z] = retry;                  // try the next draft generator before giving up
        else { P.zstat[z] = CCSX_TOO_MANY_UNUSABLE; P.nwin[z] = 0; }
    }
}

// The windows of the batch in compact order (after the last k_post: nwin[z] is final).  The polish / kinetics grids used to cover every window SLOT
// (capacity: 1.25 x the longest pass / 19 + 4 per ZMW, a host-built map) — 31 % of the workgroups found no window and left, each after holding a workgroup's
// LDS for two dependent loads.  One workgroup: exclusive scan of nwin over the ZMWs, then the map (window index -> ZMW).
__global__ __launch_bounds__(1024) void k_wmap(KParams P)
{
    __shared__ int sTot[16], sCarry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sCarry = 0;
    __syncthreads();
    for (int z0 = 0; z0 < P.n_zmw; z0 += 1024) {
        const int z = z0 + tid;
        const int v = z < P.n_zmw ? P.nwin[z] : 0;
        const int incl = wave_scan_add_i32(v);
        if (lane == 63) sTot[wv] = incl;
        __syncthreads();
        int base = sCarry;
        for (int q = 0; q < wv; ++q) base += sTot[q];
        if (z < P.n_zmw) P.wstart[z] = base + incl - v;
        __syncthreads();
        if (tid == 1023) sCarry = base + incl;
        __syncthreads();
    }
    if (tid == 0) P.wstart[P.n_zmw] = sCarry;
}
__global__ __launch_bounds__(256) void k_wmap_fill(KParams P)     // one wave per ZMW: the map entries of its windows
{
    const int z = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (z >= P.n_zmw) return;
    const int a = P.wstart[z], nw = P.nwin[z];
    for (int w = lane; w < nw; w += LANES) P.wslot_zmw[a + w] = z;
}

// ------------------------------------------------------------------------------------------------
// A1-A7 + step 7: Arrow polish of one window per workgroup (PW_THREADS = 4 waves).
//
// v4 (round 3; tables: round 6).  LDS holds: sCTX[obs][ctx] = (ME, INS) as float2, 17 entries per observation row (16 contexts, one zero entry;
// row 12 = all zeros): the scoring's table; per-column entries (DL, offset of the column's context in a row of sCTX) for a long read's sweep (sColJ);
// the quad fill's per-column table sTC[strand][obs][column] + sDLC (see TC_LO below); gamma/beta of one chunk of up to EIGHT reads
// (sGB, even row stride: the fill's lane = row stores step by S - 1 words from lane to lane, the scoring's lane = column loads by 1:
// both conflict free).  Fill: lane = read row, anti-diagonal sweep, neighbours via DPP wave shifts; two short reads per wave (lanes
// 0-31 / 32-63); a chunk's two pair tasks run as four alpha-only / beta-only sweeps, one per wave.  Candidate filter
// (docs/how-does-ccs-work.md:80-83): the step-3 alignments' dirty bits give a per-position pile-up margin; unambiguous
// non-homopolymer positions enumerate no mutations.  Scoring: a pool of (64 compacted mutation lanes) x (usable read) units over the
// waves, serial over read rows exactly as the SPEC orders the operations; per-read gains are summed in 2^-16 fixed point with LDS
// integer atomics, so the sum does not depend on which wave scored which read.
#ifndef PW_THREADS
#define PW_THREADS 256                // 4 waves.  Round 2 (ms per 2048 ZMWs, 10 x 10 kb): 256 threads x 3 workgroups/CU 95, 512 x 2 (all ten reads in one
#endif                                //   LDS chunk) 101-103, 256 x 2 141, 320 x 2 208.  Round 3 (ms per 8192): 256 x 4 155, 384 x 3 225, 384 x 2 234-255, 512 x 2 229-260
#ifndef PW_MINWAVES
#define PW_MINWAVES 4
#endif
#define PW_WAVES (PW_THREADS / 64)
#ifndef PW_CHUNK_READS
#define PW_CHUNK_READS 8              // reads per gamma/beta chunk (0: as many as fit).  A chunk costs one fill sweep + one scoring pass however many
                                      // reads it holds (ms of k_polish per 8192 ZMWs: 2 reads per chunk 234, 4 reads 169.5, as many as fit = 4-5 reads 175.7):
                                      // four reads = two pair tasks = four alpha-only / beta-only units = every wave busy for ONE sweep, a fifth read
                                      // adds a task and turns the units into full alpha+beta sweeps on three waves.  Eight-wave workgroups with 8 / 10
                                      // reads per chunk (2 per CU, 80 KB each) ran 260 / 229 ms: profiles/r03_polish_fill_variants.txt
#endif
#ifndef PW_MAXREADS
#define PW_MAXREADS 32                // passes per GROUP: the per-read arrays, observations and chunk plan of k_polish / k_kinetics describe one group.  Round 4
                                      // (profiles/r04_group_size.txt, ms of k_polish at 10 passes / the configs[4] mix / 30 passes x 20 kb): 64 157 / 311 / 861 with 16-bit
                                      // observation codes; one byte per observation: 32 156 / 254 / 705, 16 156 / 258 / 755, 12 155 / 260 / 763
#endif
#ifndef PW_LDS_BYTES
#define PW_LDS_BYTES 40960            // static + dynamic LDS of one workgroup: FOUR workgroups per CU fill its 160 KB exactly.  Round 3 sweep (ms of
                                      // k_polish per 8192 ZMWs 10 x 10 kb, alone / under the draft stage of the next batch): 52992 (3 per CU) 217.9 / 305.3,
                                      // 40960 184.6 / 264.8, 40448 189.6 / 273.3, 38912 195.0 / 279.3, 36864 199.7 / 281.1, 32256 (5 per CU) 225.6 / 307.8
#endif
#ifndef CTXS
#define CTXS 17                       // entries per observation row of sCTX: 16 contexts + one zero entry (odd stride: (obs + ctx) mod 16 spreads the scoring's look-ups over the bank pairs)
#endif
#define FILL16_MAXBW 24               // widest band (diagonals) of a read that shares a wave with three others (k_polish's quad sweep)
// Round 6: the quad fill's per-COLUMN table.  sTC[strand][obs][column] = (ME, INS) of the column's context for that observation (row 12 and the columns from
// J on: zeros) — the lane of a quad unit (a read row: fixed observation, a new column every step) reads it with ONE ds_read_b64 at a running address + immediate;
// the 16 lanes of a DPP row are on 16 CONSECUTIVE columns, so whatever their observations they hit 16 different bank pairs (row stride 32 entries).  Until round 5 a
// lane took the column's context offset from a staggered per-column entry and then the pair from its observation's row of sCTX[obs][ctx]: two dependent LDS reads,
// the second at (obs + ctx) mod 16 — 42 % of the kernel's LDS cycles were bank conflicts.  sDLC[strand][column] = DL of the column's context (1 outside 0 .. J-1).
// A lane forms column indices from TC_LO .. 31 + TC_HI while it is off the band; what it reads there is never used but has to be FINITE (0 * x), so the tables sit
// between guards of finite floats: [sDL, sZP, padding] in front, sDLC behind.
#define TC_LO 56                      // columns below 0 / above 31 a quad lane may form: alpha -27 .. 67, beta -53 .. 62 (a read much shorter than its quad's longest keeps
#define TC_HI 38                      // stepping after its own last column), DL one column lower
#define TC_ROW 32                     // entries per observation row
#define TC_STRAND ((CCSX_NOBS + 1) * TC_ROW)                 // entries per strand
#define DLC_S (33 + TC_LO)            // sDLC: [TC_LO pad | strand 0: columns 0 .. 32 | TC_LO pad | strand 1 | TC_HI pad] — the middle pad serves both strands
#define DLC_TOTAL (TC_LO + DLC_S + 33 + TC_HI)
#define TAB_TAIL (2 * TC_HI - 48)     // floats of padding behind sDL / sZP, which with them are the guard behind sTC
#define OBS_TC(o) ((o) * (TC_ROW * 8))         // the byte offset of observation o's row in a strand of sTC
#define OBS_CODE(o) ((o) * (CTXS * 8))        // the byte offset of observation o's row in sCTX (sObs holds the 8-bit observation itself)

struct LaneMut {                     // per-lane constants of one mutation on one strand
    int c, q, kA, kB, isdel, fin;    // kA / kB index a row of sCTX; 16 = the zero entry (see lane_mut)
    int qoff;                        // q - c - 1 (0: insertion, 1: substitution / deletion): how far beta(i+1, q) sits above gamma(i, c) in diagonals (SPEC v8 joint band test)
    float dlA, dlL;
};

__device__ __forceinline__ LaneMut lane_mut(int type, int c, int x, const uint8_t *t, int J, int lf, const float *sDL)
{
    // branch-free: both template neighbours are loaded unconditionally (clamped indices; sT has 32 entries), then selected
    LaneMut L;
    const int tm1 = t[c > 0 ? c - 1 : 0];
    const int nidx = (type == 2) ? c : c + 1;
    const int tn = t[nidx < 31 ? nidx : 31];
    const int fin = (type == 2) ? (c == J) : (c + 1 == J);
    const int nx = fin ? 0 : tn;                             // the template base after the mutated column (0 when there is none)
    int pA = (c > 0) ? tm1 : lf;
    const int xA = (type == 1) ? nx : x;
    if (pA > 3) pA = (xA + 2) & 3;
    const int kA = pA * 4 + xA, kB = (type == 1) ? kA : xA * 4 + nx;
    const int q = (type == 2) ? c + 1 : c + 2;
    L.c = c; L.q = q > J ? J : q; L.isdel = (type == 1); L.fin = fin; L.qoff = (type == 2) ? 0 : 1;
    L.dlA = sDL[kA]; L.dlL = sDL[kB];
    // the SPEC's "no stay move" cases read the ZERO entry of the observation's row: a deletion whose extension is the final column (INS[kA] is unused and
    // b = a, so ME[kA] is not needed either), and any extension that reaches the final column (INS[kB] unused; ME[kB] only feeds the link, which such a lane does
    // not report).  Until round 5 the table carried a second half of (ME, 0) copies for them: 1.6 KB of LDS
    L.kA = (type == 1 && fin) ? (CTXS - 1) : kA;
    L.kB = fin ? (CTXS - 1) : kB;
    return L;
}

// 32-bit LDS pointers (address space 3): they survive being made opaque to the optimiser as LDS pointers (ds_read, not flat)
typedef const float __attribute__((address_space(3))) *lds_cf;
typedef const char __attribute__((address_space(3))) *lds_cc;

typedef const uint8_t __attribute__((address_space(3))) *lds_cu8;
struct ScoreChain {                  // running state of one (lane, read) mutation evaluation
    float ap, bp, acc, b, bq;
    float2 pA, pB;
    lds_cf g, be;                    // gamma(i, c) and beta(i+1, q) of the row about to be processed: both advance by the read's pitch per row
    lds_cu8 op;                      // observation of that row (0..11, 12 = none)
    int dg;                          // SPEC v6/v8: gamma's diagonal relative to the read's band, c - i - dlo; the row's two cells are on the band iff 0 <= dg <= bw - 1 - (q - c - 1)
};

// v in the lanes of the wave mask m, 0 elsewhere: one v_cndmask on a mask that already sits in scalar registers
__device__ __forceinline__ float lanes_or_zero(unsigned long long m, float v)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

// one row of the extend+link recursion.  tA / tB = the lane's columns of sCTX (context kA / kB).  Every lane walks
// its own rows (the SPEC's band around the window diagonal starts at a per-lane row), so the row's observation code comes from LDS;
// row I of a read carries code 12 = the all-zero row of sCTX ("no base left": the SPEC's i < I cases become exact +0 products).
// SPEC v6: gamma / beta are stored on the read's band only; a scoring band that is clamped into a corner of the window reads cells outside it, which are zeros
// (the address of such a cell aliases a neighbouring row's: the value is replaced, never used).
__device__ __forceinline__ void score_step(ScoreChain &s, const LaneMut &L, lds_cc tA, lds_cc tB, int pitch, unsigned lim)
{
    typedef const unsigned long long __attribute__((address_space(3))) *lds_cu64;
    // SPEC v8 "joint band test": gamma(i, c) and beta(i+1, q) of the row sit q - c - 1 diagonals apart; both are zeros unless both are on the band — one compare
    // (lim = bw - 1 - (q - c - 1)) whose mask serves both selects (v7: a test per cell)
#ifndef CCSX_EXP_NO_BANDMASK                                // experiment (timing only, wrong results in the corners of a window): what the band test costs
    const unsigned long long on = __builtin_amdgcn_uicmp((unsigned)s.dg, lim, 37 /* ule */);
    const float gmm = lanes_or_zero(on, *s.g);
    const float bqn = lanes_or_zero(on, *s.be);
#else
    const float gmm = *s.g, bqn = *s.be;
#endif
    const int o256 = __mul24((int)*s.op, CTXS * 8);   // byte offset of the observation's row in sCTX
    const unsigned long long va = *(lds_cu64)(tA + o256), vb = *(lds_cu64)(tB + o256);   // one ds_read_b64 each
    const float2 nA = make_float2(__uint_as_float((unsigned)va), __uint_as_float((unsigned)(va >> 32)));
    const float2 nB = make_float2(__uint_as_float((unsigned)vb), __uint_as_float((unsigned)(vb >> 32)));
    s.g += pitch; s.be += pitch; s.op += 1; s.dg -= 1;
    const float insA = s.pA.y, meA = s.pA.x, insB = s.pB.y;
    // SPEC v8 "fused recurrences": every multiply-add is one fma, nested as written in DESIGN.md §2
    const float a = __builtin_fmaf(s.ap, insA, gmm);
    float b;
    if (L.isdel) b = a;
    else b = __builtin_fmaf(s.bp, insB, __builtin_fmaf(a, L.dlA, s.ap * meA));
    s.acc = __builtin_fmaf(b, __builtin_fmaf(L.dlL, s.bq, nB.x * bqn), s.acc);
    s.ap = a; s.bp = b; s.b = b; s.pA = nA; s.pB = nB; s.bq = bqn;
}

// first row of a lane's scoring band: floor((2 c I + J) / (2 J)) - Wr, clamped so that nrows rows fit into 0..I (SPEC "banded link");
// the quotient through a float reciprocal with an exact remainder correction (all values < 2^12)
__device__ __forceinline__ int band_row0(int c, int I, int J2, float invJ2, int J, int Wr, int nrows)
{
    const int num = __mul24(2 * c, I) + J;
    int q = (int)((float)num * invJ2);
    const int r = num - __mul24(q, J2);
    q += (r >= J2 ? 1 : 0) - (r < 0 ? 1 : 0);
    int i0 = q - Wr;
    i0 = i0 < 0 ? 0 : i0;
    const int hi = I + 1 - nrows;
    return i0 > hi ? hi : i0;
}

// fixed-point image of one read's log2-likelihood gain (SPEC "integer sum over reads")
__device__ __forceinline__ int dq_fix(float d)
{
    if (d < -DQ_CLAMP) d = -DQ_CLAMP;
    if (d > DQ_CLAMP) d = DQ_CLAMP;
    return (int)floorf(d * DQ_SCALE + 0.5f);
}

// error probability reported for a position the candidate filter skipped (pile-up margin g = clean - dirty)
__device__ __forceinline__ float skip_perr(int g, float floor_)
{
    if (g < 0) g = 0;
    if (g > 12) g = 12;
    const float p = 8.0f * det_exp2f(-3.0f * (float)g);
    return p < floor_ ? floor_ : p;                          // SPEC v7: the pile-up supports no claim beyond Q50 (opts.max_qv)
}

// SPEC v7 "repeat-count floor": lane x's longest period-p tandem tract among the visible window bases, from the wave mask m (bit k: v[k] == v[k+p]); 0 = none
// (SPEC v8: bit 8 of the result = the tract is OPEN, i.e. runs into an end of the visible template of nvis bases; on equal lengths the flags are ORed)
__device__ __forceinline__ int tract_len(unsigned long long m, int x, int p, int nvis)
{
    const int lo = x - p < 0 ? 0 : x - p;
    const unsigned long long cand = m & ((2ull << x) - 1ull) & ~((1ull << lo) - 1ull);      // the runs that reach x start their last equality at k in [x - p, x]
    if (!cand) return 0;
    int best = 0;
#pragma unroll
    for (int which = 0; which < 2; ++which) {                // the lowest and the highest such k: the same run, or the two runs that touch x
        const int k = which ? 63 - __clzll((long long)cand) : __ffsll((long long)cand) - 1;
        const unsigned long long zb = ~m & ((1ull << k) - 1ull);
        const int i = zb ? 64 - __clzll((long long)zb) : 0;
        const int j = k + (__ffsll((long long)(~m >> k)) - 1);
        const int L = j - i + p, op = ((i == 0) || (j + p >= nvis)) ? 256 : 0;
        if (L > (best & 255)) best = L | op; else if (L == (best & 255)) best |= op;
    }
    return best;
}


// SPEC v6 "banded fill": the diagonals d = j - i on which alpha / beta of a (read, window) pair exist (exact zeros elsewhere), and how gamma / beta are laid out in
// LDS.  A band narrower than a matrix row is stored BY DIAGONAL: cell (i, j) at i * (BWp - 1) + (j - dlo), BWp = BW | 1 floats per row (odd: the fill's lane stride
// BWp - 2 and the scoring lanes' ~ BWp are then conflict free) — 15 floats per row instead of 28 at I = J, so a chunk holds twice the reads.  A band that wide or wider
// (|I - J| >= 5: rare) keeps the row layout i * S + j.  Both are "org + i * pitch + j".
struct FillBand { int dlo, dhi, bw, pitch, org, rowsz; };
__device__ __forceinline__ FillBand fill_band_of(int I, int J, int S)
{
    FillBand b;
    const int dIJ = I > J ? I - J : J - I;
    const int W = SCORE_BAND + FILL_MARGIN + (dIJ > 2 ? dIJ - 2 : 0);
    const int lo = (J - I < 0 ? J - I : 0) - W, hi = (J - I > 0 ? J - I : 0) + W;
    b.dlo = lo < -I ? -I : lo; b.dhi = hi > J ? J : hi;
    b.bw = b.dhi - b.dlo + 1;
    const int bwp = b.bw | 1;
    const bool diag = bwp < S;
    b.rowsz = diag ? bwp : S; b.pitch = diag ? bwp - 1 : S; b.org = diag ? -b.dlo : 0;
    return b;
}

// the windows a launch piece covers, and the XCD-contiguous order of its blocks: block b -> the (b / 8)-th window of the (b % 8)-th eighth; -1 = no window
__device__ __forceinline__ int polish_piece_windows(int total, int slot0, int grid) { const int c = total - slot0; return c < 0 ? 0 : (c > grid ? grid : c); }
__device__ __forceinline__ int xcd_contiguous(unsigned b, int count)
{
    const int per = (count + 7) >> 3;
    const int k = (int)(b >> 3), r = (int)(b & 7u) * per + k;
    return (k < per && r < count) ? r : -1;
}

// PWT threads, PWMIN workgroups' worth of waves per SIMD, PWCH reads per gamma/beta chunk: ONE instantiation is shipped, 256 x 4 x 4.  Round 4 measured
// the 512-thread shape (2 workgroups of 80 KB per CU, 8 reads per chunk) at 14 / 18 / 24 passes and on the configs[4] mix: 2.2 / 2.3 / 1.8 / 1.7 times
// SLOWER than this one (profiles/r04_c4_shapes.txt; its apparent 2.4x win at 30 passes x 20 kb came from a grid of more than 2^32 threads that covered 23 % of the windows).
template <int PWT, int PWMIN, int PWCH>
__global__ __launch_bounds__(PWT, PWMIN) void k_polish_t(KParams P, int slot0)
{
    // [obs][ctx] = (ME, INS); entry 16 of a row and row 12 = zeros ("no base").  Odd row stride: with the banded link every lane looks up its OWN
    // observation row, and with an even stride all lanes of one context hit the same bank pair whatever their rows (35 % of the LDS cycles were bank
    // conflicts in round 2); 17 spreads them by (obs + ctx) mod 16
    __shared__ float2 sCTX[(CCSX_NOBS + 1) * CTXS];
    // one float array: [sDLC] [sTC: 2 strands x 13 observation rows x 32 columns x (ME, INS)] [sDL 16 | sZP 32 | padding] — the first and the last part
    // double as the finite guards of sTC (see TC_LO): 2 TC_LO floats in front, 2 TC_HI behind
    static_assert(DLC_TOTAL >= 2 * TC_LO && TAB_TAIL >= 0 && (DLC_TOTAL & 1) == 0, "sTab: guards / alignment of sTC");
    __shared__ __attribute__((aligned(16))) float sTab[DLC_TOTAL + 2 * TC_STRAND * 2 + 48 + TAB_TAIL];
    float *const sDLC = sTab;
    float2 *const sTC = (float2 *)(sTab + DLC_TOTAL);
    float *const sDL = sTab + DLC_TOTAL + 2 * TC_STRAND * 2, *const sZP = sDL + 16;         // sZP: z-score MU[16], VAR[16]
    __shared__ uint8_t sT[2][32];                            // template: [0] forward, [1] reverse complement
    __shared__ uint8_t sTd[32];                              // the draft's window as it was (the large-insertion trim of a reloaded group compares with it)
    // dynamic LDS: observation codes of the batch's largest ZMW ([reads][68]: 63 codes + look-ahead slack), then gamma/beta.
    // One BYTE per read row (round 3 stored the 16-bit byte offset of the observation's sCTX row: at 30 passes those 4 KB left room for only three
    // reads per gamma/beta chunk; the scoring rows are bound by their LDS round trips, the extra multiply per row is not measurable).
    uint8_t (*sObs)[68] = (uint8_t (*)[68])dyn_lds;
    float *sGB = (float *)((uint8_t *)dyn_lds + P.pw_obs_bytes);
    const int GB_FLOATS = P.pw_gb_floats;
    __shared__ int sI[PW_MAXREADS], sGoff[PW_MAXREADS], sBoff[PW_MAXREADS];   // segment length; where gamma / beta (i, j) = sGB[off + i * pitch + j] of the chunk's reads start
    __shared__ int sBand[PW_MAXREADS];                       // the read's band and layout (SPEC v6): pitch | rowsz << 8 | bw << 16 | (dlo + 128) << 24
    __shared__ unsigned sDirty[PW_MAXREADS];                 // window-relative pile-up dirty bits of each read
    __shared__ uint8_t sStrand[PW_MAXREADS], sValid[PW_MAXREADS];
    __shared__ unsigned sZdrop[(CCSX_MAX_PASSES + 32) / 32]; // z-score gate, one BIT per pass (all groups; a group is one word): decided on the draft window (round 0), then kept
    __shared__ float sBase[PW_MAXREADS], sB00[PW_MAXREADS];   // alpha(I,J) / beta(0,0) of the chunk's reads (the fill's two halves meet here)
    __shared__ short2 sTask[PW_MAXREADS];                    // fill tasks: (read A, read B or -1)
    __shared__ int sDeltaI[256];                             // fixed-point sums of the per-read gains; converted in place to float
    float *sDelta = (float *)sDeltaI;                        // (each thread converts its own entry after the scoring barrier)
    __shared__ uint8_t sMvalid[256];
    __shared__ int sCtl[12];                                 // 0:J 1:cs 2:ce 3:nacc 7:ev bits
    __shared__ float sZS[4];                                 // z-score sums: M fwd, V fwd, M rev, V rev
    __shared__ float sPskip[36];                             // error probability of a position if it is skipped (travels with the base)
    __shared__ int sCnt[(PWT / 64)];
    __shared__ uint8_t sList[256];                           // compacted valid mutation lanes (a lane index < 256)

    const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);      // wave-uniform values are made scalar explicitly (rfl):
    // loop counters, read indices and observation codes then live in SGPRs and cost no vector instructions
    PHASE_T0();
    // ---- locate (zmw, window).  The prologue is a chain of dependent global loads; every level issues all of its
    // loads before the first use (clamped indices instead of branches), so the chain is 4 round trips deep
    // (z-level scalars -> window bounds + per-read metadata + tables -> entry rows -> segments), not one per array.
    // XCD-aware order: block b runs on XCD b % 8 (observed, a speed matter only), every XCD has its own L2, and neighbouring windows of a ZMW share the lines of
    // their entry rows, dirty masks and read segments — so XCD x takes the x-th CONTIGUOUS eighth of the windows there are, not every eighth window
    // (slot0: a batch of more than 2^24 - 256 window slots is launched in pieces, see ccsx_launch_all; the grid covers the slot capacity, a multiple of 8,
    // the map only the windows there are)
    const int nwin_here = polish_piece_windows(P.wstart[P.n_zmw], slot0, (int)gridDim.x);
#ifdef CCSX_EXP_HOT_PROLOGUE                                // experiment (timing only, wrong results): every workgroup takes one of the same 4096 windows, so the
    const int bid_ = slot0 + xcd_contiguous(blockIdx.x, nwin_here);   // prologue's loads come from the L2 — what its HBM latency costs the whole kernel
    const int bid = bid_ < slot0 ? bid_ : slot0 + ((bid_ - slot0) & 4095);
#else
    const int bid = slot0 + xcd_contiguous(blockIdx.x, nwin_here);
#endif
    if (bid < slot0) return;
    const int z = P.wslot_zmw[bid];                         // device-built compact map (k_wmap): no dependent search
    const int wbo = P.wb_off[z], nw = P.nwin[z], Ld = P.draft_len[z];
    const int r0 = P.read_off[z], nreads = P.nreads_used[z];
    const int64_t so = P.seq_off[z];
    const int w = bid - P.wstart[z];
    const int32_t *wb = P.wbounds + wbo;
    const uint8_t *draft = P.draft + so;
    const int wb0 = wb[w], wb1 = wb[w + 1];
    const int64_t bo_r0 = P.base_off[r0];
    const int fl0 = P.flags[r0 + (P.zref[z] & 255)] & 1;    // orientation of the draft = that of its backbone pass
    int ws = wb0 - CCSX_WIN_OVERHANG; if (ws < 0) ws = 0;
    int we = wb1 + CCSX_WIN_OVERHANG; if (we > Ld) we = Ld;
    const int idx_ws = (w == 0) ? 0 : 2 * w - 1, idx_we = (w == nw - 1) ? 2 * nw - 1 : 2 * (w + 1);
    const int lfv = draft[ws > 0 ? ws - 1 : 0], rfv = draft[we < Ld ? we : Ld - 1];
    const int lf = ws > 0 ? lfv : 4, rf = we < Ld ? rfv : 4;
    // the (at most three) alignment intervals this window spans: start columns relative to ws and lengths
    const int c1 = need_col(wb, nw, Ld, idx_ws + 1), c2 = (idx_ws + 2 <= idx_we) ? need_col(wb, nw, Ld, idx_ws + 2) : we;
    const int maxins = P.opts.max_insertion_size == 0 ? 30 : P.opts.max_insertion_size;
    {
        // level 2: everything that needs only z / r0 / the window bounds
        const int e0 = tid < CCSX_NOBS * 16 ? tid : 0;      // (observation e0 >> 4, context e0 & 15)
        const size_t tz = (size_t)z * 192;
        const int i0 = (e0 & 15) * CCSX_NOBS + (e0 >> 4);
        const float me0 = P.tabME[tz + i0], in0 = P.tabINS[tz + i0];
        const float dl = P.tabDL[(size_t)z * 16 + (tid & 15)];
        const float zp = P.tabZ[(size_t)z * 32 + (tid & 31)];
        const int tcl = tid < we - ws ? tid : we - ws - 1;
        const uint8_t dr = draft[ws + tcl];
        if (tid < CCSX_NOBS * 16) sCTX[(e0 >> 4) * CTXS + (e0 & 15)] = make_float2(me0, in0);
        if (tid < CTXS) sCTX[CCSX_NOBS * CTXS + tid] = make_float2(0.0f, 0.0f);
        if (tid < CCSX_NOBS) sCTX[tid * CTXS + (CTXS - 1)] = make_float2(0.0f, 0.0f);     // (the zero entry of every row)
        if (tid < TAB_TAIL) sZP[32 + tid] = 0.0f;            // (the padding behind sZP: part of the guard behind sTC)
        if (tid < 16) sDL[tid] = dl;
        if (tid < 32) sZP[tid] = zp;
        if (tid < we - ws) { sT[0][tid] = dr; sTd[tid] = dr; }
        if (tid == 0) { sCtl[0] = we - ws; sCtl[1] = wb0 - ws; sCtl[2] = wb1 - ws; }
    }
    // SPEC v5: a ZMW's passes (up to CCSX_MAX_PASSES = 255) are taken in GROUPS of PW_MAXREADS = 32: the per-read arrays below, the observation
    // codes and the chunk plan always describe one group (local read index = pass - g0).  A ZMW of at most 32 passes — most — loads
    // its one group here and never again; larger ones reload group after group in every round (rare, so the reload is not optimised).
    const int ngroups = (nreads + PW_MAXREADS - 1) / PW_MAXREADS;
    // levels 2 + 3 of the prologue for the reads g0 .. g0 + ng - 1: metadata, entry rows of the window's two edge columns, dirty masks
    auto load_meta = [&](int g0, int ng) -> int {
        int trimflag = 0;
        const int rcl = tid < ng ? tid : ng - 1;
        const int rr = r0 + g0 + rcl;
        const int64_t bo0 = P.base_off[rr], bo1 = P.base_off[rr + 1], eo = P.ent_off[rr];
        const int flr = P.flags[rr] & 1, av = P.avalid[rr];
        // level 3: entry rows of the window's two edge columns (in bounds for every read; ignored unless the read mapped)
        const int a = P.ent[eo + idx_ws], b = P.ent[eo + idx_we];
        const unsigned m1 = P.dmask[eo + idx_ws + 1];
        const unsigned m2 = (idx_ws + 2 <= idx_we) ? P.dmask[eo + idx_ws + 2] : 0u;
        const unsigned m3 = (idx_ws + 3 <= idx_we) ? P.dmask[eo + idx_ws + 3] : 0u;
        if (tid < ng) {
            const int L = (int)(bo1 - bo0);
            const int st = flr != fl0 ? 1 : 0;
            int n = -1, na = 0;
            if (av) {
                n = b - a;
                if (n > L) n = -1;                           // (entry rows are rows of this pass: anything else is not a segment)
                // SPEC "trim large insertions": a segment more than max_insertion_size bases longer than the window is cut down
                // to the window's length below (sBoff keeps the full length until the chunk plan re-uses it)
                if (n >= 0 && maxins > 0 && n > (we - ws) + maxins) { trimflag = 1; sBoff[tid] = n; n = we - ws; }
                else { sBoff[tid] = 0; if (n < 0 || n > CCSX_IMAX) n = -1; }
                na = st ? L - b : a;
            } else sBoff[tid] = 0;
            sI[tid] = n; sStrand[tid] = (uint8_t)st;
            sGoff[tid] = (int)(bo0 - bo_r0) + na;            // segment start relative to the ZMW's first base (sGoff is re-planned later)
            // interval k covers draft positions col(k-1) .. col(k)-1: bit (p - col(k-1))
            sDirty[tid] = av ? (m1 | (m2 << (c1 - ws)) | (m3 << (c2 - ws))) : 0u;
        }
        return trimflag;
    };
    // (the LAST group first: the candidate filter below walks the groups from the last to the first, so every group is loaded once and group 0 is the one that is
    // resident when the rounds begin — ADVICE r04: a ZMW of 33-64 passes used to load group 0 twice here and once more in round 0)
    const int glast = (ngroups > 1 ? ngroups - 1 : 0) * PW_MAXREADS;
    const int ng0 = nreads - glast < PW_MAXREADS ? nreads - glast : PW_MAXREADS;
    const int trimflag = load_meta(glast, ng0);
    int resident = glast;                                   // the group whose per-read arrays and observation codes are in LDS (wave-uniform)
    if (tid < (CCSX_MAX_PASSES + 32) / 32) sZdrop[tid] = 0u;
    // level 4 for the group's ng reads: the read segments (native orientation), four reads per wave in flight; then the rare trim
    auto load_obs = [&](int ng, int tflag) {
    const int anytrim = __syncthreads_or(tflag);
    for (int rb = 0; rb < ng; rb += 4 * (PWT / 64)) {
        uint8_t bq[4], pq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = rb + (PWT / 64) * q + wave;
            const int n = r < ng ? sI[r] : -1;
            const int64_t p = bo_r0 + ((lane < n) ? sGoff[r] + lane : 0);
            bq[q] = P.bases[p]; pq[q] = P.pw[p];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = rb + (PWT / 64) * q + wave;
            if (r < ng) {
                const int n = sI[r];
                sObs[r][lane] = (lane < n) ? (uint8_t)obs_of(bq[q], pq[q]) : (uint8_t)(lane == n ? CCSX_NOBS : 0);   // row n: "no base"
                if (lane < 4) sObs[r][64 + lane] = 0;
            }
        }
    }
    if (anytrim) {
        // rare: cut over-long segments down to J bases = their first s and last J - s bases, s = the split with the most diagonal
        // matches of prefix and suffix against the window in read orientation (ties: the smallest s).  The main loop above has
        // already stored the first J codes; the same wave owns the read here.
        const int J0 = we - ws;
        for (int r = wave; r < ng; r += (PWT / 64)) {
            const int nfull = rfl(sBoff[r]);
            if (nfull == 0) continue;
            const int st = rfl((int)sStrand[r]);
            const int64_t p0 = bo_r0 + sGoff[r];
            const bool in = lane < J0;
            const int bp = in ? P.bases[p0 + lane] : 0;
            const int bs = in ? P.bases[p0 + nfull - J0 + lane] : 0, ps = in ? P.pw[p0 + nfull - J0 + lane] : 0;
            const int T = in ? (st ? 3 - (int)sTd[J0 - 1 - lane] : (int)sTd[lane]) : 9;
            const unsigned long long mp = __ballot(in && (bp & 3) == T), ms = __ballot(in && (bs & 3) == T);
            const int tot = (lane <= J0) ? __popcll(mp & ((1ull << lane) - 1ull)) + __popcll(ms >> lane) : -1;
            const int sb = 63 - (rfl(wave_max_i32((tot << 6) | (63 - lane))) & 63);
            if (in && lane >= sb) sObs[r][lane] = (uint8_t)obs_of(bs, ps);
        }
    }
    };
    load_obs(ng0, trimflag);
    // ---- step 7, candidate filter: pile-up margin of every window position over the reads with a usable segment (of every group: a ZMW
    // of more than 32 passes (PW_MAXREADS) walks its groups from the last to the first here, so that group 0 is the one loaded when the rounds begin)
    {
        int nuse = 0, nd = 0;                                 // (wave 0)
        for (int g0 = (ngroups - 1) * PW_MAXREADS; g0 >= 0; g0 -= PW_MAXREADS) {
            const int ng = nreads - g0 < PW_MAXREADS ? nreads - g0 : PW_MAXREADS;
            if (g0 != resident) {
                __syncthreads();
                const int tf = load_meta(g0, ng);
                load_obs(ng, tf);
                __syncthreads();
                resident = g0;
            }
            if (wave == 0) {
                const int vIr = lane < ng ? sI[lane] : -1;                      // lane = read: one load each, then v_readlane per read
                const unsigned vDr = lane < ng ? sDirty[lane] : 0u;
                for (int r = 0; r < ng; ++r) if (rl(vIr, r) >= 0) { ++nuse; nd += (int)(((unsigned)rl((int)vDr, r) >> (lane & 31)) & 1u); }
            }
        }
        if (wave == 0) {
            const int J0 = we - ws;
            const int margin = nuse - 2 * nd;
            const bool inw = lane < J0;
            const unsigned neg = (unsigned)__ballot(inw && margin < 0);
            // positions within SKIP_SPREAD of a dirty majority are polished too
            const unsigned nearneg = neg | (neg << 1) | (neg << 2) | (neg << 3) | (neg >> 1) | (neg >> 2) | (neg >> 3);
            const bool ok = inw && !P.opts.disable_heuristics && margin >= SKIP_MARGIN && !((nearneg >> lane) & 1u);
            const unsigned ev = (unsigned)__ballot(ok);
            if (lane == 0) sCtl[7] = (int)ev;
            if (lane < 36) sPskip[lane] = inw ? skip_perr(margin, P.perr_floor) : 0.0f;
        }
    }
    const int half = lane >> 5, hrow = lane & 31;          // fill: which read of the pair, row within it
    PHASE(0);
#ifdef CCSX_EXIT_AFTER_PROLOGUE                             // experiment: what the prologue alone costs in kernel time
    __syncthreads();
    if (tid == 0) P.wsum[(size_t)(P.wb_off[z] - z) + w] = (float)sI[0] + (float)sObs[0][0] + sPskip[0] + sCTX[5].x;
    return;
#endif
#ifdef CCSX_EXP_REPEAT                                      // experiment: the compute of a window CCSX_EXP_REPEAT times after ONE prologue
    __syncthreads();                                        // (T(R=2) - T(R=1) = a window's compute without the memory latency at its head)
    const int xr_t = sT[0][tid & 31], xr_c0 = sCtl[0], xr_c1 = sCtl[1], xr_c2 = sCtl[2], xr_c7 = sCtl[7];
    const float xr_p = sPskip[tid < 36 ? tid : 0];
    for (int xrep = 0; xrep < CCSX_EXP_REPEAT; ++xrep) {
    __syncthreads();
    if (tid < 32) sT[0][tid] = (uint8_t)xr_t;
    if (tid < 36) sPskip[tid] = xr_p;
    if (tid < (CCSX_MAX_PASSES + 32) / 32) sZdrop[tid] = 0u;
    if (tid == 0) { sCtl[0] = xr_c0; sCtl[1] = xr_c1; sCtl[2] = xr_c2; sCtl[7] = xr_c7; }
    __syncthreads();
#endif
    int iters = 0, nonconv = 0;
    unsigned skmask = 0;                                     // positions skipped in the current round (wave-uniform)
    int nvalid_last = 0, nvfull_last = 0;                    // usable reads of the last round: all / full-length passes only
    const int nfull = P.nfull[z];
    for (int it = 0; it < CCSX_MAX_ITER; ++it) {
        // (round 4 measured the round without three of its barriers — this one in later rounds, the one after sT[1], the second one of the lane compaction:
        // 155.6 -> 156.1 ms, no gain; the barriers stay where they make the hazards obvious)
        __syncthreads();
        const int J = rfl(sCtl[0]);
        const float invJ2 = 1.0f / (float)(2 * J);           // band_row0's reciprocal
        const int S = (J + 2) & ~1;                          // EVEN row stride >= J+1: in the fill lane = row writes column t - row, so the
                                                             // lane-to-lane address stride is S - 1, which must be odd to be bank-conflict free
        if (tid < J) sT[1][tid] = (uint8_t)(3 - sT[0][J - 1 - tid]);
        __syncthreads();
        auto tbase = [&](int sd, int j) -> int { return (int)sT[sd][j]; };   // base j of the window template on strand sd
        const int lfr = (rf < 4) ? 3 - rf : 4;
        // z-score expectation of the window template on each strand, summed in column order (SPEC); the gate is decided in round 0
        // only.  Lane j fetches column j's terms, the ordered sum takes them with v_readlane (no chain of dependent LDS loads).
        if (it == 0 && wave < 2 && P.opts.min_zscore != 0.0f) {
            const int j = lane < J ? lane : 0;
            const int prev = j > 0 ? tbase(wave, j - 1) : (wave ? lfr : lf);
            const int k = ctx_of(prev, tbase(wave, j));
            const float mu = sZP[k], va = sZP[16 + k];
            float M = 0.0f, V = 0.0f;
            for (int q = 0; q < J; ++q) {
                M = M + __int_as_float(rl(__float_as_int(mu), q));
                V = V + __int_as_float(rl(__float_as_int(va), q));
            }
            if (lane == 0) { sZS[2 * wave] = M; sZS[2 * wave + 1] = V; }
        }
        // positions skipped this round: evidence bit set and not inside a homopolymer of the CURRENT template
        {
            const uint8_t *t = sT[0];
            const int c = lane & 31;
            const int tc = c < J ? t[c] : 9;
            const int prev = c > 0 ? (c - 1 < J ? t[c - 1] : 8) : lf, next = c + 1 < J ? t[c + 1] : rf;
            const bool hp = (prev == tc) || (next == tc);
            skmask = (unsigned)__ballot(lane < J && (((unsigned)sCtl[7] >> c) & 1u) && !hp);
        }
        // valid mutation lanes of this round, compacted: lane m = slot*32 + c is valid iff the SPEC enumerates it;
        // thread t then scores the t-th valid lane, so unused lanes cost nothing
        {
            const uint8_t *t = sT[0];
            const int sl0 = tid >> 5, c0 = tid & 31;
            int v0;
            if (sl0 < 3) v0 = c0 < J;
            else if (sl0 == 3) v0 = c0 < J && !(c0 > 0 && t[c0 - 1] == t[c0]);
            else if (sl0 < 8) v0 = c0 <= J && !(c0 > 0 && t[c0 - 1] == sl0 - 4);
            else v0 = 0;                                                      // threads 256.. own no mutation lane
            {                                                                 // candidate filter (insertions after the last column follow J-1)
                const int cc = c0 < J ? c0 : J - 1;
                if (v0 && ((skmask >> cc) & 1u)) v0 = 0;
                // a quiet position inside a homopolymer keeps only the mutations that change the run's LENGTH: deletion of the
                // run's first base, insertion of the run's base before it (the enumeration admits both at run starts only)
                else if (v0 && (((unsigned)sCtl[7] >> cc) & 1u) && !(c0 < J && (sl0 == 3 || (sl0 >= 4 && sl0 - 4 == t[c0])))) v0 = 0;
            }
            const unsigned long long bal = __ballot(v0);
            if (lane == 0) sCnt[wave] = __popcll(bal);
            if (tid < 256) { sMvalid[tid] = (uint8_t)v0; sDeltaI[tid] = 0; }
            __syncthreads();
            int basew = 0;
            for (int q = 0; q < wave; ++q) basew += sCnt[q];
            if (v0) sList[basew + __popcll(bal & ((1ull << lane) - 1ull))] = (uint8_t)tid;
            __syncthreads();
        }
        // the fill's per-column tables of this round's template (sT[1] is complete: barriers since).  sTC[strand][obs][column]: the pair of the column's context in
        // the observation's row of sCTX — zeros from column J on ("no stay in the final column") and for observation 12 ("no base": the zero row);
        // sDLC[strand][column]: DL of the column's context, 1 outside 0 .. J-1 (beta's start value passes through column J as 1 * beta).  Boundary cells are data,
        // not control: the sweeps are branch-free.
        for (int e = tid; e < 2 * TC_STRAND; e += PWT) {
            const int sd = e >= TC_STRAND ? 1 : 0, e2 = e - sd * TC_STRAND, o = e2 >> 5, j = e2 & 31;
            float2 v = make_float2(0.0f, 0.0f);
            if (j < J) {
                const int prev = j > 0 ? tbase(sd, j - 1) : (sd ? lfr : lf);
                v = sCTX[o * CTXS + ctx_of(prev, tbase(sd, j))];
            }
            sTC[e] = v;
        }
        for (int e = tid; e < DLC_TOTAL; e += PWT) {
            const int sd = e >= TC_LO + 33 ? 1 : 0, j = e - TC_LO - sd * DLC_S;
            float v = 1.0f;
            if (j >= 0 && j < J) {
                const int prev = j > 0 ? tbase(sd, j - 1) : (sd ? lfr : lf);
                v = sDL[ctx_of(prev, tbase(sd, j))];
            }
            // (the entry right in front of sTC is column -1 of strand 0's observation row 0: row 1's lane reads it at step 0 while row 0's start value 1 is still in its
            // neighbour, so its ME has to be an exact zero like column 31 of the row before every other observation row — no lane uses a DL that far out)
            sDLC[e] = e >= DLC_TOTAL - 2 ? 0.0f : v;
        }
        int nvm = 0;
#pragma unroll
        for (int q = 0; q < (PWT / 64); ++q) nvm += rfl(sCnt[q]);
        const int nblk = (nvm + 63) >> 6;
        PHASE(1);
        int nvalid = 0, nvfull = 0;
        int curblk = -1;                                     // the block whose lane constants this wave holds (they survive the chunks of a round)
        LaneMut LF, LR;
        int myM = 0; bool mval = false;
        // ---- groups of PW_MAXREADS passes (one group for nearly every ZMW), and in a group: chunks of reads whose gamma/beta fit the LDS budget
        for (int g0 = 0; g0 < nreads; g0 += PW_MAXREADS) {
        const int ng = nreads - g0 < PW_MAXREADS ? nreads - g0 : PW_MAXREADS;
        if (g0 != resident) {                                // (the per-read arrays and observation codes of the group; one group only: never reloaded)
            __syncthreads();
            const int tf = load_meta(g0, ng);
            load_obs(ng, tf);
            resident = g0;
        }
        int rbeg = 0;
        while (rbeg < ng) {
            __syncthreads();
            int rend, nlong, nshort;
            {   // lane = read: the greedy plan by prefix sum and ballots.  EVERY wave computes it (identical values, benign identical
                // LDS writes): a wave then reads only what it wrote itself, so no barrier is needed before the fill
                const int r = lane;
                const int n = (r >= rbeg && r < ng) ? sI[r] : -1;
                const bool cand = n >= 0;
                const FillBand fb = fill_band_of(cand ? n : 0, J, S);
                const int need = cand ? (2 * n + 3) * fb.rowsz : 0;          // gamma rows 0..n, beta rows 0..n+1 (row n+1: zeros)
                const int incl = wave_scan_add_i32(need);
                const unsigned long long bcand = __ballot(cand);
                const unsigned long long over = __ballot(cand && (incl > GB_FLOATS || (PWCH > 0 && __popcll(bcand & ((1ull << lane) - 1ull)) >= PWCH)));
                const int rend_ = over ? (int)__ffsll((long long)over) - 1 : ng;             // the first read that does not fit any more
                if (r >= rbeg && r < rend_) {
                    if (!cand) { sGoff[r] = -1; sValid[r] = 0; }
                    else {
                        const int off = incl - need;
                        sGoff[r] = off + fb.org; sBoff[r] = off + (n + 1) * fb.rowsz + fb.org;
                        sBand[r] = fb.pitch | (fb.rowsz << 8) | (fb.bw << 16) | ((fb.dlo + 128) << 24);
                    }
                }
                const bool inchunk = cand && r < rend_;
                // a short segment whose band leaves a lane time to change rows (see the quad sweep) shares a wave with three others; the rest — more than
                // 31 bases, or |I - J| >= 5 — takes a wave of its own (rare)
                const bool qd = inchunk && n <= 31 && fb.bw <= FILL16_MAXBW, lng = inchunk && !qd;
                const unsigned long long bl = __ballot(lng), bs = __ballot(qd);
                const unsigned long long lower = (1ull << lane) - 1ull;
                const int nl_ = __popcll(bl);
                if (lng) sTask[__popcll(bl & lower)] = make_short2((short)r, (short)-1);
                if (qd) sTask[nl_ + __popcll(bs & lower)] = make_short2((short)r, (short)0);
                rend = rfl(rend_); nlong = rfl(nl_); nshort = rfl(__popcll(bs));
            }
            PHASE(2);
            // ---- A1/A2: fill (SPEC v6: on the band of diagonals only).  Work units, one wave each, every unit ONE anti-diagonal sweep:
            //   quad units — FOUR short reads per wave, one per 16-lane DPP row, alpha and beta of a quad on different waves;
            //   long units — one read per wave (lane = row), alpha and beta on different waves.
            // alpha(I,J) / beta(0,0) of a read meet in LDS, and every wave derives the reads' validity from them after the barrier (lane = read; identical values
            // in every wave: no second barrier).  Eight short reads = two quads = four units: every wave runs one sweep per chunk.
#ifdef CCSX_EXP_NO_FILL                                     // experiment (timing only, wrong results)
            const int nquad = 0, nunit_f = 0, rpq = 4;
#else
            // (reads per quad unit: always four.  Spreading a small chunk — three passes, the tail chunk of ten — over all the waves, two reads or one per unit, was
            // measured and is SLOWER: k_polish 313.9 against 301.6 ms at 10 passes, 95.8 against 92.4 at 3 — the kernel is bound by its instruction count, and a
            // sweep costs the same instructions whatever it holds)
            const int rpq = 4;
            const int nquad = (nshort + rpq - 1) / rpq, nunit_f = 2 * nquad + 2 * nlong;
#endif
            for (int fu = wave; fu < nunit_f; fu += (PWT / 64)) {
              if (fu < 2 * nquad) {
                // ---- quad unit.  Lane l of a DPP row owns read rows l and l + 16 (a short segment has at most 32 rows): row i is on the band for the 15-or-so
                // steps t = i + j in [2 i + dlo, 2 i + dhi], row i + 16 exactly 32 steps later, so with a band of at most FILL16_MAXBW = 24 diagonals the lane is
                // idle for at least 8 steps in between and changes rows there.  The neighbour row's cell comes by a ROTATION inside the DPP row (row_ror: lane 0
                // takes lane 15, whose row 15 precedes lane 0's row 16; while lane 0 is on row 0, lane 15 has not started and holds a zero).  A lane that is not on
                // the band holds a zero running cell, which is what its neighbours must read past the band's edge.  Look-ups run four steps ahead: slot k holds DL and the
                // (ME, INS) pair of the step's column — read from the per-column tables sDLC / sTC at a running pointer + immediate — and is refilled right after use.  The row
                // change therefore happens in two parts: the look-up side (observation row of sTC, 16 columns back) at the first iteration whose look-ups no longer
                // serve the old row, the compute side (activity window, store pointer) one iteration later.
                const int quad = fu >> 1, isb = fu & 1;
                const int g4 = lane >> 4, l16 = lane & 15;
                const int qi = rpq * quad + g4;
                const bool have = g4 < rpq && qi < nshort;
                const int myr = sTask[nlong + (have ? qi : rpq * quad)].x;
                const int I = sI[myr], sd = sStrand[myr], band = sBand[myr];
                const int pitch = band & 255, rowsz = (band >> 8) & 255, bdlo = (int)((unsigned)band >> 24) - 128, bdhi = bdlo + ((band >> 16) & 255) - 1;
                const int row0 = l16, row1 = l16 + 16;
                const bool ok0 = have && row0 <= I, ok1 = have && row1 <= I;
                const int jlo0 = (row0 + bdlo > 0) ? row0 + bdlo : 0, jhi0 = (row0 + bdhi < J) ? row0 + bdhi : J;
                const int jlo1 = (row1 + bdlo > 0) ? row1 + bdlo : 0, jhi1 = (row1 + bdhi < J) ? row1 + bdhi : J;
                int Tmax = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int iq = rfl(sI[rfl((int)sTask[nlong + ((q < rpq && rpq * quad + q < nshort) ? rpq * quad + q : rpq * quad)].x)]); Tmax = iq > Tmax ? iq : Tmax; }
                Tmax += J;
                const int NEVER = 1 << 20;
#define LDPR(ROWP, OFF) (*(const float2 *)((ROWP) + (OFF)))
                if (!isb) {
                    // alpha: row 0 first.  Step t: the lane's row i computes column j = t - i
                    const int obA0 = OBS_TC((row0 >= 1 && ok0) ? (int)sObs[myr][row0 - 1] : 12);       // o_{i-1}; 12 = no base: row 0 has no diagonal / stay
                    const int obA1 = OBS_TC(ok1 ? (int)sObs[myr][row1 - 1] : 12);
                    // tA[x] (bytes: 8 x) = the pair of column x - row in the lane's observation row of sTC; dA[x] = DL of column x - row - 1.  Both run with t; the
                    // look-up side changes rows (observation row, 16 columns back) one iteration before the compute side
                    const char *tA = (const char *)(sTC + sd * TC_STRAND) + obA0 - 8 * row0;
                    const int dtA = obA1 - obA0 - 8 * 16;
                    const float *dA = sDLC + sd * DLC_S + (TC_LO - 1 - row0);
                    const int tA0 = ok0 ? row0 + jlo0 : NEVER, tA1 = row1 + jlo1;
                    unsigned uJ = ok0 ? (unsigned)(jhi0 - jlo0) : 0u;
                    const unsigned uJ1 = (unsigned)(jhi1 - jlo1);
                    const int tsw = ok1 ? ((row0 + jhi0) & ~3) : NEVER;          // = the first multiple of 4 >= (last step of row i) - 3
                    float *gA = sGB + sGoff[myr] + row0 * pitch - row0;               // gA[x] = gamma(row, x - row)
                    const int dgA = 16 * (pitch - 1), dcnt = tA0 - tA1;
                    float acur = (ok0 && row0 == 0) ? 1.0f : 0.0f, mnext = 0.0f;     // mnext = alpha(i-1, j-1) * ME(j-1) of the next step, formed a step ahead
#define CCSX_A_INIT(K) float dl##K = dA[K]; float2 p##K = LDPR(tA, 8 * (K));
                    CCSX_A_INIT(0) CCSX_A_INIT(1) CCSX_A_INIT(2) CCSX_A_INIT(3)
#undef CCSX_A_INIT
                    int cnt = -tA0;
#define CCSX_A_STEP(K)                                                                                                     \
                    {                                                                                                      \
                        const float up = row_ror1_f32(acur);         /* alpha(i-1, j) */                                     \
                        const float gmm = __builtin_fmaf(acur, dl##K, mnext);   /* SPEC v8: fused */                      \
                        const unsigned long long on = __builtin_amdgcn_uicmp((unsigned)(cnt + (K)), uJ, 37 /* ule */);    \
                        if (__builtin_amdgcn_inverse_ballot_w64(on)) gA[(K)] = gmm;                                        \
                        acur = lanes_or_zero(on, __builtin_fmaf(up, p##K.y, gmm));   /* row 0 and column J read zero entries: + 0.  (ONE compare: the mask serves the store's exec and the select) */ \
                        mnext = up * p##K.x;                                                                               \
                        p##K = LDPR(tA, 8 * ((K) + 4));              /* the slot's next use: four steps on */              \
                        dl##K = dA[(K) + 4];                                                                               \
                    }
                    for (int t = 0; t <= Tmax; t += 4, tA += 32, dA += 4, gA += 4, cnt += 4) {
                        if (t == tsw) { tA += dtA; dA -= 16; }
                        if (t - 4 == tsw) { uJ = uJ1; cnt += dcnt; gA += dgA; }
                        CCSX_A_STEP(0) CCSX_A_STEP(1) CCSX_A_STEP(2) CCSX_A_STEP(3)
                    }
#undef CCSX_A_STEP
                    // (a lane's running cell is zero once its row has left the band: alpha(I,J) = gamma(I,J) — no stay in the final column — is read back from LDS)
                    if (have && l16 == 0) sBase[myr] = sGB[sGoff[myr] + I * pitch + J];
                } else {
                    // beta: row I first.  Step t: the lane's row i computes column j = J - (t - (I - i)); a lane with two rows starts on row l + 16
                    const int rowF = ok1 ? row1 : row0, jloF = ok1 ? jlo1 : jlo0, jhiF = ok1 ? jhi1 : jhi0;
                    const int obB0 = OBS_TC((ok0 && row0 < I) ? (int)sObs[myr][row0] : 12);             // o_i; 12 = no base: row I emits nothing more
                    const int obBF = ok1 ? OBS_TC(row1 < I ? (int)sObs[myr][row1] : 12) : obB0;
                    // tB[-x] (bytes: -8 x) = the pair of column J + I - row - x in the lane's observation row of sTC; dB[-x] = DL of that column
                    const char *tB = (const char *)(sTC + sd * TC_STRAND) + obBF + 8 * (J + I - rowF);
                    const int dtB = obB0 - obBF + 8 * 16;
                    const float *dB = sDLC + sd * DLC_S + (TC_LO + J + I - rowF);
                    const int tB0 = ok0 ? I - rowF + J - jhiF : NEVER, tB1 = I - row0 + J - jhi0;
                    unsigned uJ = ok0 ? (unsigned)(jhiF - jloF) : 0u;
                    const unsigned uJ1 = (unsigned)(jhi0 - jlo0);
                    const int tsw = ok1 ? ((I - row1 + J - jlo1) & ~3) : NEVER;
                    float *bE = sGB + sBoff[myr] + rowF * pitch + (J + I - rowF);     // bE[-x] = beta(row, J + I - row - x)
                    const int dbE = -16 * (pitch - 1), dcnt = tB0 - tB1;
                    float bcur = (ok0 && rowF == I) ? 1.0f : 0.0f, t1next = 0.0f;    // t1next = ME(j) * beta(i+1, j+1) of the next step
#define CCSX_B_INIT(K) float dk##K = dB[-(K)]; float2 q##K = LDPR(tB, -8 * (K));
                    CCSX_B_INIT(0) CCSX_B_INIT(1) CCSX_B_INIT(2) CCSX_B_INIT(3)
#undef CCSX_B_INIT
                    int cnt = -tB0;
#define CCSX_B_STEP(K, KN)                                                                                                 \
                    {                                                                                                      \
                        const float dn = row_rol1_f32(bcur);         /* beta(i+1, j) */                                    \
                        const float bv = __builtin_fmaf(dk##K, bcur, __builtin_fmaf(q##K.y, dn, t1next));   /* SPEC v8: fused */ \
                        const unsigned long long on = __builtin_amdgcn_uicmp((unsigned)(cnt + (K)), uJ, 37 /* ule */);    \
                        if (__builtin_amdgcn_inverse_ballot_w64(on)) bE[-(K)] = bv;                                        \
                        bcur = lanes_or_zero(on, bv);                                                                      \
                        t1next = q##KN.x * dn;                       /* (slot KN holds the pair of the next step's column) */ \
                        q##K = LDPR(tB, -8 * ((K) + 4));                                                                   \
                        dk##K = dB[-(K) - 4];                                                                              \
                    }
                    for (int t = 0; t <= Tmax; t += 4, tB -= 32, dB -= 4, bE -= 4, cnt += 4) {
                        if (t == tsw) { tB += dtB; dB += 16; }
                        if (t - 4 == tsw) { uJ = uJ1; cnt += dcnt; bE += dbE; }
                        CCSX_B_STEP(0, 1) CCSX_B_STEP(1, 2) CCSX_B_STEP(2, 3) CCSX_B_STEP(3, 0)
                    }
#undef CCSX_B_STEP
                    // zero row I+1 of beta; beta(0,0)
                    const int org_ = pitch == S ? 0 : -bdlo;
                    if (have) for (int x = l16; x < rowsz; x += 16) sGB[sBoff[myr] - org_ + (I + 1) * rowsz + x] = 0.0f;
                    if (have && l16 == 0) sB00[myr] = sGB[sBoff[myr]];
                }
#undef LDPR
              } else {
                // ---- long unit: one read per wave, lane = row, the plain loop (alpha-only or beta-only)
                const int tk = (fu - 2 * nquad) >> 1, mode = 1 + ((fu - 2 * nquad) & 1);      // 1: alpha only, 2: beta only
                const int myr = rfl((int)sTask[tk].x);
                const int row = lane;
                const int I = rfl(sI[myr]);
                const int Tmax = I + J;
                const int sd = sStrand[myr];
                const float *dlcol = sDLC + sd * DLC_S + TC_LO;      // DL by column (1 at column J)
                const bool rowok = row <= I;
                // SPEC v6: the columns of this lane's row that lie on the read's band, jlo .. jhi; the cells outside are zeros and are neither computed nor stored
                const int band = sBand[myr];
                const int pitch = band & 255, rowsz = (band >> 8) & 255, bdlo = (int)((unsigned)band >> 24) - 128, bdhi = bdlo + ((band >> 16) & 255) - 1;
                const int jlo = (row + bdlo > 0) ? row + bdlo : 0, jhi = (row + bdhi < J) ? row + bdhi : J;
                // the lane's observation rows of the per-column table sTC
                const int op = OBS_TC((row >= 1 && rowok) ? (int)sObs[myr][row - 1] : 12);     // o_{i-1}; 12 = no base: row 0 has no diagonal / stay
                const int oc = OBS_TC((row < I) ? (int)sObs[myr][row] : 12);                    // o_i;     12 = no base: row I emits nothing more
                const float2 *rowA = (const float2 *)((const char *)(sTC + sd * TC_STRAND) + op), *rowB = (const float2 *)((const char *)(sTC + sd * TC_STRAND) + oc);
                // activity windows: alpha computes column j = t - row for t in [row + jlo, row + jhi]; beta computes column
                // jb = J - (t - (I - row)) for t in [I - row + J - jhi, I - row + J - jlo]
                const int tA0 = rowok ? row + jlo : (1 << 20), tB0 = rowok ? I - row + J - jhi : (1 << 20);
                const float one0 = (row == 0) ? 1.0f : 0.0f, oneI = (row == I) ? 1.0f : 0.0f;
                float *gA = sGB + sGoff[myr] + row * pitch - row;
                float *bB = sGB + sBoff[myr] + row * pitch + (J + I - row - 1);
                // start values chosen so that the general recurrence yields the boundary cells: gamma(i,0) = 0*x + one0*1,
                // beta(i,J) = (0 + 0) + 1*oneI (column J of the tables is zero with DL = 1)
                float acur = one0, updiag = 0.0f, mePrev = 0.0f, dlPrev = 1.0f;
                float bcur = oneI, dndiag = 0.0f;
                // a lane whose step is not on the band holds a ZERO running cell (SPEC v6: its neighbours read the cell past the band's edge as zero); row 0 / row I
                // keep their start value until their first step, which is the sweep's first
                const unsigned uJ = rowok ? (unsigned)(jhi - jlo) : 0u;   // (a lane beyond the read's rows: tA0 / tB0 keep it off; its jlo .. jhi may be an empty, i.e. negative, range)
                const int tAc = rowok ? row : (1 << 20);             // (the loop carries ME / DL of the previous column from column 0 on)
                const unsigned uJc = (unsigned)J;
                if (mode == 1) {
                    for (int t = 0; t <= Tmax; t += 2, gA += 2) {
#define CCSX_LA_STEP(T, AOFF)                                                                                         \
                        {                                                                                                  \
                            const float up = wave_shr1_f32_z(acur);  /* all rows of the read shift together (full exec) */ \
                            float nv = 0.0f;                                                                               \
                            if ((unsigned)((T) - tAc) <= uJc) {      /* alpha, column j = T - row of the window */        \
                                const float2 pr = rowA[(T) - row];                                                         \
                                const float dlc = dlcol[(T) - row];                                                        \
                                const float gmm = __builtin_fmaf(acur, dlPrev, updiag * mePrev);                           \
                                if ((unsigned)((T) - tA0) <= uJ) { gA[(AOFF)] = gmm; nv = __builtin_fmaf(up, pr.y, gmm); }   /* ... on the band; row 0 and column J read zero entries: + 0 */ \
                                mePrev = pr.x; dlPrev = dlc;                                                               \
                            }                                                                                              \
                            acur = nv;                                                                                     \
                            updiag = up;                                                                                   \
                        }
                        CCSX_LA_STEP(t, 0)
                        CCSX_LA_STEP(t + 1, 1)
#undef CCSX_LA_STEP
                    }
                    if (lane == 0) sBase[myr] = sGB[sGoff[myr] + I * pitch + J];   // (alpha(I,J) = gamma(I,J): no stay in the final column)
                } else {
                    for (int t = 0; t <= Tmax; t += 2, bB -= 2) {
#define CCSX_LB_STEP(T, BOFF)                                                                                         \
                        {                                                                                                  \
                            const float dn = wave_shl1_f32_z(bcur);                                                        \
                            float nv = 0.0f;                                                                               \
                            if ((unsigned)((T) - tB0) <= uJ) {       /* beta, column jb = J - (T - (I - row)), on the band */ \
                                const int jb = J + I - row - (T);    /* the column the step computes */                   \
                                const float2 pr = rowB[jb];                                                                \
                                const float bv = __builtin_fmaf(dlcol[jb], bcur, __builtin_fmaf(pr.y, dn, pr.x * dndiag)); \
                                bB[(BOFF)] = bv;                                                                           \
                                nv = bv;                                                                                   \
                            }                                                                                              \
                            bcur = nv;                                                                                     \
                            dndiag = dn;                                                                                   \
                        }
                        CCSX_LB_STEP(t, 1)
                        CCSX_LB_STEP(t + 1, 0)
#undef CCSX_LB_STEP
                    }
                    const int org_ = pitch == S ? 0 : -bdlo;      // (row layout: no origin shift)
                    if (row < rowsz) sGB[sBoff[myr] - org_ + (I + 1) * rowsz + row] = 0.0f;
                    if (lane == 0) sB00[myr] = sGB[sBoff[myr]];
                }
              }
            }
            __syncthreads();
            PHASE(3);
            // read validity (lane = read; every wave computes the same values): alpha/beta agreement, the z-score gate of round 0
            int vOk = 0; float vLa = 0.0f;
            if (lane >= rbeg && lane < rend && sGoff[lane] >= 0) {
                const float aIJ = sBase[lane], b00 = sB00[lane];
                const int I = sI[lane], sd = sStrand[lane];
                if (aIJ > TINY_P && b00 > TINY_P) {
#ifdef CCSX_EXP_CHEAP_VALIDITY                              // experiment (timing only, wrong results): what the two logarithms of the validity step cost
                    vLa = aIJ; const float lb = b00 - (b00 - aIJ);
#else
                    vLa = det_log2f(aIJ); const float lb = det_log2f(b00);
#endif
                    float df = vLa - lb; if (df < 0.0f) df = -df;
                    vOk = !(df > AB_TOL);
                    if ((sZdrop[g0 >> 5] >> lane) & 1u) vOk = 0;       // (g0 is a multiple of PW_MAXREADS = 32, lane < 32 here)
                    else if (vOk && it == 0 && P.opts.min_zscore != 0.0f) {   // A7 z-score gate, round 0 only (x4 per emitted base = 2 bits per read base)
                        const float zd = (vLa - (float)(2 * I)) - sZS[2 * sd];
                        const float zm = P.opts.min_zscore;
                        if (zd < 0.0f && zd * zd > (zm * zm) * sZS[2 * sd + 1]) { vOk = 0; atomicOr(&sZdrop[g0 >> 5], 1u << lane); }   // (every wave sets the same bit)
                    }
                }
            }
#ifdef CCSX_EXP_ALL_VALID                                   // experiment (timing only): every read with a segment counts as usable, whatever the fill produced
            vOk = (lane >= rbeg && lane < rend && sGoff[lane] >= 0) ? 1 : 0;
#endif
            // usable reads of the chunk, in read order: every wave builds the list in a register (lane k = the k-th usable read) with one
            // ds_permute (valid lane r sends r to lane rank(r), the others fill the remaining lanes): no LDS list, no second barrier
            int vRlist, nv_chunk;
            {
                const bool v = vOk != 0;
                const unsigned long long bv = __ballot(v), lower = (1ull << lane) - 1ull;
                nv_chunk = rfl(__popcll(bv));
                nvfull += __popcll(bv & (nfull - g0 >= 64 ? ~0ull : (nfull - g0 <= 0 ? 0ull : (1ull << (nfull - g0)) - 1ull)));   // (partial passes sit behind the full ones)
                const int dest = v ? __popcll(bv & lower) : nv_chunk + __popcll(~bv & lower);
                vRlist = __builtin_amdgcn_ds_permute(dest << 2, lane);
            }
            PHASE(7);
            // ---- A3/A4: work pool.  A unit = (block of 64 compacted mutation lanes) x (one usable read), numbered block-major; every
            // wave takes an equal contiguous share and walks it two reads at a time (two independent chains per lane) while the
            // block stays the same — 2 blocks x 5 reads are 3+3+2+2 units, not 2+2+1+1 pair tasks.  Gains are added to sDeltaI in
            // fixed point (order independent).
            {
#ifdef CCSX_EXP_SKIP_ROUND2_SCORE
                const int nv = nv_chunk, nunits = (it == 0) ? nblk * nv : 0;   // experiment: upper bound of what neighbourhood-only rescoring could save
#elif defined(CCSX_EXP_NO_SCORE)
                const int nv = nv_chunk, nunits = 0;
#else
                const int nv = nv_chunk, nunits = nblk * nv;
#endif
                nvalid += nv;
                // per-read scalars of the chunk's usable reads, one read per lane: a unit takes them with v_readlane (no chain of
                // dependent LDS loads at the start of every unit)
                int vR, vI, vSt, vG, vB, vBd; float vBase;
                {
                    vR = lane < nv ? vRlist : rl(vRlist, 0);
                    vI = sI[vR]; vSt = (int)sStrand[vR]; vG = sGoff[vR]; vB = sBoff[vR]; vBd = sBand[vR]; vBase = __shfl(vLa, vR);
                }
                const int u_end = ((wave + 1) * nunits) / (PWT / 64);
                for (int u = (wave * nunits) / (PWT / 64); u < u_end;) {
                    int blk = 0, k0 = u;                                     // u = blk * nv + k0 (a window has one or two blocks of lanes: no integer division)
                    while (k0 >= nv) { k0 -= nv; ++blk; }
                    const bool two = (u + 1 < u_end) && (k0 + 1 < nv);       // the next unit is mine and in the same block
                    u += two ? 2 : 1;
                    if (blk != curblk) {
                        curblk = blk;
                        const int li = blk * 64 + lane;
                        mval = li < nvm;
                        myM = mval ? sList[li] : 0;
                        const int slot = myM >> 5, cpos = myM & 31;
                        int type, x = 0;
                        const uint8_t *t = sT[0];
                        if (slot < 3) { type = 0; x = (t[cpos < J ? cpos : 0] + 1 + slot) & 3; }
                        else if (slot == 3) type = 1;
                        else { type = 2; x = slot - 4; }
                        if (mval) {
                            LF = lane_mut(type, cpos, x, sT[0], J, lf, sDL);
                            LR = lane_mut(type, (type == 2) ? J - cpos : J - 1 - cpos, 3 - x, sT[1], J, lfr, sDL);
                        } else { LF = lane_mut(0, 0, 0, sT[0], J, lf, sDL); LR = LF; }
                    }
                    int dq;
                    if (!two) {
                        // a single unit (the usual case with four reads per chunk and one block of lanes): one chain, and none of the
                        // second chain's set-up
                        const int ra = rl(vR, k0), Ia = rl(vI, k0);
                        const LaneMut La = rl(vSt, k0) ? LR : LF;
                        const int gA_ = rl(vG, k0), bA_ = rl(vB, k0), bdA = rl(vBd, k0);
                        const int pA = bdA & 255, dloA = (int)((unsigned)bdA >> 24) - 128; const unsigned bwA = (unsigned)(bdA >> 16) & 255u;
                        const int dA = Ia > J ? Ia - J : J - Ia, WrA = SCORE_BAND + (dA > 2 ? dA - 2 : 0);
#ifdef CCSX_EXP_NO_ROWS                                     // experiment (timing only): a unit without its rows
                        const int nrA = 0;
#else
                        const int nrA = (Ia < 2 * WrA ? Ia : 2 * WrA) + 1;
#endif
                        const int i0a = band_row0(La.c, Ia, 2 * J, invJ2, J, WrA, nrA);
                        lds_cc tAa = (lds_cc)(sCTX + La.kA), tBa = (lds_cc)(sCTX + La.kB);
                        ScoreChain ca;
                        ca.ap = ca.bp = ca.acc = ca.b = 0.0f; ca.pA = ca.pB = make_float2(0.f, 0.f);
                        ca.g = (lds_cf)(sGB + gA_ + __mul24(i0a, pA) + La.c); ca.be = (lds_cf)(sGB + bA_ + __mul24(i0a, pA) + La.q);
                        ca.dg = La.c - i0a - dloA;
                        ca.bq = *ca.be; if ((unsigned)(La.q - i0a - dloA) >= bwA) ca.bq = 0.0f;   // beta(i0, q): the first row's own cell, a zero off the band
                        ca.be += pA;
                        const unsigned limA = bwA - 1u - (unsigned)La.qoff;
                        ca.op = (lds_cu8)(&sObs[ra][0] + i0a);
                        asm volatile("" : "+v"(ca.g), "+v"(ca.be), "+v"(tAa), "+v"(tBa), "+v"(ca.op));
                        {   // two rows per iteration: the chain's carried values (previous table pairs, beta, a, b) then rotate between two
                            // register sets instead of being copied at every row (3 v_mov + a loop counter per row before)
                            int ia = 0;
                            for (; ia + 2 <= nrA; ia += 2) { score_step(ca, La, tAa, tBa, pA, limA); score_step(ca, La, tAa, tBa, pA, limA); }
                            if (ia < nrA) score_step(ca, La, tAa, tBa, pA, limA);
                        }
                        const float res = La.fin ? ca.b : ca.acc;
#ifdef CCSX_EXP_NO_SCORE_LOG                                // experiment (timing only): a unit without its logarithm and fixed-point conversion
                        dq = __float_as_int(res) >> 20;
#else
                        dq = dq_fix(det_log2f(res) - __int_as_float(rl(__float_as_int(vBase), k0)));
#endif
                    } else {
                    const int k1 = k0 + 1;
                    const int ra = rl(vR, k0), rb = rl(vR, k1);
                    const int Ia = rl(vI, k0), Ib = rl(vI, k1);
                    const LaneMut La = rl(vSt, k0) ? LR : LF;
                    const LaneMut Lb = rl(vSt, k1) ? LR : LF;
                    const int gA_ = rl(vG, k0), bA_ = rl(vB, k0), gB_ = rl(vG, k1), bB_ = rl(vB, k1), bdA = rl(vBd, k0), bdB = rl(vBd, k1);
                    const int pA = bdA & 255, dloA = (int)((unsigned)bdA >> 24) - 128; const unsigned bwA = (unsigned)(bdA >> 16) & 255u;
                    const int pB = bdB & 255, dloB = (int)((unsigned)bdB >> 24) - 128; const unsigned bwB = (unsigned)(bdB >> 16) & 255u;
                    // SPEC "banded link": every lane scores the rows around its column's point on the window diagonal only
                    const int dA = Ia > J ? Ia - J : J - Ia, WrA = SCORE_BAND + (dA > 2 ? dA - 2 : 0);
                    const int nrA = (Ia < 2 * WrA ? Ia : 2 * WrA) + 1;
                    const int dB = Ib > J ? Ib - J : J - Ib, WrB = SCORE_BAND + (dB > 2 ? dB - 2 : 0);
                    const int nrB = (Ib < 2 * WrB ? Ib : 2 * WrB) + 1;
                    const int i0a = band_row0(La.c, Ia, 2 * J, invJ2, J, WrA, nrA);
                    const int i0b = band_row0(Lb.c, Ib, 2 * J, invJ2, J, WrB, nrB);
                    lds_cc tAa = (lds_cc)(sCTX + La.kA), tBa = (lds_cc)(sCTX + La.kB);
                    lds_cc tAb = (lds_cc)(sCTX + Lb.kA), tBb = (lds_cc)(sCTX + Lb.kB);
                    ScoreChain ca, cb;
                    ca.ap = ca.bp = ca.acc = ca.b = 0.0f; ca.pA = ca.pB = make_float2(0.f, 0.f);
                    cb.ap = cb.bp = cb.acc = cb.b = 0.0f; cb.pA = cb.pB = make_float2(0.f, 0.f);
                    ca.g = (lds_cf)(sGB + gA_ + __mul24(i0a, pA) + La.c); ca.be = (lds_cf)(sGB + bA_ + __mul24(i0a, pA) + La.q);
                    cb.g = (lds_cf)(sGB + gB_ + __mul24(i0b, pB) + Lb.c); cb.be = (lds_cf)(sGB + bB_ + __mul24(i0b, pB) + Lb.q);
                    ca.dg = La.c - i0a - dloA; cb.dg = Lb.c - i0b - dloB;
                    ca.bq = *ca.be; if ((unsigned)(La.q - i0a - dloA) >= bwA) ca.bq = 0.0f;
                    cb.bq = *cb.be; if ((unsigned)(Lb.q - i0b - dloB) >= bwB) cb.bq = 0.0f;
                    ca.be += pA; cb.be += pB;
                    const unsigned limA = bwA - 1u - (unsigned)La.qoff, limB = bwB - 1u - (unsigned)Lb.qoff;
                    ca.op = (lds_cu8)(&sObs[ra][0] + i0a); cb.op = (lds_cu8)(&sObs[rb][0] + i0b);
                    // opaque to the optimiser from here: the running pointers hold complete LDS addresses (otherwise the
                    // dynamic-LDS base is re-added at every use)
                    asm volatile("" : "+v"(ca.g), "+v"(ca.be), "+v"(cb.g), "+v"(cb.be), "+v"(tAa), "+v"(tBa), "+v"(tAb), "+v"(tBb), "+v"(ca.op), "+v"(cb.op));
                    const int nmin = nrA < nrB ? nrA : nrB;
                    int i = 0;
                    for (; i + 2 <= nmin; i += 2) {                            // both chains, two rows per iteration (no register copies between rows)
                        score_step(ca, La, tAa, tBa, pA, limA);
                        score_step(cb, Lb, tAb, tBb, pB, limB);
                        score_step(ca, La, tAa, tBa, pA, limA);
                        score_step(cb, Lb, tAb, tBb, pB, limB);
                    }
                    for (int ia = i; ia < nrA; ++ia) score_step(ca, La, tAa, tBa, pA, limA);
                    for (int ib = i; ib < nrB; ++ib) score_step(cb, Lb, tAb, tBb, pB, limB);
                    {
                        const float res = La.fin ? ca.b : ca.acc;
                        dq = dq_fix(det_log2f(res) - __int_as_float(rl(__float_as_int(vBase), k0)));
                    }
                    {
                        const float res = Lb.fin ? cb.b : cb.acc;
                        dq += dq_fix(det_log2f(res) - __int_as_float(rl(__float_as_int(vBase), k1)));
                    }
                    }
                    if (mval) atomicAdd(&sDeltaI[myM], dq);
                }
            }
            rbeg = rend;
            PHASE(4);
        }
        }
        ++iters;
        nvalid_last = nvalid; nvfull_last = nvfull;
        __syncthreads();
        float delta = 0.0f;
        if (tid < 256) { delta = (float)sDeltaI[tid] * (1.0f / DQ_SCALE); sDelta[tid] = delta; }   // same slot, same thread
        const int fav = (tid < 256 && sMvalid[tid] && delta > MUT_EPS) ? 1 : 0;
        const int anyfav = __syncthreads_or(fav);
#ifdef CCSX_EXP_ONE_ROUND                                   // experiment (timing only): exactly one round per window whatever the gains say, so that variants which
        break;                                              // corrupt the gains (rows / logarithm compiled out) do not change the control flow they are compared under
#endif
        if (!anyfav || P.qv_only) break;                    // (CCSX_QV_ONLY: the gains of the sequence as given are all that is wanted)
        if (it == CCSX_MAX_ITER - 1) { nonconv = 1; break; }
        // ---- A5: greedy selection (wave 0; lane l owns m = l, l+64, l+128, l+192 which share position l&31)
        if (wave == 0) {
            int candmask = 0;
            for (int k = 0; k < 4; ++k) { int m = lane + 64 * k; if (sMvalid[m] && sDelta[m] > MUT_EPS) candmask |= 1 << k; }
            int Jn = J, nacc = 0;
            unsigned accpos = 0;                             // positions of the accepted mutations (at most one per position: they are >= MUT_SEP apart)
            int accm = 0;                                    // lane c (< 32): the mutation accepted at position c
            for (;;) {
                float bd = -1.0f; int bm = 1 << 20;
                for (int k = 0; k < 4; ++k) if (candmask & (1 << k)) { int m = lane + 64 * k; float dv = sDelta[m]; if (dv > bd) { bd = dv; bm = m; } }
                const float wmax = wave_max_f32(bd);
                if (!(wmax > 0.0f)) break;
                const int msel = wave_min_i32((bd == wmax) ? bm : (1 << 20));
                if (lane == (msel & 63)) candmask &= ~(1 << (msel >> 6));
                const int sl = msel >> 5, c = msel & 31;
                if (sl >= 4 && Jn >= CCSX_JMAX) continue;
                if (sl == 3 && Jn <= JMIN_DEL + 1) continue;
                if (sl >= 4) ++Jn; else if (sl == 3) --Jn;
                if (lane == c) accm = msel;
                accpos |= 1u << c;
                ++nacc;
                if (it >= MULTI_ROUNDS) break;
                int dc = (lane & 31) - c; if (dc < 0) dc = -dc;
                if (dc < MUT_SEP) candmask = 0;
            }
            {
                // apply in descending position order.  lane = column: the template base and the skip probability of every column sit in a register
                // and an insertion / deletion is one lane shift (round 3 walked both arrays in LDS on lane 0, a dependent load -> store per column)
                int Jc = J, cs = sCtl[1], ce = sCtl[2];
                unsigned ev = (unsigned)sCtl[7];
                int tb = lane < 32 ? (int)sT[0][lane] : 0;
                float ps = lane < 36 ? sPskip[lane] : 0.0f;
                while (accpos) {
                    const int c = 31 - __clz((int)accpos);
                    accpos &= ~(1u << c);
                    const int m = rl(accm, c), sl = m >> 5;
                    if (sl < 3) { if (lane == c) tb = (tb + 1 + sl) & 3; }
                    else if (sl >= 4) {
                        const int tu = __shfl_up(tb, 1);
                        const float pu = __shfl_up(ps, 1);
                        tb = lane < c ? tb : (lane == c ? sl - 4 : tu);
                        if (lane < 32) ps = lane < c ? ps : (lane == c ? 0.0f : pu);      // (entries 32 .. 35 stay, as in the column-by-column walk)
                        ++Jc;
                        if (c < cs) { ++cs; ++ce; } else if (c < ce) ++ce;
                        // the evidence bit and the skip probability travel with their base; an inserted base is a candidate
                        const unsigned lowm = (1u << c) - 1u;
                        ev = (ev & lowm) | ((ev & ~lowm) << 1);
                    } else {
                        const int td = __shfl_down(tb, 1);
                        const float pd = __shfl_down(ps, 1);
                        if (lane >= c) tb = td;
                        if (lane >= c && lane < 31) ps = pd;
                        --Jc;
                        if (c < cs) { --cs; --ce; } else if (c < ce) --ce;
                        const unsigned lowm = (1u << c) - 1u;
                        ev = (ev & lowm) | ((ev >> 1) & ~lowm);
                    }
                    // re-open the neighbourhood of the applied mutation for the following rounds
                    for (int q = c - SKIP_SPREAD; q <= c + SKIP_SPREAD; ++q) if (q >= 0 && q < 32) ev &= ~(1u << q);
                }
                if (lane < 32) sT[0][lane] = (uint8_t)tb;
                if (lane < 36) sPskip[lane] = ps;
                if (lane == 0) { sCtl[0] = Jc; sCtl[1] = cs; sCtl[2] = ce; sCtl[3] = nacc; sCtl[7] = (int)ev; }
            }
        }
        __syncthreads();
        if (sCtl[3] == 0) break;
    }
    __syncthreads();
    PHASE(5);
    // ---- A6: QVs of the core positions from the last scoring round
    const int J = sCtl[0], cs = sCtl[1], ce = sCtl[2];
    const size_t wi = (size_t)(P.wb_off[z] - z) + w;
    // every mutation lane's term exp2(min(delta, 20)) is computed by the lane's own thread (all waves at once; the core threads below used to
    // evaluate their 8 - 12 exponentials one after the other while three waves waited), then summed per position in the SPEC's order
#ifndef CCSX_EXP_NO_QV_EXP                                  // (experiment, timing only: the QV exponentials compiled out)
    if (tid < 256 && sMvalid[tid]) { float dv = sDelta[tid]; if (dv > 20.0f) dv = 20.0f; sDelta[tid] = det_exp2f(dv); }
#endif
    __syncthreads();
    // SPEC v7 "repeat-count floor": tandem tracts (period 1..4) of the converged template with its flanks, as wave masks (the core's threads sit in wave 0)
    unsigned long long tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0;
    const int voff = lf < 4 ? 1 : 0;
    if (wave == 0) {
        const int nvis = J + voff + (rf < 4 ? 1 : 0);
        auto vat = [&](int i) -> int { return i < voff ? lf : (i - voff < J ? (int)sT[0][i - voff < 31 ? i - voff : 31] : rf); };
        const int vx = vat(lane);
        tm1 = __ballot(lane + 1 < nvis && vx == vat(lane + 1));
        tm2 = __ballot(lane + 2 < nvis && vx == vat(lane + 2));
        tm3 = __ballot(lane + 3 < nvis && vx == vat(lane + 3));
        tm4 = __ballot(lane + 4 < nvis && vx == vat(lane + 4));
    }
    // (wave-uniform masks: whether ANY tract is long enough to matter is scalar arithmetic — on ordinary sequence none is, and the per-base work is skipped)
    auto has_run = [](unsigned long long m, int n) -> bool { unsigned long long x = m; for (int i = 1; i < n; ++i) x &= m >> i; return x != 0; };
    const bool any_tract = has_run(tm1, 8 - 1) || has_run(tm2, 10 - 2) || has_run(tm3, 12 - 3) || has_run(tm4, 16 - 4);
    float pl = 0.0f;                                        // this position's error probability
    if (tid < ce - cs) {
        const int c = cs + tid;
        float p;
        if ((skmask >> c) & 1u) p = sPskip[c];              // skipped by the candidate filter: error probability from the pile-up margin
        else {
            float s = (((unsigned)sCtl[7] >> c) & 1u) ? sPskip[c] : 0.0f;   // quiet homopolymer position: its untested mutations
            for (int sl = 0; sl < 8; ++sl) {
                int m = sl * 32 + c;
                if (sMvalid[m]) s = s + sDelta[m];
            }
            if (c == J - 1) for (int sl = 4; sl < 8; ++sl) {
                int m = sl * 32 + J;
                if (sMvalid[m]) s = s + sDelta[m];
            }
            p = __fdiv_rn(s, 1.0f + s);
        }
        if (any_tract) {
            const int x = c + voff;
            const int nvis_ = J + voff + (rf < 4 ? 1 : 0);
            const int T1 = tract_len(tm1, x, 1, nvis_), T2 = tract_len(tm2, x, 2, nvis_), T3 = tract_len(tm3, x, 3, nvis_), T4 = tract_len(tm4, x, 4, nvis_);
            // SPEC v8: a CLOSED tract (both ends inside the visible template) is resolved by the likelihood — its floor falls with the square of the passes used beyond REP_NP0
            const int npw = nvalid_last;
            const float closed_scale = npw > REP_NP0 ? __fdiv_rn((float)(REP_NP0 * REP_NP0), (float)(npw * npw)) : 1.0f;
            float fl = 0.0f;
#define CCSX_TRACT_FLOOR(TT, MINL) if (((TT) & 255) >= (MINL)) { float f = __fdiv_rn(REP_ERRS, (float)((TT) & 255)); if (!((TT) & 256) && npw > REP_NP0) f = f * closed_scale; fl = f > fl ? f : fl; }
            CCSX_TRACT_FLOOR(T1, 8) CCSX_TRACT_FLOOR(T2, 10) CCSX_TRACT_FLOOR(T3, 12) CCSX_TRACT_FLOOR(T4, 16)
#undef CCSX_TRACT_FLOOR
            if (p < fl) p = fl;
        }
        if (p < P.perr_floor) p = P.perr_floor;             // SPEC v7: no base claims more than Q50 (opts.max_qv)
        float qv = -3.01029996f * det_log2f(p);
        if (qv < 0.0f) qv = 0.0f;
        if (qv > 93.0f) qv = 93.0f;
        P.wseq[wi * 32 + tid] = sT[0][c];
        P.wqv[wi * 32 + tid] = qv;
        pl = p;
    }
    if (wave == 0) {                                         // the core positions (<= 32) all sit in wave 0: ordered sum by v_readlane
        float wsum = 0.0f;
        for (int k = 0; k < ce - cs; ++k) wsum = wsum + __int_as_float(rl(__float_as_int(pl), k));
      if (tid == 0) {
        P.wsum[wi] = wsum;
        P.wmeta[wi] = make_int4(ce - cs, nvalid_last | (nvfull_last << 8), nonconv, iters);
        if (P.wtmeta) P.wtmeta[wi] = make_short2((short)J, (short)cs);
      }
    }
    if (P.wtpl && tid < J) P.wtpl[wi * 32 + tid] = sT[0][tid];
#ifdef CCSX_EXP_REPEAT
    }
#endif
    PHASE(6);
    PHASE_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// N4: HiFi kinetics (docs/faq/kinetics.md:8-18; SPEC DESIGN.md §2.9).  One workgroup per window, one wave per read:
// lane = read row of a global alignment of the read's segment to the CONVERGED window template (read orientation),
// the insertion chain of a column is one fused DPP max-scan, the 2-bit moves of a row stay in the lane's registers
// (31 columns = 62 bits) and the traceback walks them with v_readlane on wave-uniform scalars — no DP matrix in
// memory at all.  Matching DIAG cells add the read base's decoded IPD / PW frames to integer LDS sums (order
// independent, so atomics keep the result deterministic); means are re-encoded with CodecV1.
__device__ __forceinline__ int codec_v1_decode(int c)
{
    return c < 64 ? c : (c < 128 ? 64 + (c - 64) * 2 : (c < 192 ? 192 + (c - 128) * 4 : 448 + (c - 192) * 8));
}
__device__ __forceinline__ int codec_v1_encode(int f)
{
    if (f < 64) return f;
    if (f < 192) return 64 + (f - 64 + 1) / 2;
    if (f < 448) return 128 + (f - 192 + 2) / 4;
    const int c = 192 + (f - 448 + 4) / 8;
    return c > 255 ? 255 : c;
}
__device__ __forceinline__ int kin_mean_code(unsigned sum, unsigned cnt)
{
    return cnt ? codec_v1_encode((int)((2u * sum + cnt) / (2u * cnt))) : 0;
}

// scalar traceback of one read: rows live in lanes base..base+n; bit c of (upm, lfm) in the lane of row i says the
// move of cell (i, c+1) is UP / LEFT (neither = DIAG).  Returns, for lane base+c, the read row matched (DIAG) to
// column c of the read-oriented template, or -1.
__device__ __forceinline__ int kin_traceback(int upm, int lfm, int n, int J, int base, int lane)
{
    int i = n, j = J, myrow = -1;
    while (i > 0 && j > 0) {                                 // DIAG needs both; the rest of the path is a straight edge
        const unsigned su = (unsigned)rl(upm, base + i) >> (j - 1), sl = (unsigned)rl(lfm, base + i) >> (j - 1);
        const int up = su & 1u, lf = sl & 1u;
        if (!(up | lf) && lane == base + j - 1) myrow = i - 1;
        i -= 1 - lf;
        j -= 1 - up;
    }
    return myrow;
}

// the same walk for segments of <= 31 bases, one iteration per DIAG RUN instead of per cell: nd = (up | left) << (32 - row)
// puts the cells of one diagonal (i - j constant) at the same bit position in every lane, so a single
// ballot lists the non-DIAG cells of the current diagonal and the highest one at or below row i ends the run.  A
// typical window has ~3 indels per read: ~4 iterations instead of ~30 steps.
__device__ __forceinline__ int kin_traceback_runs(int upm, unsigned long long nd, int n, int J, int base, int lane)
{
    int i = n, j = J, myrow = -1;
    const int c = lane - base;
    while (i > 0 && j > 0) {
        const int p = j + 31 - i;                            // bit of cell (i', j') on this diagonal: (j'-1) + 32 - i'
        const unsigned rows = (unsigned)(__ballot((int)((nd >> p) & 1ull)) >> base);
        const unsigned below = rows & (unsigned)((2ull << i) - 1ull);         // non-DIAG cells at rows <= i (row 0 always is one)
        const int istar = 31 - __builtin_clz(below | 1u);
        int run = i - istar;
        if (run > j) run = j;
        if (c >= j - run && c < j) myrow = c + i - j;        // DIAG cell (c + 1 + i - j, c + 1) matches read row c + i - j
        i -= run; j -= run;
        if (i > 0 && j > 0) {                                // (i, j) is UP or LEFT
            if (((unsigned)rl(upm, base + i) >> (j - 1)) & 1u) --i; else --j;
        }
    }
    return myrow;
}

__global__ __launch_bounds__(256) void k_kinetics(KParams P, int slot0)
{
    __shared__ uint8_t sT[2][32];
    __shared__ unsigned sK[2][3][32];                       // [strand][ipd, pw, count][forward column]
    __shared__ int sN[PW_MAXREADS], sOff[PW_MAXREADS];      // segment length (-1 = unusable), offset of the segment in the read
    __shared__ uint8_t sSt[PW_MAXREADS];
    __shared__ short2 sTask[PW_MAXREADS];
    __shared__ int sNT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = slot0 + xcd_contiguous(blockIdx.x, polish_piece_windows(P.wstart[P.n_zmw], slot0, (int)gridDim.x));   // (as k_polish)
    if (bid < slot0) return;
    const int z = P.wslot_zmw[bid];
    const int w = bid - P.wstart[z];
    const int nw = P.nwin[z];
    const size_t wi = (size_t)(P.wb_off[z] - z) + w;
    const short2 tm = P.wtmeta[wi];
    const int J = tm.x, cs = tm.y, ce = cs + P.wmeta[wi].x;
    const int r0 = P.read_off[z], nreads = P.nreads_used[z];
    const int idx_ws = (w == 0) ? 0 : 2 * w - 1, idx_we = (w == nw - 1) ? 2 * nw - 1 : 2 * (w + 1);
    if (tid < J) { const uint8_t b = P.wtpl[wi * 32 + tid]; sT[0][tid] = b; sT[1][J - 1 - tid] = (uint8_t)(3 - b); }
    if (tid < 192) (&sK[0][0][0])[tid] = 0u;
    // SPEC v5: up to CCSX_MAX_PASSES passes, taken in groups of PW_MAXREADS (the sums in sK run over all groups)
    for (int g0 = 0; g0 < nreads; g0 += PW_MAXREADS) {
    const int ng = nreads - g0 < PW_MAXREADS ? nreads - g0 : PW_MAXREADS;
    __syncthreads();
    if (tid < ng) {                                          // one lane per read: segment of the read inside this window
        const int rr = r0 + g0 + tid;
        int n = -1, off = 0;
        const int st = ((P.flags[rr] ^ P.flags[r0 + (P.zref[z] & 255)]) & 1) ? 1 : 0;
        if (P.avalid[rr]) {
            const int32_t *ent = P.ent + P.ent_off[rr];
            const int a = ent[idx_ws], b = ent[idx_we];
            n = b - a;
            if (n < 0 || n > CCSX_IMAX) n = -1;
            off = st ? (int)(P.base_off[rr + 1] - P.base_off[rr]) - b : a;
        }
        sN[tid] = n; sOff[tid] = off; sSt[tid] = (uint8_t)st;
    }
    __syncthreads();
    if (tid == 0) {                                          // tasks: two short segments share a wave (32 lanes each)
        int nt = 0, pend = -1;
        for (int r = 0; r < ng; ++r) {
            const int n = sN[r];
            if (n < 0) continue;
            if (n > 31) sTask[nt++] = make_short2((short)r, (short)-1);
            else if (pend < 0) pend = r;
            else { sTask[nt++] = make_short2((short)pend, (short)r); pend = -1; }
        }
        if (pend >= 0) sTask[nt++] = make_short2((short)pend, (short)-1);
        sNT = nt;
    }
    __syncthreads();
    const int ntask = sNT;
    unsigned tlo[2], thi[2];                                 // template bit planes per strand: bit c = column c (wave-uniform)
    for (int sd = 0; sd < 2; ++sd) {
        const int b = lane < J ? sT[sd][lane] : 0;
        tlo[sd] = (unsigned)__ballot(b & 1); thi[sd] = (unsigned)__ballot(b & 2);
    }
    for (int tk = wave; tk < ntask; tk += 4) {
        const short2 task = sTask[tk];
        const bool paired = rfl((int)task.y) >= 0;
        const int half = (paired && lane >= 32) ? 1 : 0;
        const int base = half * 32, row = lane - base;
        const int myr = half ? task.y : task.x;
        const int n = sN[myr], st = sSt[myr];
        const int64_t p0 = P.base_off[r0 + g0 + myr] + sOff[myr];
        const int rbv = (row >= 1 && row <= n) ? (P.bases[p0 + row - 1] & 3) : 0;   // lane of row i holds read base i-1
        const uint8_t *t = sT[st];
        // bit c of mt: read base of this row == template column c (the two bit planes of the template come as ballots)
        const unsigned mt = ~((tlo[st] ^ ((rbv & 1) ? ~0u : 0u)) | (thi[st] ^ ((rbv & 2) ? ~0u : 0u)));
        // scores are kept shifted by +5 inside a column (h5 = h + 5): diag5 = x + 8*match, left5 = H + 1.  Rows beyond
        // the read compute garbage that no lower lane and no traceback step ever reads.
        const int c1 = 4 * row - 5;
        int H = row * SC_INS;
        int upm = 0, lfm = 0;
        for (int j = 1; j <= J; ++j) {
            int x = wave_shr1_i32(H, NEGV);
            if (row == 0) x = NEGV;                          // (lane 32 of a pair would otherwise see the other read)
            const int diag5 = x + (int)(((mt >> (j - 1)) & 1u) << 3);
            const int left5 = H + 1;
            const int h5 = diag5 >= left5 ? diag5 : left5;
            // insertion chain: v(i) = max_k<=i h(k) + (i-k)*SC_INS, one fused DPP max-scan (per half when paired)
            const int d0 = h5 + c1;
            const int v = (paired ? half_scan_max_i32(d0) : wave_scan_max_i32(d0)) - 4 * row;     // = v(i) exactly
            const unsigned bit = 1u << (j - 1);
            if (v + 5 > h5) upm |= bit;
            else if (diag5 < left5) lfm |= bit;
            H = v;
        }
        int myrow;
        const int nA = rfl(sN[task.x]);
        if (nA <= 31) {
            const unsigned long long nd = (unsigned long long)(unsigned)(upm | lfm) << (32 - (row & 31));
            myrow = kin_traceback_runs(upm, nd, nA, J, 0, lane);
            if (paired) {
                const int rowB = kin_traceback_runs(upm, nd, rfl(sN[task.y]), J, 32, lane);
                if (half) myrow = rowB;
            }
        } else myrow = kin_traceback(upm, lfm, nA, J, 0, lane);
        if (row < J && myrow >= 0) {
            const int jf = st ? J - 1 - row : row;
            const int64_t p = p0 + myrow;
            if (jf >= cs && jf < ce && (P.bases[p] & 3) == t[row]) {
                atomicAdd(&sK[st][0][jf], (unsigned)codec_v1_decode(P.ipd[p]));
                atomicAdd(&sK[st][1][jf], (unsigned)codec_v1_decode(P.pw[p]));
                atomicAdd(&sK[st][2][jf], 1u);
            }
        }
    }
    }
    __syncthreads();
    if (tid < ce - cs) {
        const int c = cs + tid;
        uchar4 k;
        k.x = (uint8_t)kin_mean_code(sK[0][0][c], sK[0][2][c]); k.y = (uint8_t)kin_mean_code(sK[0][1][c], sK[0][2][c]);
        k.z = (uint8_t)kin_mean_code(sK[1][0][c], sK[1][2][c]); k.w = (uint8_t)kin_mean_code(sK[1][1][c], sK[1][2][c]);
        P.wkin[wi * 32 + tid] = k;
    }
}

// ------------------------------------------------------------------------------------------------
// step 10: concatenate window cores; rq = 1 - mean(p_err); ec, np, status.  One wave per ZMW.
__global__ __launch_bounds__(64) void k_stitch(KParams P)
{
    __shared__ double sSum;
    __shared__ int sHist[256];                              // windows by number of passes used (0 .. CCSX_MAX_PASSES)
    __shared__ int sOff[LANES], sLen[LANES];                // a block of 64 windows: offset of each core in the block's output, its length
    __shared__ float sQ[LANES * 32];                        // ... the block's QVs and bases, compacted
    __shared__ uint8_t sS[LANES * 32];
    const int z = blockIdx.x, lane = threadIdx.x;
    int stat = P.zstat[z];
    const int nw = (stat == CCSX_SUCCESS) ? P.nwin[z] : 0;
    const size_t w0 = (size_t)(P.wb_off[z] - z);
    const int64_t so = P.seq_off[z], cap = P.seq_off[z + 1] - so;
    int run = 0, nvs = 0, its = 0, ncv = 0;
    if (lane == 0) sSum = 0.0;
    for (int q = lane; q < 256; q += LANES) sHist[q] = 0;
    __syncthreads();
    for (int wbase = 0; wbase < nw; wbase += LANES) {
        const int w = wbase + lane;
        int4 mt = make_int4(0, 0, 0, 0);
        if (w < nw) { mt = P.wmeta[w0 + w]; if ((unsigned)(mt.y >> 8) <= (unsigned)CCSX_MAX_PASSES) atomicAdd(&sHist[mt.y >> 8], 1); mt.y &= 255; }   // np: full-length passes; ec: all
        int pre = mt.x;                                     // inclusive scan of lengths
#pragma unroll
        for (int s = 1; s < LANES; s <<= 1) { int o = __shfl_up(pre, s); if (lane >= s) pre += o; }
        const int off = run + pre - mt.x;
        // The block's 64 window rows are read row-major (coalesced) and compacted in LDS, then written out position-major (coalesced).  Before, a lane copied its
        // own window base by base: 4-byte stores 88 bytes apart — the kernel wrote 1.3 MB per ZMW for 60 KB of results (PMC, profiles/r04_traffic.json).
        sOff[lane] = pre - mt.x; sLen[lane] = mt.x;
        __syncthreads();
        const int nrow = (nw - wbase < LANES ? nw - wbase : LANES) * 32;
        const size_t rowbase = (w0 + (size_t)wbase) * 32;
        for (int q0 = 0; q0 < nrow; q0 += 4 * LANES) {
            float fq[4]; uint8_t fs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int q = q0 + u * LANES + lane; const int qc = q < nrow ? q : nrow - 1; fq[u] = P.wqv[rowbase + qc]; fs[u] = P.wseq[rowbase + qc]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * LANES + lane;
                if (q < nrow) { const int wl = q >> 5, k = q & 31; if (k < sLen[wl]) { const int d = sOff[wl] + k; sQ[d] = fq[u]; sS[d] = fs[u]; } }
            }
        }
        __syncthreads();
        const int tot = __shfl(pre, 63);
        for (int p0 = lane; p0 < tot; p0 += LANES) {
            if (run + p0 < cap) {
                const float qv = sQ[p0];
                P.out_seq[so + run + p0] = sS[p0];
                P.out_raw[so + run + p0] = qv;
                P.out_qual[so + run + p0] = (uint8_t)(qv + 0.5f);
            }
        }
        if (P.out_kin) {                                    // (--hifi-kinetics only: the four kinetics planes keep the window-by-window copy)
            for (int k = 0; k < mt.x; ++k) {
                if (off + k < cap) {
                    const uchar4 kk = P.wkin[(w0 + w) * 32 + k];
                    uint8_t *o = P.out_kin + so + off + k;
                    o[0] = kk.x; o[P.kin_plane] = kk.y; o[2 * P.kin_plane] = kk.z; o[3 * P.kin_plane] = kk.w;
                }
            }
        }
        run += tot;
        nvs += mt.y; ncv |= mt.z; its += mt.w;
        // fixed-order (window order) double sum of per-window float sums
        for (int k = 0; k < LANES; ++k) {
            float v = __shfl((w < nw) ? P.wsum[w0 + w] : 0.0f, k);
            if (lane == 0 && wbase + k < nw) sSum += (double)v;
        }
        __syncthreads();                                    // (the next block reuses sOff / sLen / sQ / sS)
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { nvs += __shfl_xor(nvs, s); its += __shfl_xor(its, s); ncv |= __shfl_xor(ncv, s); }
    __syncthreads();
    // np = mode over windows of the passes used for polishing (docs/faq/accuracy-vs-passes.md:18-24); ties: the smaller count
    int npmode;
    {
        int key = 0;                                         // (windows << 8) | (255 - passes): the most windows, then the smaller count
        for (int v = lane; v < 256; v += LANES) { const int k = (sHist[v] << 8) | (255 - v); key = k > key ? k : key; }
        npmode = 255 - (wave_max_i32(key) & 255);
    }
    if (lane == 0) {
        if (stat == CCSX_SUCCESS) P.np[z] = npmode;
        int64_t len = run;
        const bool overflow = len > cap;                    // never a silently truncated HiFi read
        if (overflow) len = cap;
        float rq = 0.0f, ec = 0.0f;
        if (stat == CCSX_SUCCESS) {
            rq = len > 0 ? (float)(1.0 - sSum / (double)run) : 0.0f;
            ec = nw > 0 ? (float)((double)nvs / (double)nw) : 0.0f;
            if (overflow) { stat = CCSX_CAPACITY; len = 0; }
            else if (len == 0) stat = CCSX_EMPTY_WINDOW;
            else if (ncv) stat = CCSX_NON_CONVERGENT;
            else if (rq < P.opts.min_rq) stat = CCSX_LOW_RQ;
        } else len = 0;
        P.out_status[z] = stat; P.out_len[z] = (int32_t)len; P.out_rq[z] = rq; P.out_ec[z] = ec;
        P.out_iters[z] = its; P.out_nwin[z] = nw;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from ccsx_api.cpp, which is plain C++)
static void trace_sync(hipStream_t st, const char *what)
{
    static const bool on = getenv("CCSX_TRACE") != nullptr;     // debugging aid: serialise and name every launch
    if (!on) return;
    hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[ccsx] %s done: %s\n", what, hipGetErrorString(e));
}

// dynamic LDS of k_polish: [reads][68] observations (a byte each) for the largest ZMW of the batch (at most one group of PW_MAXREADS), the rest of the workgroup's
// budget holds gamma/beta of one chunk of reads
int ccsx_polish_lds(int max_reads, int *obs_bytes, int *gb_floats)
{
    hipFuncAttributes fa;                                  // per call: the attribute is per device, handles live on several
    const void *fn = (const void *)k_polish_t<PW_THREADS, PW_MINWAVES, PW_CHUNK_READS>;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return -1;
    const int static_bytes = (int)fa.sharedSizeBytes;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_BYTES - static_bytes) != hipSuccess) return -1;
    if (max_reads > PW_MAXREADS) max_reads = PW_MAXREADS;
    if (max_reads < 1) max_reads = 1;
    *obs_bytes = ((max_reads * 68) + 15) & ~15;
    *gb_floats = (PW_LDS_BYTES - static_bytes - *obs_bytes) / 4;
    return 0;
}

// every launch status is captured: returns NULL, or the name of the first launch that failed (ccsx_api.cpp reports it)
#define LAUNCH_CHECK(name) do { if (hipGetLastError() != hipSuccess && !failed) failed = name; } while (0)
const char *ccsx_launch_all(const KParams &P, hipStream_t st, hipStream_t st_polish, hipEvent_t *ev /* [7] or NULL */, int mode, hipStream_t st_aux, hipEvent_t *ev_aux /* [7] or NULL */)
{
    // Two-stage queue of docs/img/ccs-impl.png ("Draft Stage" -> queue -> "Polish Stage"): the draft stage (tables, POA, alignment
    // cascade, accounting) is enqueued on `st`, the polish stage (polish, kinetics, stitch) on `st_polish`, which waits for the
    // draft stage's last kernel through ev[3].  With two different streams the draft stage of batch k+1 runs UNDER the polish
    // stage of batch k (the register-only one-wave POA and the LDS-bound polish workgroups share the SIMDs); the shared POA /
    // alignment scratch is touched by the draft stage only, so `st` alone orders its users.  st_polish == st: serial stages.
    const char *failed = nullptr;
    if (ev && hipEventRecord(ev[0], st) != hipSuccess) failed = "hipEventRecord";
    if (hipMemsetAsync(P.ticket_poa, 0, 256, st) != hipSuccess && !failed) failed = "hipMemsetAsync";   // debug / phase-profile words (CCSX_DEBUG_CHECKS, CCSX_PROFILE_PHASES builds)
    if (hipMemsetAsync(P.avalid, 0, (size_t)(P.n_reads > 0 ? P.n_reads : 1), st) != hipSuccess && !failed) failed = "hipMemsetAsync";   // passes beyond top_passes are never visited by a kernel
    {
        int n = P.n_zmw * CCSX_NCTX;
        hipLaunchKernelGGL(k_setup, dim3((n + 255) / 256), dim3(256), 0, st, P);
        LAUNCH_CHECK("k_setup");
    }
    trace_sync(st, "k_setup");
    if (ev) (void)hipEventRecord(ev[1], st);
    const size_t lds_read = (((size_t)P.maxL_max + 15) / 16) * 4 + 64 + 4 * (CCSX_MAX_PASSES + 1);   // the packed read; k_poa_init: the lengths of up to 255 passes
    // pass 0 = the draft; pass 1 = the fallback draft of the ZMWs k_post marked (their waves run, all others leave at once:
    // the second round of launches costs microseconds unless something failed)
    // CCSX_RUN_POLISH (the polish seam): the drafts are the caller's — k_draft_in instead of the generators, one alignment round whose outcome is final
    // (P.opts.no_fallback_draft is set for such a run)
    if (mode == CCSX_RUN_POLISH) {
        hipLaunchKernelGGL(k_draft_in, dim3(P.n_zmw), dim3(64), 0, st, P);
        LAUNCH_CHECK("k_draft_in");
    }
    for (int pass = 0; pass < ((P.opts.no_fallback_draft || mode == CCSX_RUN_POLISH) ? 1 : 3); ++pass) {
        int cov = pass ? 2 * P.opts.max_poa_cov : P.opts.max_poa_cov;
        if (cov > PW_MAXREADS_SPEC) cov = PW_MAXREADS_SPEC;
        if (cov > P.max_reads) cov = P.max_reads;          // no ZMW of the batch has more passes
        for (int z0 = 0; z0 < P.n_zmw && mode != CCSX_RUN_POLISH; z0 += P.poa_slots) {
            const int nb = (P.n_zmw - z0) < P.poa_slots ? (P.n_zmw - z0) : P.poa_slots;
            // the graphs [g0, g0 + ng) on stream s: initial graph, one DP (four graphs per wave) + one threading kernel per pass of the POA, heaviest path
            auto poa_range = [&](hipStream_t s, int g0, int ng, hipEvent_t after_first_dp, hipEvent_t before_first_dp) {
                hipLaunchKernelGGL(k_poa_init, dim3(ng), dim3(64), lds_read, s, P, z0, pass, g0);
                LAUNCH_CHECK("k_poa_init");
                if (before_first_dp && hipStreamWaitEvent(s, before_first_dp, 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
                for (int rr = 1; rr < cov && pass < 2; ++rr) {
                    hipLaunchKernelGGL(k_poa_dp, dim3((ng + 3) / 4), dim3(64), 0, s, P, z0, pass, rr, g0 / 4);
                    LAUNCH_CHECK("k_poa_dp");
                    if (rr == 1 && after_first_dp && hipEventRecord(after_first_dp, s) != hipSuccess && !failed) failed = "hipEventRecord";
                    hipLaunchKernelGGL(k_poa_thread, dim3(ng), dim3(64), lds_read, s, P, z0, pass, rr, g0);
                    LAUNCH_CHECK("k_poa_thread");
                }
                if (pass < 2) {                            // (pass 2 = last resort: k_poa_init writes the draft itself)
                    hipLaunchKernelGGL(k_poa_finish, dim3(ng), dim3(64), 0, s, P, z0, pass, g0);
                    LAUNCH_CHECK("k_poa_finish");
                }
            };
            // Pass 0 of a large batch runs as TWO half-batches on two streams, the second one a DP behind the first: k_poa_dp saturates the VALU and k_poa_thread
            // waits for HBM, their registers and LDS fit one SIMD together (94 + 64 VGPRs), so the threading of one half runs under the DP of the other.
            if (st_aux && ev_aux && pass == 0 && cov > 1 && nb >= 4096) {
                const int ha = ((nb / 2) + 3) & ~3;
                if (hipEventRecord(ev_aux[0], st) != hipSuccess && !failed) failed = "hipEventRecord";
                if (hipStreamWaitEvent(st_aux, ev_aux[0], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
                poa_range(st, 0, ha, ev_aux[1], nullptr);
                poa_range(st_aux, ha, nb - ha, nullptr, ev_aux[1]);
                if (hipEventRecord(ev_aux[2], st_aux) != hipSuccess && !failed) failed = "hipEventRecord";
                if (hipStreamWaitEvent(st, ev_aux[2], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
            } else poa_range(st, 0, nb, nullptr, nullptr);
        }
        trace_sync(st, "k_poa");
        if (ev && pass == 0) (void)hipEventRecord(ev[2], st);
        // alignment cascade: four passes per wave in 16-row bands, then the 64-row retry of the few that failed there
        if (hipMemsetAsync(P.align_retry, 0, 64, st) != hipSuccess && !failed) failed = "hipMemsetAsync";
        bool tb_aside_out = false; int tb_launches = 0;
        {
            const size_t lds16 = 4 * (CH16 / 16 + 3) * sizeof(uint32_t);
            // The trace-back of a launch runs on the second stream: beside the NEXT launch of k_align16 (a batch whose quads take several launches has two
            // scratch regions, launch c uses region c & 1 and waits for the trace-back of launch c - 2) and, the last one, beside the 64-row retry and the split
            // alignment — those read the retry list the 16-row kernel wrote, not the entries the trace-back writes, and their scratch lies behind the stored
            // moves.  k_post waits for all of them.  (A trace-back is a long dependent walk of few waves: 42 % of k_align16's time on the 3-50-pass mix.)
            static const bool aside_ok = [] { const char *e = getenv("CCSX_TB_ASIDE"); return !(e && e[0] == '0'); }();   // (A/B switch)
            const bool aside = aside_ok && st_aux && ev_aux && pass == 0 && P.n_quads >= 4096;
            bool tb_aside = false;
            int c = 0;
            for (int qb = 0; qb < P.n_quads; qb += P.align16_slots, ++c) {
                const int nb = (P.n_quads - qb) < P.align16_slots ? (P.n_quads - qb) : P.align16_slots;
                const int region = P.align16_regions > 1 ? (c & 1) : 0;
                if (aside && c >= 2 && hipStreamWaitEvent(st, ev_aux[5 + (c & 1)], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";   // the region's last reader
                if (aside && c >= 1 && P.align16_regions <= 1 && hipStreamWaitEvent(st, ev_aux[5 + ((c - 1) & 1)], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";   // (one region: in sequence)
                hipLaunchKernelGGL(k_align16, dim3(nb), dim3(64), lds16, st, P, qb, pass, region);
                LAUNCH_CHECK("k_align16");
                hipStream_t s_tb = st;
                if (aside) {
                    if (hipEventRecord(ev_aux[3 + (c & 1)], st) != hipSuccess && !failed) failed = "hipEventRecord";
                    if (hipStreamWaitEvent(st_aux, ev_aux[3 + (c & 1)], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
                    s_tb = st_aux; tb_aside = true;
                }
                hipLaunchKernelGGL(k_align16_tb, dim3((4 * nb + 63) / 64), dim3(64), 0, s_tb, P, qb, nb, region);   // one lane per pass: entry rows / dirty masks from the stored moves
                LAUNCH_CHECK("k_align16_tb");
                if (aside && hipEventRecord(ev_aux[5 + (c & 1)], st_aux) != hipSuccess && !failed) failed = "hipEventRecord";
            }
            tb_launches = c;
            tb_aside_out = tb_aside;
        }
        {
            const int g = P.align_slots < 1 ? 1 : (P.align_slots > 4096 ? 4096 : P.align_slots);
            hipLaunchKernelGGL(k_align, dim3(g), dim3(64), lds_read, st, P, pass);
            LAUNCH_CHECK("k_align");
        }
        trace_sync(st, "k_align");
        if (P.align_slots >= 2) {                          // the split alignment uses two scratch slots per workgroup
            const int g = P.align_slots / 2 > 2048 ? 2048 : P.align_slots / 2;
            hipLaunchKernelGGL(k_rescue, dim3(g), dim3(64), lds_read, st, P, pass);
            LAUNCH_CHECK("k_rescue");
        }
        if (tb_aside_out) for (int c = tb_launches > 2 ? tb_launches - 2 : 0; c < tb_launches; ++c)      // (the trace-backs run in order on one stream: the last one per region)
            if (hipStreamWaitEvent(st, ev_aux[5 + (c & 1)], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
        hipLaunchKernelGGL(k_post, dim3((P.n_zmw + 255) / 256), dim3(256), 0, st, P, pass);
        LAUNCH_CHECK("k_post");
    }
    if (mode == CCSX_RUN_DRAFT) {                                   // the draft seam ends here: drafts, window bounds, alignments and statuses are final
        if (ev) for (int k : {3, 6, 4, 5}) if (hipEventRecord(ev[k], st) != hipSuccess && !failed) failed = "hipEventRecord";
        return failed;
    }
    hipLaunchKernelGGL(k_wmap, dim3(1), dim3(1024), 0, st, P);     // the batch's windows in compact order: the polish stage's grid map
    LAUNCH_CHECK("k_wmap");
    hipLaunchKernelGGL(k_wmap_fill, dim3((P.n_zmw + 3) / 4), dim3(256), 0, st, P);
    LAUNCH_CHECK("k_wmap_fill");
    if (ev) {
        if (hipEventRecord(ev[3], st) != hipSuccess && !failed) failed = "hipEventRecord";
        if (st_polish != st && hipStreamWaitEvent(st_polish, ev[3], 0) != hipSuccess && !failed) failed = "hipStreamWaitEvent";
        (void)hipEventRecord(ev[6], st_polish);            // the polish stage starts here (after the queue between the stages)
    } else if (st_polish != st && !failed) failed = "two streams need events";
    st = st_polish;
    // One workgroup per window slot.  A grid may not exceed 2^32 threads in all: 256 threads x 16.7 M slots — 8192 ZMWs of 30 passes x 20 kb have 10.8 M, and a
    // larger batch would silently lose its tail (round 4 met exactly this with a 512-thread experiment: 75 % of the ZMWs "failed").  The slots are therefore
    // launched in pieces of at most 2^24 - 256 workgroups (just under the limit, so that a 16384-ZMW batch of 10 kb inserts — 8.5 M slots of capacity — is ONE launch:
    // the profile's per-launch average and bench.py's per-batch duration then describe the same thing; CCSX_POLISH_MAX_BLOCKS: a test hook that forces small pieces).
    static const long long max_blocks = [] { const char *e = getenv("CCSX_POLISH_MAX_BLOCKS"); long long v = e ? atoll(e) : 0; return v > 8 ? (v & ~7ll) : (1ll << 24) - 256; }();   // (a multiple of 8: the kernels take their windows in XCD-contiguous order)
    for (long long s0 = 0; s0 < P.total_wslots; s0 += max_blocks) {
        const unsigned nb = (unsigned)((P.total_wslots - s0) < max_blocks ? (P.total_wslots - s0) : max_blocks);
        hipLaunchKernelGGL((k_polish_t<PW_THREADS, PW_MINWAVES, PW_CHUNK_READS>), dim3((nb + 7u) & ~7u), dim3(PW_THREADS),
                           (size_t)P.pw_obs_bytes + (size_t)P.pw_gb_floats * 4, st, P, (int)s0);
        LAUNCH_CHECK("k_polish");
    }
    trace_sync(st, "k_polish");
    if (P.opts.hifi_kinetics) {
        for (long long s0 = 0; s0 < P.total_wslots; s0 += max_blocks) {
            const unsigned nb = (unsigned)((P.total_wslots - s0) < max_blocks ? (P.total_wslots - s0) : max_blocks);
            hipLaunchKernelGGL(k_kinetics, dim3((nb + 7u) & ~7u), dim3(256), 0, st, P, (int)s0);
            LAUNCH_CHECK("k_kinetics");
        }
        trace_sync(st, "k_kinetics");
    }
    if (ev) (void)hipEventRecord(ev[4], st);
    hipLaunchKernelGGL(k_stitch, dim3(P.n_zmw), dim3(64), 0, st, P);
    LAUNCH_CHECK("k_stitch");
    if (ev && hipEventRecord(ev[5], st) != hipSuccess && !failed) failed = "hipEventRecord";
    return failed;
}
