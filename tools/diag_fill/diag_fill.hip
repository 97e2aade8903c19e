// diag_fill.hip — micro-benchmark for the diagonal-band alpha fill proposed in DESIGN.md §8.8 (a TOOL, not part of the library).
//
// Kernel B ("staircase"): a lane owns two adjacent diagonals of a read's band (dlo + 2m, dlo + 2m + 1) and walks a staircase: per iteration h it computes the
// cell (i, j0) = (h - m, dlo + m + h) on its even diagonal, then (i, j0 + 1) on its odd one.  Eight reads of ~ 15 diagonals share one wave64 sweep.
// Kernel A ("rows"): the shape of today's fill — lane = read row, two reads per wave, anti-diagonal sweep over the FULL matrix.
// Both are written plainly (no hand pipelining of the look-ups), compute alpha and store gamma to LDS as the product does, and are checked bit for bit against a
// column-order CPU fill (kernel B against the banded one, kernel A against the full one; -ffp-contract=off on both sides).  Reported: ns per (read, window) alpha
// sweep at full occupancy and the VALU / LDS instructions per read the ISA shows.  tools/diag_fill_model.py is the same schedule in numpy.
//   build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/diag_fill/diag_fill.hip -o /tmp/diag_fill && /tmp/diag_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define J 26                          // window columns
#define NOBS 12
#define CTXS 33                       // pairs per observation row: 32 contexts + one zero pad (row NOBS and column 32 are zeros)
#define RPW 8                         // reads per window
#define OG 16                         // guard entries ("no base" = row NOBS) on both sides of a read's observation codes
#define OBSROW (OG + 64 + OG)
#define CG 16                         // guard entries of the column table (columns -CG .. J + CG)
#define SCORE_BAND 5
#define FILL_BAND 2
#define GMAX_B 704                    // floats of gamma per read slot in LDS: banded, (I + 1) * BW <= 30 * 23
#define GMAX_A 928                    // ... full matrix, row stride 28

struct Tables { float2 pair[(NOBS + 1) * CTXS]; float dl[17]; };
struct ReadIn { int I; unsigned char obs[64]; };
struct Task { int nread; int read[RPW]; int lane0[RPW]; int hlo, hhi; };

__device__ __host__ inline void band_of(int I, int *dlo, int *dhi)
{
    const int dIJ = I > J ? I - J : J - I;
    const int W = FILL_BAND + SCORE_BAND + (dIJ > 2 ? dIJ - 2 : 0);
    *dlo = (J - I < 0 ? J - I : 0) - W; *dhi = (J - I > 0 ? J - I : 0) + W;
}

__device__ __forceinline__ float wave_shr1_z(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float wave_shl1_z(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true)); }

// shared set-up: tables, the window's column entries (DL of column j, byte offset of its context in a pair row), the reads' observation codes
__device__ __forceinline__ void stage(const Tables *T, const unsigned char *ctx, const ReadIn *reads, const int *rid, int nread,
                                      float2 *sPair, int2 *sCol, unsigned char (*sObs)[OBSROW], int *sI)
{
    const int lane = threadIdx.x;
    for (int e = lane; e < (NOBS + 1) * CTXS; e += 64) sPair[e] = T->pair[e];
    for (int e = lane; e < J + 2 * CG + 1; e += 64) {
        const int j = e - CG;
        const int k = (j >= 0 && j < J) ? ctx[j] : 32;                       // column J and every column outside the window: the zero pad, DL = 1
        sCol[e] = make_int2(__float_as_int(k < 32 ? T->dl[k] : 1.0f), k * 8);
    }
    for (int q = 0; q < RPW; ++q) {
        const bool have = q < nread;
        const int I = have ? reads[rid[q]].I : 0;
        if (lane == 0) sI[q] = I;
        for (int e = lane; e < OBSROW; e += 64) { const int i = e - OG; sObs[q][e] = (have && i >= 0 && i < I) ? reads[rid[q]].obs[i] : (unsigned char)NOBS; }
    }
    __syncthreads();
}

// ---- kernel B: the staircase.  One wave = one task (up to RPW reads, lanes assigned by the host)
__global__ __launch_bounds__(64) void k_stair(const Tables *T, const unsigned char *ctx, const ReadIn *reads, const Task *tasks, float *out, int reps)
{
    __shared__ float2 sPair[(NOBS + 1) * CTXS];
    __shared__ int2 sCol[J + 2 * CG + 1];
    __shared__ unsigned char sObs[RPW][OBSROW];
    __shared__ int sI[RPW];
    __shared__ float sG[RPW][GMAX_B];
    const Task tk = tasks[blockIdx.x];
    const int lane = threadIdx.x;
    stage(T, ctx, reads, tk.read, tk.nread, sPair, sCol, sObs, sI);
    // which read, which pair of diagonals
    int q = -1;
    for (int r = 0; r < tk.nread; ++r) if (lane >= tk.lane0[r]) q = r;
    int m = 0, I = 0, dlo = 0, dhi = -1;
    if (q >= 0) { I = sI[q]; band_of(I, &dlo, &dhi); m = lane - tk.lane0[q]; if (2 * m > dhi - dlo) { q = -1; m = 0; } }
    const int BW = dhi - dlo + 1;
    const bool hasE = q >= 0, hasO = q >= 0 && 2 * m + 1 <= BW - 1;
    const bool first = m == 0, last = q < 0 || 2 * m + 2 > BW - 1;           // no lane of this read on the left / right
    const int qq = q < 0 ? 0 : q;
    for (int rep = 0; rep < reps; ++rep) {
        float E = 0.0f, O = 0.0f;
        int h = tk.hlo;
        int i = h - m, j0 = dlo + m + h;
        const unsigned char *op = &sObs[qq][OG + i - 1];                     // code of row i - 1 (guards: "no base")
        const int2 *cp = &sCol[CG + j0 - 1];                                 // entries of columns j0 - 1, j0, j0 + 1
        float *gp = &sG[qq][0] + (h * BW - m * (BW - 2));                    // gamma(i, j0): i * (BW - 1) + (j0 - dlo)
        int2 c0 = cp[0], c1 = cp[1];
        for (; h <= tk.hhi; ++h, ++i, ++j0, ++op, ++cp, gp += BW) {
            const int2 c2 = cp[2];
            const int orow = (int)op[0] * (CTXS * 8);
            const float2 P0 = *(const float2 *)((const char *)sPair + orow + c0.y);
            const float2 P1 = *(const float2 *)((const char *)sPair + orow + c1.y);
            const float2 P2 = *(const float2 *)((const char *)sPair + orow + c2.y);
            const bool rowok = (unsigned)i <= (unsigned)I;
            {   // even diagonal: cell (i, j0).  left = the left neighbour lane's odd cell, up = own odd cell, diagonal = own even cell
                float L = wave_shr1_z(O); if (first) L = 0.0f;
                float g = (E * P0.x) + (L * __int_as_float(c0.x));
                if ((i | j0) == 0) g = 1.0f;
                const float a = g + O * P1.y;
                if (hasE && rowok && (unsigned)j0 <= (unsigned)J) { E = a; gp[0] = g; }
            }
            {   // odd diagonal: cell (i, j0 + 1).  left = own even cell (new), up = the right neighbour lane's even cell (new), diagonal = own odd cell
                float U = wave_shl1_z(E); if (last) U = 0.0f;
                float g = (O * P1.x) + (E * __int_as_float(c1.x));
                if ((i | (j0 + 1)) == 0) g = 1.0f;
                const float a = g + U * P2.y;
                if (hasO && rowok && (unsigned)(j0 + 1) <= (unsigned)J) { O = a; gp[1] = g; }
            }
            c0 = c1; c1 = c2;
        }
        __syncthreads();
    }
    // results: gamma of every band cell, in a full-matrix layout for the check
    for (int r = 0; r < tk.nread; ++r) {
        const int Ir = sI[r]; int lo, hi; band_of(Ir, &lo, &hi); const int bw = hi - lo + 1;
        for (int e = lane; e < (Ir + 1) * (J + 1); e += 64) {
            const int ii = e / (J + 1), jj = e % (J + 1), d = jj - ii;
            out[(size_t)tk.read[r] * 64 * 32 + ii * 32 + jj] = (d >= lo && d <= hi) ? sG[r][ii * (bw - 1) + (jj - lo)] : 0.0f;
        }
    }
}

// ---- kernel A: lane = read row, two reads per wave (rows 0..31 each), anti-diagonal sweep over the full matrix, gamma with row stride 28
__global__ __launch_bounds__(64) void k_rows(const Tables *T, const unsigned char *ctx, const ReadIn *reads, const int2 *pairs, float *out, int reps)
{
    __shared__ float2 sPair[(NOBS + 1) * CTXS];
    __shared__ int2 sCol[J + 2 * CG + 1];
    __shared__ unsigned char sObs[RPW][OBSROW];
    __shared__ int sI[RPW];
    __shared__ float sG[2][GMAX_A];
    __shared__ float sPad[RPW * GMAX_B - 2 * GMAX_A];                     // (the same LDS footprint, hence the same waves per SIMD, as kernel B)
    const int lane = threadIdx.x, half = lane >> 5, row = lane & 31;
    int rid[RPW] = {pairs[blockIdx.x].x, pairs[blockIdx.x].y, 0, 0, 0, 0, 0, 0};
    stage(T, ctx, reads, rid, 2, sPair, sCol, sObs, sI);
    const int I = sI[half];
    const bool rowok = row <= I;
    const int orow = (int)sObs[half][OG + row - 1] * (CTXS * 8);             // o_{i-1}: fixed per lane
    const int Tmax = (sI[0] > sI[1] ? sI[0] : sI[1]) + J;
    for (int rep = 0; rep < reps; ++rep) {
        float acur = row == 0 ? 1.0f : 0.0f, updiag = 0.0f, mePrev = 0.0f, dlPrev = 1.0f;
        const int2 *cp = &sCol[CG - row];                                    // cp[t] = entry of column t - row
        float *gp = &sG[half][row * 28 - row];
        for (int t = 0; t <= Tmax; ++t) {
            const float up = wave_shr1_z(acur);
            const int2 c = cp[t];
            const float2 P = *(const float2 *)((const char *)sPair + orow + c.y);
            if (rowok && (unsigned)(t - row) <= (unsigned)J) {
                const float g = (updiag * mePrev) + (acur * dlPrev);
                gp[t] = g;
                acur = g + up * P.y;
                mePrev = P.x; dlPrev = __int_as_float(c.x);
            }
            updiag = up;
        }
        __syncthreads();
    }
    if (reps < 0) sPad[lane] = 0.0f;                                         // (keeps the padding allocated)
    for (int r = 0; r < 2; ++r) {
        const int Ir = sI[r];
        for (int e = lane; e < (Ir + 1) * (J + 1); e += 64) { const int ii = e / (J + 1), jj = e % (J + 1); out[(size_t)rid[r] * 64 * 32 + ii * 32 + jj] = sG[r][ii * 28 + jj]; }
    }
}

// column-order reference (the oracle's fill(), alpha part), banded or full
static void ref_fill(const Tables &T, const unsigned char *ctx, const ReadIn &rd, bool banded, std::vector<float> &gam)
{
    const int I = rd.I; int dlo = -1000, dhi = 1000; if (banded) band_of(I, &dlo, &dhi);
    std::vector<float> alp(64 * 32, 0.0f); gam.assign(64 * 32, 0.0f);
    for (int j = 0; j <= J; ++j) for (int i = 0; i <= I; ++i) {
        if (j - i < dlo || j - i > dhi) continue;
        float g;
        if (j == 0) g = i == 0 ? 1.0f : 0.0f;
        else {
            const float m = i > 0 ? alp[(i - 1) * 32 + j - 1] * T.pair[rd.obs[i - 1] * CTXS + ctx[j - 1]].x : 0.0f;
            const float dl = alp[i * 32 + j - 1] * T.dl[ctx[j - 1]];
            g = m + dl;
        }
        gam[i * 32 + j] = g;
        const float st = (i > 0 && j < J) ? alp[(i - 1) * 32 + j] * T.pair[rd.obs[i - 1] * CTXS + ctx[j]].y : 0.0f;
        alp[i * 32 + j] = g + st;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int nwin = argc > 1 ? atoi(argv[1]) : 32768, reps = argc > 2 ? atoi(argv[2]) : 8;
    srand(12345);
    auto rnd = [] { return (float)(rand() & 0xffff) / 65536.0f; };
    Tables T; memset(&T, 0, sizeof(T));
    for (int o = 0; o < NOBS; ++o) for (int k = 0; k < 32; ++k) T.pair[o * CTXS + k] = make_float2(0.05f + 0.6f * rnd(), 0.01f + 0.1f * rnd());
    for (int k = 0; k < 16; ++k) T.dl[k] = 0.01f + 0.1f * rnd();
    unsigned char ctx[J]; for (int j = 0; j < J; ++j) ctx[j] = rand() & 15;
    const int nreads = nwin * RPW;
    std::vector<ReadIn> reads(nreads);
    for (auto &r : reads) { r.I = J - 3 + rand() % 7; for (int i = 0; i < 64; ++i) r.obs[i] = (unsigned char)(rand() % NOBS); }
    // tasks of kernel B: the reads of a window packed into waves of <= 64 lanes
    std::vector<Task> tasks;
    double lanes_used = 0;
    {   // (all reads of this benchmark share one window template, so waves are filled across windows: as many reads as fit 64 lanes, at most RPW)
        Task t; memset(&t, 0, sizeof(t)); int nl = 0; t.hlo = 1 << 20; t.hhi = -1;
        auto flush = [&] { if (t.nread) { tasks.push_back(t); lanes_used += nl; } memset(&t, 0, sizeof(t)); nl = 0; t.hlo = 1 << 20; t.hhi = -1; };
        for (int r = 0; r < nreads; ++r) {
            int dlo, dhi; band_of(reads[r].I, &dlo, &dhi);
            const int BW = dhi - dlo + 1, L = (BW + 1) / 2;
            if (nl + L > 64 || t.nread == RPW) flush();
            t.read[t.nread] = r; t.lane0[t.nread] = nl; ++t.nread; nl += L;
            for (int m = 0; m < L; ++m) {
                const bool hasO = 2 * m + 1 <= BW - 1;                      // the lane's odd cell (i, j0 + 1) enters the window one iteration before its even one
                const int lo = std::max(m, -dlo - m - (hasO ? 1 : 0)), hi = std::min(reads[r].I + m, J - dlo - m);
                t.hlo = std::min(t.hlo, lo); t.hhi = std::max(t.hhi, hi);
            }
        }
        flush();
    }
    std::vector<int2> pairs(nreads / 2); for (int p = 0; p < nreads / 2; ++p) pairs[p] = make_int2(2 * p, 2 * p + 1);
    Tables *dT; unsigned char *dctx; ReadIn *dreads; Task *dtasks; int2 *dpairs; float *dout;
    CK(hipMalloc(&dT, sizeof(T))); CK(hipMalloc(&dctx, J)); CK(hipMalloc(&dreads, sizeof(ReadIn) * nreads)); CK(hipMalloc(&dtasks, sizeof(Task) * tasks.size()));
    CK(hipMalloc(&dpairs, sizeof(int2) * pairs.size())); CK(hipMalloc(&dout, sizeof(float) * 64 * 32 * (size_t)nreads));
    CK(hipMemcpy(dT, &T, sizeof(T), hipMemcpyHostToDevice)); CK(hipMemcpy(dctx, ctx, J, hipMemcpyHostToDevice));
    CK(hipMemcpy(dreads, reads.data(), sizeof(ReadIn) * nreads, hipMemcpyHostToDevice)); CK(hipMemcpy(dtasks, tasks.data(), sizeof(Task) * tasks.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dpairs, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> out((size_t)64 * 32 * nreads), gref;
    for (int which = 0; which < 2; ++which) {
        float ms1 = 0, msR = 0;
        for (int pass = 0; pass < 3; ++pass) {                                // warm-up, 1 repetition, `reps` repetitions: the difference is the sweeps alone
            const int rp = pass == 2 ? reps : 1;
            CK(hipMemset(dout, 0, sizeof(float) * 64 * 32 * (size_t)nreads));
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_stair, dim3((unsigned)tasks.size()), dim3(64), 0, 0, dT, dctx, dreads, dtasks, dout, rp);
            else hipLaunchKernelGGL(k_rows, dim3((unsigned)pairs.size()), dim3(64), 0, 0, dT, dctx, dreads, dpairs, dout, rp);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (pass == 1) ms1 = ms; if (pass == 2) msR = ms;
        }
        CK(hipMemcpy(out.data(), dout, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
        long bad = 0; const int ncheck = std::min(nreads, 4096);
        for (int r = 0; r < ncheck; ++r) {
            ref_fill(T, ctx, reads[r], which == 0, gref);
            for (int i = 0; i <= reads[r].I; ++i) for (int j = 0; j <= J; ++j) {
                float a = out[(size_t)r * 64 * 32 + i * 32 + j], b = gref[i * 32 + j];
                if (memcmp(&a, &b, 4) != 0) { if (bad < 5) printf("  mismatch read %d (I %d) cell (%d, %d): %g vs %g\n", r, reads[r].I, i, j, a, b); ++bad; }
            }
        }
        const double per = (msR - ms1) * 1e6 / ((double)(reps - 1) * nreads);
        printf("%s: %zu waves for %d reads (%.1f reads per wave%s), mismatching cells in %d checked reads: %ld; %.3f ms for 1 sweep set, %.3f ms for %d -> %.2f ns per (read, window) alpha sweep\n",
               which == 0 ? "staircase (banded, 2 diagonals per lane)" : "rows      (full matrix, lane = row)     ", which == 0 ? tasks.size() : pairs.size(), nreads,
               (double)nreads / (which == 0 ? tasks.size() : pairs.size()), which == 0 ? "" : "", ncheck, bad, ms1, msR, reps, per);
        if (which == 0) printf("   lanes used per staircase wave: %.1f of 64\n", lanes_used / tasks.size());
    }
    return 0;
}
