// valu_peak.hip — VALU issue-rate calibration for gfx950.
// Question: how many cycles does one wave64 VALU instruction occupy a SIMD's issue port?  (4 as on a SIMD-16, or 2 as
// /opt/skills/guides/MI355X_MICROARCH.md states for CDNA4's SIMD-32.)  Each wave runs a loop of 64 independent-chain
// instructions of one class (8 accumulators, so the dependent latency never limits issue); grids put 1, 2, 4 or 8 waves
// on every SIMD.  Reported: SIMD cycles per wave-instruction = (s_memtime cycles of the slowest wave) / (instructions per
// wave x waves per SIMD), and the chip-wide rate from HIP events.
//   build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak/valu_peak.hip -o tools/valu_peak/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { OP_FMA, OP_ADD, OP_MUL, OP_MAXI, OP_ADDU, OP_CNDMASK, OP_MAXI_DPP, OP_MOV_DPP, OP_PKFMA, OP_READLANE, OP_LSHLADD, OP_MOV, OP_CMP, OP_DSREAD, OP_CNDMASK_S, OP_CMP_CND, OP_MED3, OP_MINMAX, OP_FMAC, OP_PKMUL, OP_PKADD, OP_FMA3, OP_ADD_E64, OP_MUL_DPP, OP_FMAC_DPP, OP_DEP_FMA, OP_DEP_MULADD, OP_DEP_FMAC_DPP, OP_DEP_MULADD_DPP, NOPS };
static const char *opname[NOPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_i32", "v_add_u32", "v_cndmask_b32", "v_max_i32_dpp row_shr:1",
                                   "v_mov_b32_dpp wave_shr:1", "v_pk_fma_f32", "v_readlane_b32 (to SGPR)", "v_lshl_add_u32", "v_mov_b32", "v_cmp_gt_u32 (to vcc)", "ds_read_b32 (bank-conflict free)", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cmp_gt_u32 + v_cndmask_b32 (pair)", "v_med3_i32", "v_min_i32 + v_max_i32 (pair)",
                                   "v_fmac_f32 (VOP2)", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32 (three distinct sources)", "v_add_f32_e64 (VOP3 encoding)", "v_mul_f32_dpp row_ror:1", "v_fmac_f32_dpp row_ror:1",
                                   "DEPENDENT chain: v_fma_f32 (1 per step)", "DEPENDENT chain: v_mul_f32 + v_add_f32 (2 per step)", "DEPENDENT chain: v_fmac_f32_dpp row_ror:1 (1 per step)", "DEPENDENT chain: v_mov_dpp + v_mul + v_add (3 per step)"};

template <int OP> __global__ void k(unsigned long long *cyc, float *sink, float seed)
{
    float a[8]; int b[8];
    float2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = (int)threadIdx.x * 7 + i; p[i] = make_float2(a[i], a[i] + 1.f); }
    const float c = seed * 0.5f;
    int sacc = 0;
    __shared__ int lds[256];
    lds[threadIdx.x & 255] = (int)threadIdx.x;
    const int ldsaddr = (int)(threadIdx.x & 63) * 4;
    const unsigned long long smask = __ballot((int)(threadIdx.x & 1));
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b[0]), "v"(b[1]) : "vcc");       // vcc defined for the v_cndmask test
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define ONE(i)                                                                                                         \
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));                                    \
    else if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                   \
    else if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                   \
    else if (OP == OP_MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));                       \
    else if (OP == OP_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));                       \
    else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]) : );    \
    else if (OP == OP_MAXI_DPP) asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b[i])); \
    else if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b[i]) : "v"(b[(i + 1) & 7])); \
    else if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));                \
    else if (OP == OP_READLANE) { int s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(b[i])); sacc += s_; }        \
    else if (OP == OP_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));              \
    else if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7]));                               \
    else if (OP == OP_CMP) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc");               \
    else if (OP == OP_DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(b[i]) : "v"(ldsaddr));                                \
    else if (OP == OP_CNDMASK_S) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "s"(smask)); \
    else if (OP == OP_CMP_CND) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]) : "vcc"); \
    else if (OP == OP_MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]));   \
    else if (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(a[(i + 1) & 7]));                  \
    else if (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));                      \
    else if (OP == OP_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));                      \
    else if (OP == OP_FMA3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(a[(i + 1) & 7]));               \
    else if (OP == OP_ADD_E64) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                \
    else if (OP == OP_MUL_DPP) asm volatile("v_mul_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(c)); \
    else if (OP == OP_FMAC_DPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(c)); \
    else if (OP == OP_DEP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c), "v"(a[1]));                      \
    else if (OP == OP_DEP_MULADD) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[0]) : "v"(c), "v"(a[1])); \
    else if (OP == OP_DEP_FMAC_DPP) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(c)); \
    else if (OP == OP_DEP_MULADD_DPP) asm volatile("v_mov_b32_dpp %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %1" : "+v"(a[0]), "+v"(a[2]) : "v"(c)); \
    else if (OP == OP_MINMAX) asm volatile("v_min_i32 %0, %0, %1\n\tv_max_i32 %0, %0, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]));
            REP8(ONE)
#undef ONE
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (OP == OP_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0.f; int q = sacc + lds[0];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += a[i] + p[i].x + p[i].y; q += b[i]; }
    if (s == 12345.678f || q == 0x7fffffff) sink[0] = s + q;     // keep everything live
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> static void run(int waves_per_simd, int ncu, double clk_ghz)
{
    // one workgroup of 256 threads = 1 wave per SIMD of a CU; waves_per_simd workgroups per CU
    const int grid = ncu * waves_per_simd, nw = grid * 4;
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, nw * 8); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, cyc, sink, 1.0f);      // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, cyc, sink, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ninst = (double)ITERS * 64;
    const double med = (double)h[nw / 2], mx = (double)h[nw - 1];
    // s_memtime ticks at a constant 100 MHz-derived rate on some parts; report both the tick-based and the wall-based figure
    const double wall_cyc_per_inst_simd = (ms * 1e-3 * clk_ghz * 1e9) / (ninst * waves_per_simd);
    printf("%-28s waves/SIMD %d  memtime ticks/inst/wave med %.3f max %.3f  | wall: %.3f ms -> %.3f SIMD-cycles per wave-instruction at %.2f GHz, chip %.2f T lane-ops/s\n",
           opname[OP], waves_per_simd, med / ninst, mx / ninst, ms, wall_cyc_per_inst_simd, clk_ghz,
           ninst * nw * 64 / (ms * 1e-3) / 1e12);
    hipFree(cyc); hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e-6;   // kHz -> GHz
    printf("device %s, %d CUs, clockRate %.3f GHz (nominal; the sustained clock under load is lower)\n", p.gcnArchName, p.multiProcessorCount, clk);
    for (int w : {1, 2, 4, 8}) {
        run<OP_FMA>(w, p.multiProcessorCount, clk);
        run<OP_ADD>(w, p.multiProcessorCount, clk);
        run<OP_MUL>(w, p.multiProcessorCount, clk);
        run<OP_MAXI>(w, p.multiProcessorCount, clk);
        run<OP_ADDU>(w, p.multiProcessorCount, clk);
        run<OP_CNDMASK>(w, p.multiProcessorCount, clk);
        run<OP_MAXI_DPP>(w, p.multiProcessorCount, clk);
        run<OP_MOV_DPP>(w, p.multiProcessorCount, clk);
        run<OP_PKFMA>(w, p.multiProcessorCount, clk);
        run<OP_READLANE>(w, p.multiProcessorCount, clk);
        run<OP_LSHLADD>(w, p.multiProcessorCount, clk);
        run<OP_MOV>(w, p.multiProcessorCount, clk);
        run<OP_CMP>(w, p.multiProcessorCount, clk);
        run<OP_DSREAD>(w, p.multiProcessorCount, clk);
        run<OP_CNDMASK_S>(w, p.multiProcessorCount, clk);
        run<OP_CMP_CND>(w, p.multiProcessorCount, clk);
        run<OP_MED3>(w, p.multiProcessorCount, clk);
        run<OP_MINMAX>(w, p.multiProcessorCount, clk);
        run<OP_FMAC>(w, p.multiProcessorCount, clk);
        run<OP_PKMUL>(w, p.multiProcessorCount, clk);
        run<OP_PKADD>(w, p.multiProcessorCount, clk);
        run<OP_FMA3>(w, p.multiProcessorCount, clk);
        run<OP_ADD_E64>(w, p.multiProcessorCount, clk);
        run<OP_MUL_DPP>(w, p.multiProcessorCount, clk);
        run<OP_FMAC_DPP>(w, p.multiProcessorCount, clk);
        run<OP_DEP_FMA>(w, p.multiProcessorCount, clk);
        run<OP_DEP_MULADD>(w, p.multiProcessorCount, clk);
        run<OP_DEP_FMAC_DPP>(w, p.multiProcessorCount, clk);
        run<OP_DEP_MULADD_DPP>(w, p.multiProcessorCount, clk);
    }
    return 0;
}
