// wave_ops.h — wave64 cross-lane primitives for gfx950 built on DPP (no LDS traffic, 1 VALU op each).
// DPP control codes (GFX9): row_shr:n = 0x110+n, wave_shl:1 = 0x130, wave_shr:1 = 0x138,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.
#pragma once
#include <hip/hip_runtime.h>

#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// lane l <- x[l-1]; lane 0 <- fill
__device__ __forceinline__ int wave_shr1_i32(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, DPP_WAVE_SHR1, 0xf, 0xf, false); }
// lane l <- x[l+1]; lane 63 <- fill
__device__ __forceinline__ int wave_shl1_i32(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, DPP_WAVE_SHL1, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_shr1_f32(float x, float fill) { return __int_as_float(wave_shr1_i32(__float_as_int(x), __float_as_int(fill))); }
__device__ __forceinline__ float wave_shl1_f32(float x, float fill) { return __int_as_float(wave_shl1_i32(__float_as_int(x), __float_as_int(fill))); }

// the same shifts with a zero fill: bound_ctrl:1 makes the hardware supply the 0, no register has to be pre-loaded with it
__device__ __forceinline__ float wave_shr1_f32_z(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), DPP_WAVE_SHR1, 0xf, 0xf, true)); }
__device__ __forceinline__ float wave_shl1_f32_z(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), DPP_WAVE_SHL1, 0xf, 0xf, true)); }

// rotation by one lane inside each 16-lane DPP row (row_ror:n = 0x120 + n): lane l <- x[(l - 1) & 15] / x[(l + 1) & 15] of its row
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ float row_ror1_f32(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), DPP_ROW_ROR(1), 0xf, 0xf, false)); }
__device__ __forceinline__ float row_rol1_f32(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), DPP_ROW_ROR(15), 0xf, 0xf, false)); }

__device__ __forceinline__ int wave_shr1_i32_z(int x) { return __builtin_amdgcn_mov_dpp(x, DPP_WAVE_SHR1, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_shl1_i32_z(int x) { return __builtin_amdgcn_mov_dpp(x, DPP_WAVE_SHL1, 0xf, 0xf, true); }

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// inclusive prefix maximum over the 64 lanes.  `old` = INT_MIN (the identity of max) lets the DPP combiner
// fold every move into its v_max_i32_dpp.
#define WAVE_IMIN ((int)0x80000000)
__device__ __forceinline__ int wave_scan_max_i32(int v)
{
    // Hillis-Steele inside each 16-lane row (every step reads the running value, so each one is a single fused
    // v_max_i32_dpp), then the two cross-row broadcasts: 6 VALU ops.
    int s = imax(v, __builtin_amdgcn_update_dpp(WAVE_IMIN, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(2), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(4), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(8), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_BCAST15, 0xa, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_BCAST31, 0xc, 0xf, false));
    return s;
}
// inclusive prefix maximum inside each 32-lane half (two independent scans per wave): the same ladder without the last
// cross-half broadcast, 5 VALU ops
__device__ __forceinline__ int half_scan_max_i32(int v)
{
    int s = imax(v, __builtin_amdgcn_update_dpp(WAVE_IMIN, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(2), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(4), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(8), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_BCAST15, 0xa, 0xf, false));
    return s;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_scan_add_i32(int v)
{
    int s = v + __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
    s = s + __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, false);
    s = s + __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(3), 0xf, 0xf, false);
    s = s + __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR(4), 0xf, 0xe, false);
    s = s + __builtin_amdgcn_update_dpp(0, s, DPP_ROW_SHR(8), 0xf, 0xc, false);
    s = s + __builtin_amdgcn_update_dpp(0, s, DPP_ROW_BCAST15, 0xa, 0xf, false);
    s = s + __builtin_amdgcn_update_dpp(0, s, DPP_ROW_BCAST31, 0xc, 0xf, false);
    return s;
}
// ---- the same primitives inside each 16-lane DPP row (four independent groups per wave: k_align16's four 16-row bands)
#define DPP_ROW_SHL(n) (0x100 + (n))
__device__ __forceinline__ int row_shr1_i32(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, DPP_ROW_SHR(1), 0xf, 0xf, false); }
__device__ __forceinline__ int row_shl1_i32(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, DPP_ROW_SHL(1), 0xf, 0xf, false); }
__device__ __forceinline__ int row_shl2_i32(int x, int fill) { return __builtin_amdgcn_update_dpp(fill, x, DPP_ROW_SHL(2), 0xf, 0xf, false); }
__device__ __forceinline__ int row_shr1_i32_z(int x) { return __builtin_amdgcn_mov_dpp(x, DPP_ROW_SHR(1), 0xf, 0xf, true); }
__device__ __forceinline__ int row_shl1_i32_z(int x) { return __builtin_amdgcn_mov_dpp(x, DPP_ROW_SHL(1), 0xf, 0xf, true); }
__device__ __forceinline__ int row_shl2_i32_z(int x) { return __builtin_amdgcn_mov_dpp(x, DPP_ROW_SHL(2), 0xf, 0xf, true); }
// inclusive prefix maximum inside each 16-lane row: 4 VALU ops
__device__ __forceinline__ int row_scan_max_i32(int v)
{
    int s = imax(v, __builtin_amdgcn_update_dpp(WAVE_IMIN, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(2), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(4), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_SHR(8), 0xf, 0xf, false));
    return s;
}
// maximum over each 16-lane row, in every lane of the row: four rotations (no broadcast afterwards, nothing through the LDS pipe)
__device__ __forceinline__ int row_allmax_i32(int v)
{
    int s = imax(v, __builtin_amdgcn_update_dpp(WAVE_IMIN, v, DPP_ROW_ROR(1), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_ROR(2), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_ROR(4), 0xf, 0xf, false));
    s = imax(s, __builtin_amdgcn_update_dpp(WAVE_IMIN, s, DPP_ROW_ROR(8), 0xf, 0xf, false));
    return s;
}

// wave-wide maximum, returned uniformly (scan + readlane 63)
__device__ __forceinline__ int wave_reduce_max_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_max_i32(v), 63); }
__device__ __forceinline__ int wave_reduce_add_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_add_i32(v), 63); }

// inclusive prefix "arg-max" over the 64 lanes for pairs (d, o): the pair of a lower lane replaces the running
// pair only if its d is strictly greater (ties keep the higher lane), o rides along.  7 steps, DPP only.
#define WAVE_PAIR_STEP(SRC_D, SRC_O, CTRL, RM, BM)                                                    \
    {                                                                                                 \
        const int td = __builtin_amdgcn_update_dpp(d, SRC_D, CTRL, RM, BM, false);                    \
        const int to = __builtin_amdgcn_update_dpp(o, SRC_O, CTRL, RM, BM, false);                    \
        const bool take = td > d;                                                                     \
        d = take ? td : d; o = take ? to : o;                                                         \
    }
__device__ __forceinline__ void wave_scan_max_pair(int &d, int &o)
{
    const int d0 = d, o0 = o;
    WAVE_PAIR_STEP(d0, o0, DPP_ROW_SHR(1), 0xf, 0xf)
    WAVE_PAIR_STEP(d0, o0, DPP_ROW_SHR(2), 0xf, 0xf)
    WAVE_PAIR_STEP(d0, o0, DPP_ROW_SHR(3), 0xf, 0xf)
    WAVE_PAIR_STEP(d, o, DPP_ROW_SHR(4), 0xf, 0xe)
    WAVE_PAIR_STEP(d, o, DPP_ROW_SHR(8), 0xf, 0xc)
    WAVE_PAIR_STEP(d, o, DPP_ROW_BCAST15, 0xa, 0xf)
    WAVE_PAIR_STEP(d, o, DPP_ROW_BCAST31, 0xc, 0xf)
}
