#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "wave_ops.h"
__global__ void k(const int* in, int* out)
{
    int l = threadIdx.x; int v = in[blockIdx.x * 64 + l];
    int* o = out + blockIdx.x * 64 * 6;
    o[l] = wave_shr1_i32(v, -7); o[64 + l] = wave_shl1_i32(v, -9);
    o[128 + l] = wave_scan_max_i32(v); o[192 + l] = wave_scan_add_i32(v);
    o[256 + l] = wave_reduce_max_i32(v); o[320 + l] = wave_reduce_add_i32(v);
}
int main()
{
    const int B = 64; int h[B * 64], r[B * 64 * 6];
    srand(1); for (int i = 0; i < B * 64; ++i) h[i] = (rand() % 2001) - 1000;
    int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, o); hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < B; ++b) {
        const int* v = h + b * 64; const int* q = r + b * 384; int mx = -1 << 30, sm = 0;
        for (int l = 0; l < 64; ++l) {
            mx = v[l] > mx ? v[l] : mx; sm += v[l];
            bad += q[l] != (l ? v[l - 1] : -7); bad += q[64 + l] != (l < 63 ? v[l + 1] : -9);
            bad += q[128 + l] != mx; bad += q[192 + l] != sm;
        }
        for (int l = 0; l < 64; ++l) { bad += q[256 + l] != mx; bad += q[320 + l] != sm; }
    }
    printf("wave_ops mismatches: %d\n", bad); return bad != 0;
}
// (pair scan is validated end-to-end by the alignment parity tests)
