// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the ccsx kernels use
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").  Each kernel streams a
// 2 GiB buffer (8x the 256 MiB Infinity Cache) once: rd1 / rd4 / rd16 read 1, 4, 16 bytes per lane, wr1 / wr4 write.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/calib_fetch.hip -o tools/calib/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void rd1(const uint8_t *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *out = acc;
}
__global__ void rd4(const uint32_t *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *out = acc;
}
__global__ void rd16(const uint4 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void wr1(uint8_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)i;
}
__global__ void wr4(uint32_t *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
// one wave writes 256 contiguous bytes (64 x int32) every `stride` bytes, like a k_poa score column
__global__ void wr_rows(uint32_t *p, size_t rows)
{
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) p[r * 64 + threadIdx.x] = (uint32_t)r;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    void *buf; unsigned *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 1, bytes));
    const dim3 g(256 * 32), b(256);
    hipLaunchKernelGGL(rd1, g, b, 0, 0, (const uint8_t *)buf, bytes / 4, out);        // 512 MiB with byte loads
    hipLaunchKernelGGL(rd4, g, b, 0, 0, (const uint32_t *)buf, bytes / 4, out);       // 2 GiB
    hipLaunchKernelGGL(rd16, g, b, 0, 0, (const uint4 *)buf, bytes / 16, out);        // 2 GiB
    hipLaunchKernelGGL(wr1, g, b, 0, 0, (uint8_t *)buf, bytes / 4);                   // 512 MiB
    hipLaunchKernelGGL(wr4, g, b, 0, 0, (uint32_t *)buf, bytes / 4);                  // 2 GiB
    hipLaunchKernelGGL(wr_rows, dim3(256 * 32), dim3(64), 0, 0, (uint32_t *)buf, bytes / 256);   // 2 GiB in 256-byte rows
    CK(hipDeviceSynchronize());
    std::printf("known bytes: rd1 %zu rd4 %zu rd16 %zu wr1 %zu wr4 %zu wr_rows %zu\n", bytes / 4, bytes, bytes, bytes / 4, bytes, bytes);
    return 0;
}
