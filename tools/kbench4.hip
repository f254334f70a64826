// Do 16-byte global loads/stores at arbitrary byte alignment work on gfx950, and what do they cost?  (rows of odd-sized
// grids -- 355, 437, 123 voxels -- start at arbitrary byte offsets.)  Development tool.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint32_t u32; typedef int64_t i64; typedef uint8_t u8;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

// row copy: rows of `rowlen` bytes (any value), 16-byte pieces per lane at whatever alignment the row start has
__global__ __launch_bounds__(256) void k_rows(const u8* __restrict__ in, u8* __restrict__ out, i64 nrows, i64 rowlen, i64 ia, i64 oa) {
    const i64 pieces = rowlen / 16;       // whole pieces only (tails ignored in the timing kernel)
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < nrows * pieces; t += (i64)gridDim.x * blockDim.x) {
        const i64 r = t / pieces, p = t - r * pieces;
        const u32x4 v = ((const U4*)(in + ia + r * rowlen + 16 * p))->v;
        ((U4*)(out + oa + r * rowlen + 16 * p))->v = v;
    }
}

int main() {
    const i64 N = 1ll << 30;
    u8 *in, *out;
    CK(hipMalloc(&in, N + 4096)); CK(hipMalloc(&out, N + 4096));
    std::vector<u8> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (u8)(i * 131 + (i >> 8));
    for (i64 o = 0; o < N; o += (i64)h.size()) CK(hipMemcpy(in + o, h.data(), h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const i64 rowlens[] = {1024, 1040, 355, 437, 123, 1000};
    for (i64 rl : rowlens) {
        for (int ia = 0; ia < 2; ++ia) {
            const i64 nrows = (N - 64) / rl;
            CK(hipMemset(out, 0, N));
            hipLaunchKernelGGL(k_rows, dim3(8192), dim3(256), 0, 0, in, out, nrows, rl, (i64)ia, (i64)(ia * 3));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_rows, dim3(8192), dim3(256), 0, 0, in, out, nrows, rl, (i64)ia, (i64)(ia * 3));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            // verify a window
            std::vector<u8> a(1 << 16), b(1 << 16);
            const i64 r0 = nrows / 3;
            CK(hipMemcpy(a.data(), in + ia + r0 * rl, a.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), out + ia * 3 + r0 * rl, b.size(), hipMemcpyDeviceToHost));
            size_t bad = 0; const i64 pieces = rl / 16;
            for (size_t i = 0; i < a.size(); ++i) { const i64 inrow = (i64)i % rl; if (inrow < pieces * 16 && a[i] != b[i]) ++bad; }
            const double bytes = 2.0 * nrows * pieces * 16;
            printf("rowlen %5lld in+%d out+%d : %.4f ms %.0f GB/s  mismatches %zu\n", (long long)rl, ia, ia * 3, ms, bytes / ms / 1e6, bad);
        }
    }
    return 0;
}
