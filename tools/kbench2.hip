// K1 block-tile variants: block size, columns per tile, grid size, unroll, input data pattern.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64; typedef uint8_t u8;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int BS, int UNROLL, int TC, bool NTL, bool NTS>
__global__ __launch_bounds__(BS) void k_carve_block(const u32x4* __restrict__ in, u32x4* __restrict__ out, const u8* __restrict__ mask,
                                                    i64 ncols, u32 vpc, u32 magic, i64 ntiles) {
    __shared__ u8 smask[TC];
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 col0 = t * TC;
        const int c_here = (int)(ncols - col0 < TC ? ncols - col0 : TC);
        if (t != blockIdx.x) __syncthreads();
        for (int c = threadIdx.x; c < TC; c += BS) smask[c] = c < c_here ? mask[col0 + c] : 0;
        __syncthreads();
        const u32 nvec = (u32)c_here * vpc;
        const u32x4* src = in + col0 * vpc; u32x4* dst = out + col0 * vpc;
        for (u32 v0 = 0; v0 < nvec; v0 += BS * UNROLL) {
            u32x4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const u32 vv = v0 + BS * u + threadIdx.x;
                x[u] = (u32x4)(0u);
                if (vv < nvec) { const u32 c = __umulhi(vv, magic); if (smask[c]) x[u] = ld<NTL>(src + vv); }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { const u32 vv = v0 + BS * u + threadIdx.x; if (vv < nvec) st<NTS>(dst + vv, x[u]); }
        }
    }
}
template <int UNROLL>
__global__ __launch_bounds__(1024) void k_carve_rt(const u32x4* __restrict__ in, u32x4* __restrict__ out, const u8* __restrict__ mask,
                                                   i64 ncols, u32 vpc, u32 magic, int TC) {
    extern __shared__ u8 dyn[];
    __shared__ u8 smask[64];
    const int BS = blockDim.x;
    const i64 col0 = (i64)blockIdx.x * TC;
    const int c_here = (int)(ncols - col0 < TC ? ncols - col0 : TC);
    if ((int)threadIdx.x < TC) smask[threadIdx.x] = (int)threadIdx.x < c_here ? mask[col0 + threadIdx.x] : 0;
    __syncthreads();
    const u32 nvec = (u32)c_here * vpc;
    const u32x4* src = in + col0 * vpc; u32x4* dst = out + col0 * vpc;
    u32x4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const u32 vv = BS * u + threadIdx.x;
        x[u] = (u32x4)(0u);
        if (vv < nvec) { const u32 c = __umulhi(vv, magic); if (smask[c]) x[u] = ld<true>(src + vv); }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { const u32 vv = BS * u + threadIdx.x; if (vv < nvec) st<true>(dst + vv, x[u]); }
    if (dyn[0] == 77 && threadIdx.x == 1023000) out[0] = x[0];
}
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, i64 nvec) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        u32x4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = ld<NTL>(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) st<NTS>(out + i + u * stride, x[u]);
    }
    for (; i < nvec; i += stride) st<NTS>(out + i, ld<NTL>(in + i));
}
__device__ inline u64 splitmix64(u64 z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__global__ void k_fill(u8* p, i64 n, int mode) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        u8 v;
        if (mode == 0) v = (u8)splitmix64(i);
        else if (mode == 1) { const int lab = (int)(splitmix64(1 ^ (u64)(i / 3)) & 15); v = (u8)(lab * 16 + (i % 3) * 5 + 7); }
        else v = 0;
        p[i] = v;
    }
}
int main(int argc, char** argv) {
    const i64 S = 1024;
    const i64 ncols = S * S, col = S * 3, nbytes = ncols * col, nvec = nbytes / 16;
    const u32 vpc = (u32)(col / 16);
    const u32 magic = (u32)(((1ull << 32) + vpc - 1) / vpc);
    u32x4 *in, *out; u8* mask;
    CK(hipMalloc(&in, nbytes)); CK(hipMalloc(&out, nbytes)); CK(hipMalloc(&mask, ncols));
    std::vector<u8> hm(ncols);
    for (i64 x = 0; x < S; ++x) for (i64 y = 0; y < S; ++y) {
        i64 xn = x, yn = y, dx2 = 2 * xn - 1023, adx2 = dx2 < 0 ? -dx2 : dx2, dy2 = 2 * (yn - 256);
        bool body = adx2 < 840 && yn >= 256, dome = dx2 * dx2 * 40000 + dy2 * dy2 * 90000 < 4ll * 90000 * 40000, tow = adx2 > 880 && adx2 < 960 && yn >= 96;
        hm[x * S + y] = body || dome || tow;
    }
    CK(hipMemcpy(mask, hm.data(), ncols, hipMemcpyHostToDevice));
    hipStream_t stream; CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipStreamSynchronize(stream));
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, stream));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, stream)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps; if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-52s %.4f ms  alg %.0f GB/s\n", name, best, 2.0 * nbytes / (best * 1e-3) / 1e9);
    };
#define RUN(BS, UN, TC, NTL, NTS, GRID) do { char nm[128]; const i64 nt = (ncols + TC - 1) / TC; const i64 g = (GRID) < nt ? (GRID) : nt; \
    snprintf(nm, 128, "block bs=%d un=%d tc=%d nt=%d%d grid=%lld", BS, UN, TC, NTL, NTS, (long long)g); \
    timeit(nm, [&] { hipLaunchKernelGGL((k_carve_block<BS, UN, TC, NTL, NTS>), dim3((unsigned)g), dim3(BS), 0, stream, in, out, mask, ncols, vpc, magic, nt); }); } while (0)
    std::vector<u8> ones(ncols, 1);
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, stream, (u8*)in, nbytes, 1);
    CK(hipStreamSynchronize(stream));
    for (int mm = 0; mm < 1; ++mm) {
        CK(hipMemcpy(mask, mm == 0 ? hm.data() : ones.data(), ncols, hipMemcpyHostToDevice));
        printf("==== mask: %s\n", mm == 0 ? "synthetic mask16" : "all ones (pure copy)");
        timeit("copy u4 nt grid=8192", [&] { hipLaunchKernelGGL((k_copy<4, true, true>), dim3(8192), dim3(256), 0, stream, in, out, nvec); });
        RUN(256, 4, 16, true, true, 2048);
        RUN(512, 2, 4, true, true, 262144);
        RUN(1024, 2, 8, true, true, 131072);
        for (int tc : {1, 2, 3, 4, 6, 8, 16}) for (int bs : {192, 256, 384, 512, 768, 1024}) for (int lds : {0, 20480, 40960}) {
            const int nv = tc * (int)vpc; const int un = (nv + bs - 1) / bs;
            if (un > 4) continue;
            if (lds && !(tc == 4 || tc == 8)) continue;
            char nm[128]; snprintf(nm, 128, "rt tc=%d bs=%d un=%d lds=%d", tc, bs, un, lds);
            const unsigned g = (unsigned)((ncols + tc - 1) / tc);
            if (un == 1) timeit(nm, [&] { hipLaunchKernelGGL((k_carve_rt<1>), dim3(g), dim3(bs), lds, stream, in, out, mask, ncols, vpc, magic, tc); });
            if (un == 2) timeit(nm, [&] { hipLaunchKernelGGL((k_carve_rt<2>), dim3(g), dim3(bs), lds, stream, in, out, mask, ncols, vpc, magic, tc); });
            if (un == 3) timeit(nm, [&] { hipLaunchKernelGGL((k_carve_rt<3>), dim3(g), dim3(bs), lds, stream, in, out, mask, ncols, vpc, magic, tc); });
            if (un == 4) timeit(nm, [&] { hipLaunchKernelGGL((k_carve_rt<4>), dim3(g), dim3(bs), lds, stream, in, out, mask, ncols, vpc, magic, tc); });
        }
        RUN(512, 2, 4, true, true, 262144);
    }
    return 0;
}
