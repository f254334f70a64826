// Kernel-variant microbenchmark for the K1 mask-carve sweep (development tool, not shipped).
// hipcc -O3 --offload-arch=gfx950 tools/kbench.hip -o gpurun_out/kbench && ./gpurun_out/kbench
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

// plain copy, grid-stride, UNROLL vectors in flight per lane
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

// wave-tile carve (the shipped v0 structure): one wave per 64-column tile, vpc % 64 == 0
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_carve_wave(const u32x4* __restrict__ in, u32x4* __restrict__ out, const u8* __restrict__ mask,
                                                    i64 ncols, u32 vpc, u32 magic, i64 ntiles) {
    const u32 lane = threadIdx.x & 63;
    const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (blockDim.x >> 6);
    for (i64 t = wave; t < ntiles; t += nwaves) {
        const i64 col0 = t * 64;
        const i64 c_here = ncols - col0 < 64 ? ncols - col0 : 64;
        const u8 m = (i64)lane < c_here ? mask[col0 + lane] : (u8)0;
        const u64 kbits = __ballot(m != 0);
        const u32 nvec = (u32)c_here * vpc;
        const u32x4* src = in + col0 * vpc; u32x4* dst = out + col0 * vpc;
        for (u32 v0 = 0; v0 < nvec; v0 += 64 * UNROLL) {
            u32x4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const u32 vs = v0 + 64 * u;
                x[u] = (u32x4)(0u);
                if (vs < nvec) { const u32 c = __umulhi(vs, magic); if ((kbits >> c) & 1) x[u] = ld<NTL>(src + vs + lane); }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { const u32 vs = v0 + 64 * u; if (vs < nvec) st<NTS>(dst + vs + lane, x[u]); }
        }
    }
}

// block-tile carve: the 4 waves of a block sweep one tile of TC columns together (contiguous 4 KiB per step);
// mask bits for the tile staged in LDS as ballots
template <int UNROLL, bool NTL, bool NTS, int TC>
__global__ __launch_bounds__(256) void k_carve_block(const u32x4* __restrict__ in, u32x4* __restrict__ out, const u8* __restrict__ mask,
                                                     i64 ncols, u32 vpc, i64 ntiles) {
    __shared__ u8 smask[TC];
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 col0 = t * TC;
        const int c_here = (int)(ncols - col0 < TC ? ncols - col0 : TC);
        __syncthreads();
        for (int c = threadIdx.x; c < TC; c += 256) smask[c] = c < c_here ? mask[col0 + c] : 0;
        __syncthreads();
        const u32 nvec = (u32)c_here * vpc;
        const u32x4* src = in + col0 * vpc; u32x4* dst = out + col0 * vpc;
        for (u32 v0 = 0; v0 < nvec; v0 += 256 * UNROLL) {
            u32x4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const u32 vv = v0 + 256 * u + threadIdx.x;
                x[u] = (u32x4)(0u);
                if (vv < nvec) { const u32 c = vv / vpc; if (smask[c]) x[u] = ld<NTL>(src + vv); }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { const u32 vv = v0 + 256 * u + threadIdx.x; if (vv < nvec) st<NTS>(dst + vv, x[u]); }
        }
    }
}

// column-per-wave-step carve: flat grid-stride over (column, 1KiB chunk) pairs; mask byte read per step (scalar)
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_carve_flat(const u32x4* __restrict__ in, u32x4* __restrict__ out, const u8* __restrict__ mask,
                                                    i64 nchunks /* = ncols*vpc/64 */, u32 cpc /* chunks per column */, u32 magic) {
    const u32 lane = threadIdx.x & 63;
    const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * (blockDim.x >> 6);
    for (i64 ch = wave * UNROLL; ch < nchunks; ch += nwaves * UNROLL) {
        u32x4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const i64 c = ch + u;
            x[u] = (u32x4)(0u);
            if (c < nchunks) { const i64 col = c / cpc; if (mask[col]) x[u] = ld<NTL>(in + c * 64 + lane); }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const i64 c = ch + u; if (c < nchunks) st<NTS>(out + c * 64 + lane, x[u]); }
    }
}

static u64 splitmix64(u64 z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__global__ void k_fill(u32* p, i64 n) { for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) p[i] = (u32)(i * 2654435761u) ^ (u32)(i >> 7); }

int main(int argc, char** argv) {
    const i64 S = argc > 1 ? atoll(argv[1]) : 1024;
    const i64 ncols = S * S, col = S * 3, nbytes = ncols * col, nvec = nbytes / 16;
    const u32 vpc = (u32)(col / 16);
    const u32 magic = (u32)(((1ull << 32) + vpc - 1) / vpc);
    u32x4 *in, *out; u8* mask;
    CK(hipMalloc(&in, nbytes)); CK(hipMalloc(&out, nbytes)); CK(hipMalloc(&mask, ncols));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u32*)in, nbytes / 4);
    // synthetic mask16 foreground (same formula as csrc/synth.hip), (W,H) orientation
    std::vector<u8> hm(ncols), ones(ncols, 1);
    i64 fg = 0;
    for (i64 x = 0; x < S; ++x) for (i64 y = 0; y < S; ++y) {
        i64 xn = x * 1024 / S, yn = y * 1024 / S, dx2 = 2 * xn - 1023, adx2 = dx2 < 0 ? -dx2 : dx2, dy2 = 2 * (yn - 256);
        bool body = adx2 < 840 && yn >= 256, dome = dx2 * dx2 * 40000 + dy2 * dy2 * 90000 < 4ll * 90000 * 40000, tow = adx2 > 880 && adx2 < 960 && yn >= 96;
        hm[x * S + y] = body || dome || tow; fg += hm[x * S + y];
    }
    printf("S=%lld bytes=%.2f GB fg=%.4f\n", (long long)S, nbytes / 1e9, (double)fg / ncols);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    auto timeit = [&](const char* name, auto launch, double alg_bytes) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps; tot += ms; if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-44s best %.4f ms  avg %.4f ms  alg %.0f GB/s\n", name, best, tot / 3, alg_bytes / (best * 1e-3) / 1e9);
    };
    const double full = 2.0 * nbytes;
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipMemcpy(mask, pass == 0 ? ones.data() : hm.data(), ncols, hipMemcpyHostToDevice));
        printf("---- mask: %s\n", pass == 0 ? "all ones (pure copy)" : "synthetic mask16");
        if (pass == 0) {
            for (int b : {1024, 2048, 4096, 8192}) {
                char nm[96];
                snprintf(nm, 96, "copy u4 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_copy<4, true, true>), dim3(b), dim3(256), 0, 0, in, out, nvec); }, full);
                snprintf(nm, 96, "copy u8 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_copy<8, true, true>), dim3(b), dim3(256), 0, 0, in, out, nvec); }, full);
                snprintf(nm, 96, "copy u4 plain grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_copy<4, false, false>), dim3(b), dim3(256), 0, 0, in, out, nvec); }, full);
                snprintf(nm, 96, "copy u4 ld-plain st-nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_copy<4, false, true>), dim3(b), dim3(256), 0, 0, in, out, nvec); }, full);
            }
            hipMemcpyAsync(out, in, nbytes, hipMemcpyDeviceToDevice, 0);
            timeit("hipMemcpyAsync d2d", [&] { hipMemcpyAsync(out, in, nbytes, hipMemcpyDeviceToDevice, 0); }, full);
        }
        const i64 ntiles64 = (ncols + 63) / 64;
        for (int b : {1024, 2048, 4096, 8192}) {
            char nm[96];
            snprintf(nm, 96, "wave-tile u4 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_wave<4, true, true>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, magic, ntiles64); }, full);
            snprintf(nm, 96, "wave-tile u8 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_wave<8, true, true>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, magic, ntiles64); }, full);
            snprintf(nm, 96, "wave-tile u4 plain grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_wave<4, false, false>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, magic, ntiles64); }, full);
            snprintf(nm, 96, "wave-tile u4 ld-plain st-nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_wave<4, false, true>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, magic, ntiles64); }, full);
            snprintf(nm, 96, "block-tile16 u4 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_block<4, true, true, 16>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, (ncols + 15) / 16); }, full);
            snprintf(nm, 96, "block-tile64 u4 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_block<4, true, true, 64>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, (ncols + 63) / 64); }, full);
            snprintf(nm, 96, "block-tile64 u8 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_block<8, true, true, 64>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols, vpc, (ncols + 63) / 64); }, full);
            snprintf(nm, 96, "flat u4 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_flat<4, true, true>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols * (vpc / 64), vpc / 64, 0); }, full);
            snprintf(nm, 96, "flat u8 nt/nt grid=%d", b); timeit(nm, [&] { hipLaunchKernelGGL((k_carve_flat<8, true, true>), dim3(b), dim3(256), 0, 0, in, out, mask, ncols * (vpc / 64), vpc / 64, 0); }, full);
        }
    }
    return 0;
}
