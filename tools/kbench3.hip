// Memory-system microbench for the 90-degree tile kernels: what does HBM deliver when a workgroup moves 16 KiB as
// R rows x TWB bytes with the rows 1 MiB apart (x- or n0-rows of a 1024^3 byte volume), on the read side, the write
// side or both, compared with a linear copy?  Development tool (tools/kbench3.bin, git-ignored).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32; typedef int64_t i64; typedef uint8_t u8;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

// MODE bit 0: strided read, bit 1: strided write (otherwise linear 16 KiB chunks)
template <int TWB, int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_tilecopy(const u8* __restrict__ in, u8* __restrict__ out, i64 S, int TY) {
    constexpr int LPR = TWB / 16;          // lanes per row
    constexpr int RPI = 256 / LPR;         // rows per instruction
    constexpr int R = 16384 / TWB;         // rows per tile
    const int ncb = (int)(S / TWB), nrb = (int)(S / R);
    i64 b = blockIdx.x;
    const i64 cbk = b % ncb; b /= ncb;
    const i64 rbk = b % nrb; b /= nrb;
    const i64 y0 = b * TY;
    const int tid = threadIdx.x;
    const i64 row = rbk * R + tid / LPR, col = cbk * TWB + 16 * (tid % LPR);
    const i64 lin0 = ((rbk * ncb + cbk) * S) * 16384;       // linear chunk base for plane 0 of this tile column
    u32x4 v[DEPTH][4];
    auto load = [&](u32x4 (&r)[4], i64 y) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u8* p = (MODE & 1) ? in + ((row + (i64)RPI * j) * S + y) * S + col : in + lin0 + y * 16384 + 4096 * j + 16 * tid;
            r[j] = __builtin_nontemporal_load((const u32x4*)p);
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load(v[s], y0 + s);
    for (i64 yy = y0; yy < y0 + TY; yy += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const i64 y = yy + s;
            u32x4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = v[s][j];
            if (y + DEPTH < y0 + TY) load(v[s], y + DEPTH);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u8* p = (MODE & 2) ? out + ((row + (i64)RPI * j) * S + y) * S + col : out + lin0 + y * 16384 + 4096 * j + 16 * tid;
                __builtin_nontemporal_store(w[j], (u32x4*)p);
            }
        }
    }
}

template <int TWB, int MODE, int DEPTH>
static void run(const u8* in, u8* out, i64 S, const char* name) {
    const int TY = 32;
    const i64 nblk = (S * S * S / 16384) / TY;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_tilecopy<TWB, MODE, DEPTH>), dim3((unsigned)nblk), dim3(256), 0, 0, in, out, S, TY);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_tilecopy<TWB, MODE, DEPTH>), dim3((unsigned)nblk), dim3(256), 0, 0, in, out, S, TY);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-28s TWB=%4d depth=%d  %.4f ms  %.0f GB/s\n", name, TWB, DEPTH, ms, 2.0 * S * S * S / ms / 1e6);
}

int main() {
    const i64 S = 1024;
    u8 *in, *out;
    CK(hipMalloc(&in, S * S * S)); CK(hipMalloc(&out, S * S * S));
    CK(hipMemset(in, 1, S * S * S)); CK(hipMemset(out, 0, S * S * S));
    run<128, 0, 1>(in, out, S, "linear -> linear");
    run<128, 0, 2>(in, out, S, "linear -> linear");
    run<64, 1, 1>(in, out, S, "strided read -> linear");
    run<128, 1, 1>(in, out, S, "strided read -> linear");
    run<256, 1, 1>(in, out, S, "strided read -> linear");
    run<512, 1, 1>(in, out, S, "strided read -> linear");
    run<1024, 1, 1>(in, out, S, "strided read -> linear");
    run<64, 2, 1>(in, out, S, "linear -> strided write");
    run<128, 2, 1>(in, out, S, "linear -> strided write");
    run<256, 2, 1>(in, out, S, "linear -> strided write");
    run<512, 2, 1>(in, out, S, "linear -> strided write");
    run<1024, 2, 1>(in, out, S, "linear -> strided write");
    run<128, 3, 1>(in, out, S, "strided -> strided");
    run<128, 3, 2>(in, out, S, "strided -> strided");
    run<256, 3, 1>(in, out, S, "strided -> strided");
    run<256, 3, 2>(in, out, S, "strided -> strided");
    run<512, 3, 1>(in, out, S, "strided -> strided");
    run<1024, 3, 1>(in, out, S, "strided -> strided");
    return 0;
}
