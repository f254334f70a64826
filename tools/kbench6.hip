// Timeline microbench of the 90-degree step on x-row streams (csrc/rotate_tiled.hip k_rot90wf) for grids with odd row lengths:
// where does a 50 us launch spend its time -- dispatch ramp, first load, per-segment iterations, tail?  A stand-alone restatement
// of the kernel's data movement (no masks, all cells valid) with s_memrealtime stamps per workgroup and variants of the load side.
// Development tool (tools/kbench6.bin, git-ignored):  hipcc -O3 --offload-arch=gfx950 tools/kbench6.hip -o tools/kbench6.bin
//   tools/kbench6.bin W H D [variant] [fill] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint32_t u32; typedef int64_t i64; typedef uint8_t u8; typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

struct magic { u32 m; int sa, sb; u32 d; };
static magic make_magic(u32 d) { int L = 0; while ((1ull << L) < d) ++L; magic g; g.m = (u32)(((1ull << 32) * ((1ull << L) - d)) / d + 1); g.sa = L < 1 ? L : 1; g.sb = L > 1 ? L - 1 : 0; g.d = d; return g; }
__device__ __forceinline__ u32 mdiv(u32 n, const magic g) { const u32 t = __umulhi(g.m, n); return (t + ((n - t) >> g.sa)) >> g.sb; }
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// VAR 0: unaligned 16-byte loads, one segment ahead.  1: two segments ahead.  2: aligned loads, neighbour's block by DPP-free shuffle,
// byte shift in registers.  3: no loads at all (stores only).  4: loads only (no stores).  5: variant 0 + the mask byte loads of the
// product kernel (one per source row and thread, one per output row and thread), kept raw.  6: the same bytes, one load per ROW by 2 waves, shared through LDS.
template <int VAR>
__global__ __launch_bounds__(1024) void k_wf(const u8* __restrict__ in, u8* __restrict__ out, int c0, i64 W, i64 H, i64 D, int TS, int nxt, int nch,
                                            magic mD, i64 nseg, int npc, u64* __restrict__ trace, const u8* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) u8 wtile[];
    __shared__ __attribute__((aligned(16))) u8 msh[2][512];
    const int tid = threadIdx.x;
    const i64 b = blockIdx.x;
    const i64 sgrp = b >> 3;
    const i64 xt = sgrp % nxt, sc = (sgrp / nxt) * 8 + (b & 7);
    if (sc >= nch) return;
    u64* tr = trace + b * 32;
    if (tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
    const int XW = 16 * npc;
    const i64 x0 = xt * XW;
    const i64 s_beg = sc * TS;
    const i64 s_end = s_beg + TS < nseg ? s_beg + TS : nseg;
    const i64 HD = H * D;
    const int cb = tid & 15;
    const i64 scol = x0 + 16 * cb;
    const int cmode = (VAR == 2 ? 16 * cb > XW : 16 * cb >= XW) ? 0 : (VAR == 2 || scol + 15 < D) ? 2 : (scol < D ? 1 : 0);
    const int zg = tid & 15, xg = tid >> 4;
    const int g = 15 - zg;
    const u32 rd_off = (u32)(16 * g * 256 + 16 * ((xg >> 2) ^ g) + 4 * (xg & 3));
    const bool xrow_ok = 4 * xg < XW;
    constexpr int DEPTH = VAR == 1 ? 2 : 1;
    u32x4 stg[DEPTH][4];
    u32 ms_raw = 0, md_raw = 0; u8 mrow = 0; u32 phs = 0;
    auto load_seg = [&](u32x4 (&st)[4], i64 s) {
        const bool live = s < s_end;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 4) + 64 * j;
            const u32 f = (u32)(256 * s) + (u32)(255 - lr);
            const u32 y = mdiv(f, mD), z = f - y * mD.d;
            const i64 n0 = (i64)c0 - (i64)z;
            st[j] = (u32x4)(0u);
            if (VAR == 3) continue;
            if (live && (i64)f < HD && cmode != 0 && n0 >= 0 && n0 < W) {
                const u8* sp = in + (n0 * H + (i64)y) * D + scol;
                if (VAR == 2) {
                    // aligned blocks: lane cb loads block cb of the run that starts at the row's address rounded down to 16 (one lane more
                    // than the row has blocks: npc <= 15), the neighbour's block comes by DPP (row_shl:1 within the 16 lanes of the row),
                    // the bytes are shifted into place in registers
                    const u8* base = in + (n0 * H + (i64)y) * D + x0;
                    const u32 ph = (u32)((uintptr_t)base & 15u);
                    const u8* ap = base - ph + 16 * cb;
                    st[j] = __builtin_nontemporal_load((const u32x4*)ap);       // raw: shifted where it is consumed
                    phs |= ph << (4 * j);
                } else if (cmode == 2 || sp + 16 <= in + W * HD) st[j] = __builtin_nontemporal_load((const u32x4_u*)sp);
                if (VAR == 5) ms_raw |= (u32)mask[n0 * H + (i64)y] << (8 * j);
            }
        }
        if (VAR == 5) {
            const u32 f = (u32)(256 * s) + (u32)(16 * zg);
            const u32 y = mdiv(f, mD);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const i64 x = x0 + 4 * xg + i; if (live && x < W && (i64)f < HD) md_raw |= (u32)mask[x * H + (i64)y] << (8 * i); }
        }
        if (VAR == 6 && tid < 512) {         // thread t < 256: source row t of the segment; 256 + t: output row t
            mrow = 0;
            if (tid < 256) {
                const u32 f = (u32)(256 * s) + (u32)(255 - tid);
                const u32 y = mdiv(f, mD), z = f - y * mD.d;
                const i64 n0 = (i64)c0 - (i64)z;
                if (live && (i64)f < HD && n0 >= 0 && n0 < W) mrow = mask[n0 * H + (i64)y];
            } else {
                const u32 f = (u32)(256 * s);
                const u32 y = mdiv(f, mD);
                const i64 x = x0 + tid - 256;
                if (live && x < W) mrow = mask[x * H + (i64)y];
            }
        }
    };
    load_seg(stg[0], s_beg);
    if (DEPTH == 2) load_seg(stg[1], s_beg + 1);
    int it = 0;
    for (i64 s = s_beg; s < s_end; ++s, ++it) {
        u32x4 (&cur)[4] = stg[DEPTH == 2 ? (it & 1) : 0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 4) + 64 * j;
            u32x4 val = cur[j];
            if (VAR == 2) {
                const u32x4 a = cur[j];
                u32x4 nb;
                nb.x = (u32)__builtin_amdgcn_update_dpp(0, (int)a.x, 0x101, 0xf, 0xf, true); nb.y = (u32)__builtin_amdgcn_update_dpp(0, (int)a.y, 0x101, 0xf, 0xf, true);
                nb.z = (u32)__builtin_amdgcn_update_dpp(0, (int)a.z, 0x101, 0xf, 0xf, true); nb.w = (u32)__builtin_amdgcn_update_dpp(0, (int)a.w, 0x101, 0xf, 0xf, true);
                const u32 w8[8] = {a.x, a.y, a.z, a.w, nb.x, nb.y, nb.z, nb.w};
                const u32 ph = (phs >> (4 * j)) & 15u, dq = ph >> 2, bs = ph & 3;
                u32 r5[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) r5[k] = dq == 0 ? w8[k] : dq == 1 ? w8[k + 1] : dq == 2 ? w8[k + 2] : w8[k + 3];
                val.x = __builtin_amdgcn_alignbyte(r5[1], r5[0], bs); val.y = __builtin_amdgcn_alignbyte(r5[2], r5[1], bs);
                val.z = __builtin_amdgcn_alignbyte(r5[3], r5[2], bs); val.w = __builtin_amdgcn_alignbyte(r5[4], r5[3], bs);
            }
            if (16 * cb < XW || VAR != 2) *(u32x4*)(wtile + lr * 256 + 16 * (cb ^ ((lr >> 4) & 15))) = (VAR == 5 && !((ms_raw >> (8 * j)) & 0xffu)) ? (u32x4)(0u) : val;
        }
        u32 keepm = 0xf;
        if (VAR == 2) phs = 0;
        if (VAR == 5) { keepm = 0; for (int i = 0; i < 4; ++i) keepm |= (u32)(((md_raw >> (8 * i)) & 0xffu) != 0) << i; ms_raw = 0; md_raw = 0; }
        if (VAR == 6 && tid < 512) msh[it & 1][tid] = mrow;
        __syncthreads();
        if (tid == 0 && it < 28) tr[1 + it] = __builtin_amdgcn_s_memrealtime();
        load_seg(cur, s + DEPTH);
        u32 d[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) d[rr] = *(const u32*)(wtile + rd_off + rr * 256);
        if (VAR == 6) {
            const u32x4 m16 = *(const u32x4*)(&msh[it & 1][16 * g]);
            const u32 mw[4] = {m16.x, m16.y, m16.z, m16.w};
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) if (!((mw[rr >> 2] >> (8 * (rr & 3))) & 0xffu)) d[rr] = 0;
        }
        __syncthreads();
        if (!xrow_ok || 256 * s + 16 * zg >= HD || VAR == 4) continue;
        u32 o[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32 A = d[15 - 4 * w], B = d[14 - 4 * w], Cc = d[13 - 4 * w], E = d[12 - 4 * w];
            const u32 t0 = perm(B, A, 0x05010400u), t1 = perm(B, A, 0x07030602u);
            const u32 u0 = perm(E, Cc, 0x05010400u), u1 = perm(E, Cc, 0x07030602u);
            o[0][w] = perm(u0, t0, 0x05040100u); o[1][w] = perm(u0, t0, 0x07060302u);
            o[2][w] = perm(u1, t1, 0x05040100u); o[3][w] = perm(u1, t1, 0x07060302u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            if (x >= W) continue;
            u32x4 r; r.x = o[i][0]; r.y = o[i][1]; r.z = o[i][2]; r.w = o[i][3];
            if (VAR == 5 && !((keepm >> i) & 1u)) r = (u32x4)(0u);
            if (VAR == 6) { if (!msh[it & 1][256 + 4 * xg + i]) r = (u32x4)(0u); }
            __builtin_nontemporal_store(r, (u32x4*)(out + x * HD + 256 * s + 16 * zg));
        }
    }
    if (tid == 0) { tr[30] = __builtin_amdgcn_s_memrealtime(); tr[31] = (u64)it; }
}

static int planes_per_chunk(i64 H, i64 tiles, int cus, int want_ty, int fill) {
    i64 m = (H + 8 * want_ty - 1) / (8 * want_ty);
    if (m < 1) m = 1;
    while (tiles * 8 * m < (i64)cus * fill && 8 * (m + 1) <= H) ++m;
    i64 ty = (H + 8 * m - 1) / (8 * m);
    return (int)(ty < 1 ? 1 : ty);
}

int main(int argc, char** argv) {
    const i64 W = argc > 1 ? atoll(argv[1]) : 355, H = argc > 2 ? atoll(argv[2]) : 512, D = argc > 3 ? atoll(argv[3]) : 355;
    const int var = argc > 4 ? atoi(argv[4]) : 0, fill = argc > 5 ? atoi(argv[5]) : 1, reps = argc > 6 ? atoi(argv[6]) : 5;
    const i64 n = W * H * D, HD = H * D;
    std::vector<u8> h(n);
    for (i64 i = 0; i < n; ++i) h[i] = (u8)((i * 2654435761u) >> 13);
    u8 *din, *dout, *dmask; u64* dtr;
    CK(hipMalloc(&dmask, W * H + 4096)); CK(hipMemset(dmask, 1, W * H + 4096));
    CK(hipMalloc(&din, n + 4096)); CK(hipMalloc(&dout, n + 4096));
    CK(hipMemcpy(din, h.data(), n, hipMemcpyHostToDevice));
    const i64 nx0 = (W + 255) / 256;
    const int npc = (int)((((W + nx0 - 1) / nx0) + 15) / 16);
    const i64 nseg = (HD + 255) / 256, nxt = (W + 16 * npc - 1) / (16 * npc);
    const int TS = planes_per_chunk(nseg, nxt, 256, 32, fill);
    const int nch = (int)((nseg + TS - 1) / TS);
    const unsigned blocks = (unsigned)(8ll * nxt * ((nch + 7) / 8));
    CK(hipMalloc(&dtr, (size_t)blocks * 32 * 8));
    printf("W %lld H %lld D %lld var %d: npc %d nxt %lld nseg %lld TS %d chunks %d blocks %u\n", (long long)W, (long long)H, (long long)D, var, npc, (long long)nxt, (long long)nseg, TS, nch, blocks);
    const int c0 = (int)W;
    auto launch = [&]() {
        const magic mD = make_magic((u32)D);
#define L(V) hipLaunchKernelGGL(k_wf<V>, dim3(blocks), dim3(1024), 65536, 0, din, dout, c0, W, H, D, TS, (int)nxt, nch, mD, nseg, npc, dtr, dmask)
        switch (var) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; default: L(6); }
    };
    CK(hipFuncSetAttribute((const void*)k_wf<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_wf<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %.1f us per launch, %.2f Mvoxel/us\n", 1e3 * ms / reps, n / (1e3 * ms / reps) * 1e-6);
    CK(hipMemset(dtr, 0, (size_t)blocks * 32 * 8));
    launch(); CK(hipDeviceSynchronize());
    std::vector<u64> tr((size_t)blocks * 32);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    u64 t0 = ~0ull, t1 = 0; int live = 0;
    for (unsigned b = 0; b < blocks; ++b) if (tr[b * 32]) { t0 = std::min(t0, tr[b * 32]); t1 = std::max(t1, tr[b * 32 + 30]); ++live; }
    std::vector<double> st, first, iter, dur;
    for (unsigned b = 0; b < blocks; ++b) if (tr[b * 32]) {
        const u64* t = &tr[b * 32]; const int it = (int)t[31];
        st.push_back((t[0] - t0) * 0.01); dur.push_back((t[30] - t[0]) * 0.01); first.push_back((t[1] - t[0]) * 0.01);
        for (int k = 1; k < it && k < 28; ++k) iter.push_back((t[1 + k] - t[k]) * 0.01);
    }
    auto pct = [](std::vector<double>& v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("  live workgroups %d, span %.1f us; start p50 %.1f p90 %.1f max %.1f; first barrier after p50 %.1f p90 %.1f; iteration p10 %.2f p50 %.2f p90 %.2f; wg duration p50 %.1f p90 %.1f max %.1f\n",
           live, (t1 - t0) * 0.01, pct(st, .5), pct(st, .9), pct(st, 1.0), pct(first, .5), pct(first, .9), pct(iter, .1), pct(iter, .5), pct(iter, .9), pct(dur, .5), pct(dur, .9), pct(dur, 1.0));
    if ((var <= 2 || var >= 5) && W == D) {      // check: out[x, y, z] = in[W - z, y, x] (0 where W - z is outside)
        std::vector<u8> o(n);
        CK(hipMemcpy(o.data(), dout, n, hipMemcpyDeviceToHost));
        i64 bad = 0;
        for (i64 x = 0; x < W; x += 7) for (i64 y = 0; y < H; y += 5) for (i64 z = 0; z < D; ++z) {
            const i64 n0 = W - z; const u8 want = (n0 >= 0 && n0 < W) ? h[(n0 * H + y) * D + x] : 0;
            if (o[(x * H + y) * D + z] != want) ++bad;
        }
        printf("  check: %lld mismatches (sampled)\n", (long long)bad);
    }
    return 0;
}
