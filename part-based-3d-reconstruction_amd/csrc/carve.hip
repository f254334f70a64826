// K1 mask-carve sweep, occupancy, colour apply, part overlay -- HBM-bound byte kernels (gfx950).
//
// Data layout (the reference's own): grid (W,H,D[,3]) uint8 in C order, so the D*C bytes of
// one (x,y) column are contiguous and column index xy = x*H + y also indexes the (W,H) mask.
// Every kernel here is a coalesced 16-byte-per-lane sweep; none has a contraction (no MFMA).
#include "pb3d_internal.h"
#include <vector>

namespace {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));     // 16-byte access at any byte alignment (odd-sized grids, buffer views)

__device__ __forceinline__ u32x4 ld_nt(const u32x4_u* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(u32x4_u* p, u32x4 v) { __builtin_nontemporal_store(v, p); }
// RGB side of the 16-voxels-per-lane kernels: a lane owns 48 contiguous bytes, so every 128-byte line is touched by three
// instructions of the wave.  Nontemporal hints make the line leave the L1 between them (measured 1.4x slower on the
// point sweep of project.hip), so these accesses stay plain.
#ifndef PB3D_STRIDED_NT
#define PB3D_STRIDED_NT 0
#endif
__device__ __forceinline__ u32x4 ld_s(const u32x4* p) { return PB3D_STRIDED_NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_s(u32x4* p, u32x4 v) { if (PB3D_STRIDED_NT) __builtin_nontemporal_store(v, p); else *p = v; }

// RGB stores of those kernels: the wave's 64 groups form 3 KiB contiguous in the output (vector index 3 * gw0 ...), but a lane
// holds vectors 3*lane .. 3*lane+2 of it.  Route them through a wave-private LDS window (192 vectors) so that every store
// instruction writes 1 KiB contiguous (whole 128-byte lines) instead of 16 bytes every 48.  Called by all 64 lanes.
__device__ __forceinline__ void store48_wave(u32x4_u* __restrict__ out, i64 gw0, i64 ngroups, const u32x4 r[3], u32x4* lds_w) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) lds_w[3 * lane + k] = r[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const i64 left = ngroups - gw0;
    const int nvec = 3 * (int)(left < 64 ? left : 64);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) __builtin_nontemporal_store(lds_w[idx], out + 3 * gw0 + idx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// K1 fast path: column size col = D*C is a multiple of 16 bytes, buffers 16-byte aligned.
// One workgroup (512 threads) per voxel tile of kTileVec = 768 consecutive 16-byte vectors (12 KiB:
// 4 columns of a 1024^3 RGB grid).  The slice of the 2-D mask the tile touches is staged in LDS;
// each lane looks up its column's keep byte and a wave64 ballot turns the 64 decisions into one
// scalar word: all-drop wavefronts issue no load at all, all-keep wavefronts load unpredicated.
// Dropped columns are never read -- only zeros are stored.  Many small tiles and at most two loads
// per lane measured fastest on MI355X (profiles/r01_k1_variants_run*.txt): the hardware dispatcher
// balances the stream better than a persistent grid-stride loop.
// ------------------------------------------------------------------------------------------------
constexpr int kTileVec = 768;
constexpr int kTileThreads = 512;
constexpr int kTileMaskMax = kTileVec + 2;  // vpc == 1 -> one column per vector (+ straddle)

__global__ __launch_bounds__(kTileThreads) void k_carve_tiles(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                              const u8* __restrict__ mask, i64 nvec, i64 ncols, u32 vpc,
                                                              u32 magic) {
    __shared__ u8 smask[kTileMaskMax];
    const i64 v_begin = (i64)blockIdx.x * kTileVec;
    const i64 first_col = v_begin / vpc;                       // block-uniform (scalar unit)
    const u32 rem0 = (u32)(v_begin - first_col * vpc);
    const i64 left = nvec - v_begin;
    const u32 n_here = left < kTileVec ? (u32)left : (u32)kTileVec;
    const u32 ncol_here = (rem0 + n_here + vpc - 1) / vpc;
    for (u32 c = threadIdx.x; c < ncol_here; c += kTileThreads) smask[c] = first_col + c < ncols ? mask[first_col + c] : (u8)0;
    __syncthreads();
    const u32x4* src = in + v_begin;
    u32x4* dst = out + v_begin;
    u32x4 x[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const u32 lv = threadIdx.x + kTileThreads * u;
        live[u] = lv < n_here;
        x[u] = (u32x4)(0u);
        bool keep = false;
        if (live[u]) {
            const u32 q = rem0 + lv;
            keep = smask[magic ? __umulhi(q, magic) : q / vpc] != 0;
        }
        const u64 kb = __ballot(keep);
        if (kb == ~0ull) x[u] = ld_nt(src + lv);               // whole wavefront keeps: unpredicated 1 KiB load
        else if (kb != 0 && keep) x[u] = ld_nt(src + lv);      // mixed wavefront (column boundary inside it)
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (live[u]) st_nt(dst + threadIdx.x + kTileThreads * u, x[u]);
}

// K1 generic path: any column size / alignment; one thread per byte run of 16 (slow, rarely used:
// only when D*C is not a multiple of 16 or a slab pointer is unaligned).
__global__ __launch_bounds__(256) void k_carve_bytes(const u8* __restrict__ in, u8* __restrict__ out,
                                                     const u8* __restrict__ mask, i64 nbytes, i64 col) {
    const i64 nchunks = (nbytes + 15) / 16;
    for (i64 ch = (i64)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += (i64)gridDim.x * blockDim.x) {
        i64 b = ch * 16;
        const i64 e = b + 16 < nbytes ? b + 16 : nbytes;
        i64 c = b / col;
        i64 rem = b - c * col;
        u8 keep = mask[c];
        for (; b < e; ++b) {
            if (rem == col) {
                rem = 0;
                ++c;
                keep = mask[c];
            }
            out[b] = keep ? in[b] : (u8)0;
            ++rem;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1 for any column size col = D*C >= 16 bytes and any buffer alignment (odd-sized grids: 355, 437, 123 ... voxels).  The
// volume is swept as the same flat stream of 16-byte pieces in 12 KiB tiles; a piece lies in one column or straddles
// two, in which case its bytes are masked on both sides of the boundary.  Column of a piece: one exact multiply-high
// division of its tile-relative byte offset.  Dropped pieces are never read.
// ------------------------------------------------------------------------------------------------
struct Magic32 { u32 m; int sa, sb; };
inline Magic32 make_magic32(u32 d) {     // exact u32 division, 1 <= d < 2^31 (Granlund-Montgomery round-up form)
    int L = 0;
    while ((1ull << L) < d) ++L;
    Magic32 g;
    g.m = (u32)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    g.sa = L < 1 ? L : 1;
    g.sb = L > 1 ? L - 1 : 0;
    return g;
}
__device__ __forceinline__ u32 magic_div32(u32 n, Magic32 g) {
    const u32 t = __umulhi(g.m, n);
    return (t + ((n - t) >> g.sa)) >> g.sb;
}
// dword j of a 16-byte mask whose bytes below position b (0..16) are 0xff
__device__ __forceinline__ u32 below_mask(int b, int j) {
    const int n = b - 4 * j;
    return n >= 4 ? 0xffffffffu : (n <= 0 ? 0u : ((1u << (8 * n)) - 1u));
}

__global__ __launch_bounds__(kTileThreads) void k_carve_flat(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask,
                                                             i64 nbytes, i64 ncols, u32 col, Magic32 mg) {
    __shared__ u8 smask[kTileVec + 4];
    const i64 b_begin = (i64)blockIdx.x * (kTileVec * 16);
    const i64 first_col = b_begin / col;                        // block-uniform
    const u32 rem0 = (u32)(b_begin - first_col * col);
    const i64 left = nbytes - b_begin;
    const u32 n_here = left < kTileVec * 16 ? (u32)left : (u32)(kTileVec * 16);       // bytes of this tile
    const u32 ncol_here = magic_div32(rem0 + n_here - 1, mg) + 1;
    for (u32 c = threadIdx.x; c < ncol_here; c += kTileThreads) smask[c] = first_col + c < ncols ? mask[first_col + c] : (u8)0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const u32 bo = 16u * (threadIdx.x + kTileThreads * u);   // byte offset of this lane's piece in the tile
        const bool live = bo < n_here;
        bool k0 = false, k1 = false, keep = false;
        int bnd = 16;                                            // bytes of the piece that belong to its first column
        if (live) {
            const u32 q = rem0 + bo, c0 = magic_div32(q, mg);
            const u32 to_end = col - (q - c0 * col);             // >= 1
            k0 = smask[c0] != 0;
            if (to_end < 16 && bo + to_end < n_here) { bnd = (int)to_end; k1 = smask[c0 + 1] != 0; }
            keep = k0 || k1;
        }
        if (!live) continue;
        const u8* sp = in + b_begin + bo;
        u8* dp = out + b_begin + bo;
        const bool whole = bo + 16 <= n_here;                    // else: the last piece of the volume, byte-wise
        u32x4 x = (u32x4)(0u);
        if (keep) {                                              // dropped pieces are never read
            if (whole) x = __builtin_nontemporal_load((const u32x4_u*)sp);
            else {
                u32 t4[4] = {0, 0, 0, 0};
                for (u32 b = 0; bo + b < n_here; ++b) t4[b >> 2] |= (u32)sp[b] << (8 * (b & 3));
                x.x = t4[0]; x.y = t4[1]; x.z = t4[2]; x.w = t4[3];
            }
            if (bnd < 16 || !k0) {                               // mixed piece: bytes below bnd follow k0, the rest k1
                const u32 a = k0 ? 0xffffffffu : 0u, b2 = k1 ? 0xffffffffu : 0u;
                const u32 m0 = below_mask(bnd, 0), m1 = below_mask(bnd, 1), m2 = below_mask(bnd, 2), m3 = below_mask(bnd, 3);
                x.x &= (m0 & a) | (~m0 & b2); x.y &= (m1 & a) | (~m1 & b2); x.z &= (m2 & a) | (~m2 & b2); x.w &= (m3 & a) | (~m3 & b2);
            }
        }
        if (whole) __builtin_nontemporal_store(x, (u32x4_u*)dp);
        else {
            const u32 t4[4] = {x.x, x.y, x.z, x.w};
            for (u32 b = 0; bo + b < n_here; ++b) dp[b] = (u8)(t4[b >> 2] >> (8 * (b & 3)));
        }
    }
}

// The mirror image for loads: the wave reads its 3 KiB with three contiguous 1 KiB instructions into the LDS window and
// every lane then picks up its own 48 bytes.  Called by all 64 lanes; lanes past the end get zeros.
__device__ __forceinline__ void load48_wave(const u32x4* __restrict__ in, i64 gw0, i64 ngroups, u32x4 r[3], u32x4* lds_w) {
    const int lane = threadIdx.x & 63;
    const i64 left = ngroups - gw0;
    const int nvec = 3 * (int)(left < 64 ? left : 64);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;
        lds_w[idx] = idx < nvec ? ld_nt(in + 3 * gw0 + idx) : (u32x4)(0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = lds_w[3 * lane + k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// _occupancy: 16 voxels (48 bytes in, 16 bytes out) per thread.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 byte_of(const u32* w, int idx) { return (w[idx >> 2] >> ((idx & 3) * 8)) & 0xffu; }

__global__ __launch_bounds__(256) void k_occupancy16(const u32x4* __restrict__ rgb, u32x4* __restrict__ occ, i64 ngroups) {
    __shared__ u32x4 stage[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (i64 gw0 = (i64)blockIdx.x * blockDim.x + 64 * wv; gw0 < ngroups; gw0 += (i64)gridDim.x * blockDim.x) {   // wave-uniform
        u32x4 v[3];
        load48_wave(rgb, gw0, ngroups, v, stage[wv]);
        const u32 w[12] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w};
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32 any = byte_of(w, 3 * i) | byte_of(w, 3 * i + 1) | byte_of(w, 3 * i + 2);
            o[i >> 2] |= (any ? 1u : 0u) << ((i & 3) * 8);
        }
        u32x4 r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
        if (gw0 + lane < ngroups) st_nt(occ + gw0 + lane, r);
    }
}

__global__ __launch_bounds__(256) void k_occupancy_tail(const u8* __restrict__ rgb, u8* __restrict__ occ, i64 v0, i64 nvox) {
    for (i64 v = v0 + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x)
        occ[v] = (rgb[3 * v] | rgb[3 * v + 1] | rgb[3 * v + 2]) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// apply_colored_mask_to_voxel_grid: out[x,y,z,:] = rgb[y,x,:] if carved[x,y,z] == 1 else 0.
// Fast path (D % 16 == 0): 16 voxels of one column per thread -> 16 bytes in, 48 bytes out.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void expand16(const u32 keep16, const u32 r, const u32 g, const u32 b, u32 w[12]) {
    // byte j of the 48 output bytes belongs to voxel j/3 and channel j%3.  The RGB stream repeats every 3 dwords (RGBR GBRG
    // BRGB); a dword covers parts of two voxels, whose keep masks are joined with a constant byte mask per dword phase.
    const u32 S[3] = {r | (g << 8) | (b << 16) | (r << 24), g | (b << 8) | (r << 16) | (g << 24), b | (r << 8) | (g << 16) | (b << 24)};
    u32 m[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) m[v] = (u32)__builtin_amdgcn_sbfe((int)keep16, v, 1);      // 0 or ~0
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int v0 = (4 * k) / 3, ph = (4 * k) % 3;          // first voxel of the dword and its channel phase
        // ph 0: bytes v0,v0,v0,v0+1 ; ph 1: v0,v0,v0+1,v0+1 ; ph 2: v0,v0+1,v0+1,v0+1
        const u32 lo = ph == 0 ? 0x00ffffffu : (ph == 1 ? 0x0000ffffu : 0x000000ffu);
        w[k] = S[k % 3] & ((m[v0] & lo) | (m[v0 + 1 < 16 ? v0 + 1 : 15] & ~lo));
    }
}

// bit i = (byte i of the 16 bytes == 1)
__device__ __forceinline__ u32 ones16(const u32 cw[4]) {
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 t = cw[j] ^ 0x01010101u;                                     // a byte equal to 1 becomes 0
        const u32 z = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu | t) & 0x80808080u;       // bit 7 of a byte set <=> the byte is non-zero
        const u32 e = (~z & 0x80808080u) >> 7;                                  // bit 0 of a byte set <=> the byte was 1
        bits |= ((e * 0x01020408u) >> 24) << (4 * j);                            // gather bits 0, 8, 16, 24 into 4 bits
    }
    return bits;
}

// FLAT = false: D % 16 == 0, a group of 16 voxels lies in one column.  FLAT = true: any D >= 16 -- groups are cut from the
// flat voxel stream and may straddle two columns (two pixel colours, split at voxel `bnd`).
template <bool FLAT>
__global__ __launch_bounds__(256) void k_color_apply16(const u8* __restrict__ carved, const u8* __restrict__ rgb_hw3,
                                                       u8* __restrict__ out, i64 W, i64 H, i64 D, i64 ngroups, pb3d_magic mD, pb3d_magic mH,
                                                       int small) {
    __shared__ u32x4 stage[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (i64 gw0 = (i64)blockIdx.x * blockDim.x + 64 * wv; gw0 < ngroups; gw0 += (i64)gridDim.x * blockDim.x) {   // wave-uniform
        const i64 g = gw0 + lane;
        u32x4 r[3] = {(u32x4)(0u), (u32x4)(0u), (u32x4)(0u)};
        if (g < ngroups) {
            // column of the group: exact u32 divisions when the grid has fewer than 2^32 voxels (two 64-bit divisions were ~200 of this
            // kernel's ~260 vector instructions per group, and the pixel load waits for them)
            i64 xy, x, y;
            if (small) { const u32 q = pb3d_div((u32)(16 * g), mD), qx = pb3d_div(q, mH); xy = q; x = qx; y = q - qx * mH.d; }
            else { xy = (16 * g) / D; x = xy / H; y = xy - x * H; }
            const u8* px = rgb_hw3 + (y * W + x) * 3;
            const u32x4 cv = ld_nt((const u32x4_u*)(carved + 16 * g));
            const u32 cw[4] = {cv.x, cv.y, cv.z, cv.w};
            const u32 keep16 = ones16(cw);
            u32 w[12];
            const i64 bnd = FLAT ? (xy + 1) * D - 16 * g : 16;     // voxels of the group that belong to column xy
            if (!FLAT || bnd >= 16) expand16(keep16, px[0], px[1], px[2], w);
            else {
                const i64 x1 = y + 1 < H ? x : x + 1, y1 = y + 1 < H ? y + 1 : 0;        // the next column in (x, y) order
                const u8* qx = rgb_hw3 + (y1 * W + x1) * 3;
                const u32 lo = (1u << bnd) - 1u;
                u32 w2[12];
                expand16(keep16 & lo, px[0], px[1], px[2], w);
                expand16(keep16 & ~lo, qx[0], qx[1], qx[2], w2);
#pragma unroll
                for (int k = 0; k < 12; ++k) w[k] |= w2[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { r[k].x = w[4 * k]; r[k].y = w[4 * k + 1]; r[k].z = w[4 * k + 2]; r[k].w = w[4 * k + 3]; }
        }
        store48_wave((u32x4_u*)out, gw0, ngroups, r, stage[wv]);
    }
}

// ------------------------------------------------------------------------------------------------
// global_carve(binary, rgb, 90) (reference utils/voxel_carving_utils.py:269-298) as a STREAM (round 4):
//   out[x,y,z,:] = bm[x,y] && valid(x,z) && bm[c0 - z, y] ? rgb[y,x,:] : 0
// is k_color_apply16 with the 16 keep bits of a group COMPUTED instead of loaded: the validity bits of row x (k_rot_valid's table;
// bits past D are zero, so a group that runs over a column end masks itself) AND the mask bytes bm[c0 - z0 - 15 .. c0 - z0] of image
// row y -- sixteen neighbouring bytes, one load at whatever alignment, non-zero bytes gathered into bits and reversed.  Write-only,
// any D >= 16 and any alignment of the rows (the volume is a flat stream of 16-voxel groups), every wave store 1 KB contiguous through
// the wave-private window.  Replaces the per-row piece kernels k_global_carve90v / 90f of rounds 1-3, which formed the keep bits of
// every 16-BYTE piece from six LDS byte reads (355 x 512 x 355: 65 -> 40 us; 1024^3 the same 0.5 ms, it is the 3.2 GB written).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 spread4(u32 b4) { return ((b4 * 0x00204081u) & 0x01010101u) * 0xffu; }      // bit j -> byte j = 0xff
__device__ __forceinline__ u32 nonzero16(const u32 cw[4]) {      // bit i = (byte i of the 16 bytes != 0)
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 t = cw[j];
        const u32 z = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
        bits |= (((z >> 7) * 0x01020408u) >> 24) << (4 * j);
    }
    return bits;
}
// keep bits of voxels (x, y, z0 .. z0 + 15): brow = the image row y, vrow = the validity row x.  Every load is issued at once and
// depends on nothing but the group's coordinates (written as "mask byte, then -- if set -- validity, then -- if any -- the sixteen
// bytes" the chain of dependent round trips made the kernel 1.6x slower than the piece kernels it replaces).  The sixteen bytes are the
// image columns nlo .. nlo + 15, nlo = c0 - z0 - 15; where that range leaves [0, W) (the last group of a column) the load is moved
// inside and the bits are shifted back -- no branch, no byte loop.
__device__ __forceinline__ u32 gc90_bits(const u8* __restrict__ brow, const u32* __restrict__ vrow, i64 x, int z0, int c0, int W) {
    const u32* vr = vrow + (z0 >> 5);
    const int nlo = c0 - z0 - 15;
    const int L = nlo < 0 ? 0 : (nlo > W - 16 ? W - 16 : nlo);          // (W >= 16: the launcher's condition)
    const u32 bx = brow[x];
    const u32 v0 = vr[0], v1 = vr[1];
    const u32x4 t = *(const u32x4_u*)(brow + L);
    const u32 vb = (u32)((((u64)v1 << 32) | (u64)v0) >> (z0 & 31)) & 0xffffu;
    const u32 cw[4] = {t.x, t.y, t.z, t.w};
    const u32 nz = nonzero16(cw);                                       // bit m: image column L + m
    const int sh = L - nlo;                                             // bit j of the wanted range is column nlo + j = bit j - sh of nz
    const u32 colbits = (sh >= 0 ? (sh < 16 ? nz << sh : 0u) : (sh > -16 ? nz >> (-sh) : 0u)) & 0xffffu;
    const u32 mb = __brev(colbits) >> 16;                               // column nlo + j is z = z0 + 15 - j
    return bx ? vb & mb : 0u;
}

// FLAT = false: D % 16 == 0, a group lies in one column; true: groups may straddle two columns (two pixels, split at voxel `bnd`)
// C = 3: rgb_hw3 is the (h, w, 3) colour image; C = 1: a (h, w) LABEL image, the output a 1-byte label volume (row N3: 16 bytes per lane)
template <bool FLAT, int C>
__global__ __launch_bounds__(256) void k_global_carve90s(const u8* __restrict__ bin_hw, const u8* __restrict__ rgb_hw3, u8* __restrict__ out_slab,
                                                         const u32* __restrict__ vbits, int nw, int c0, i64 W, i64 H, i64 D, i64 x_first,
                                                         i64 ngroups, pb3d_magic mD, pb3d_magic mH, int small) {
    __shared__ u32x4 stage[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (i64 gw0 = (i64)blockIdx.x * blockDim.x + 64 * wv; gw0 < ngroups; gw0 += (i64)gridDim.x * blockDim.x) {   // wave-uniform
        const i64 g = gw0 + lane;
        u32x4 r[3] = {(u32x4)(0u), (u32x4)(0u), (u32x4)(0u)};
        if (g < ngroups) {
            i64 xy, xr, y;
            if (small) { const u32 q = pb3d_div((u32)(16 * g), mD), qx = pb3d_div(q, mH); xy = q; xr = qx; y = q - qx * mH.d; }
            else { xy = (16 * g) / D; xr = xy / H; y = xy - xr * H; }
            const i64 x = x_first + xr;
            const int z0 = (int)(16 * g - xy * D);
            const i64 bnd = FLAT ? (xy + 1) * D - 16 * g : 16;    // voxels of the group that belong to column xy (>= 16: all)
            const u8* brow = bin_hw + y * W;
            const u8* px = rgb_hw3 + (y * W + x) * C;
            const u32 cr = px[0], cg = C == 3 ? px[1] : 0u, cb = C == 3 ? px[2] : 0u;
            const u32 k1 = gc90_bits(brow, vbits + x * nw, x, z0, c0, (int)W);       // (validity bits past D are zero: a straddling group masks itself)
            if constexpr (C == 1) {
                u32 k2 = 0, lab2 = 0;
                if (FLAT && bnd < 16) {
                    const i64 x1 = y + 1 < H ? x : x + 1, y1 = y + 1 < H ? y + 1 : 0;
                    const u8* brow1 = bin_hw + y1 * W;
                    lab2 = rgb_hw3[y1 * W + x1];
                    k2 = (gc90_bits(brow1, vbits + x1 * nw, x1, 0, c0, (int)W) << bnd) & 0xffffu;
                }
                const u32 l1 = cr * 0x01010101u, l2 = lab2 * 0x01010101u;
                u32x4 o;
                o.x = (spread4(k1 & 15u) & l1) | (spread4(k2 & 15u) & l2);
                o.y = (spread4((k1 >> 4) & 15u) & l1) | (spread4((k2 >> 4) & 15u) & l2);
                o.z = (spread4((k1 >> 8) & 15u) & l1) | (spread4((k2 >> 8) & 15u) & l2);
                o.w = (spread4(k1 >> 12) & l1) | (spread4(k2 >> 12) & l2);
                __builtin_nontemporal_store(o, (u32x4_u*)(out_slab + 16 * g));
                continue;
            }
            u32 w[12];
            expand16(k1, cr, cg, cb, w);
            if (FLAT && bnd < 16) {
                const i64 x1 = y + 1 < H ? x : x + 1, y1 = y + 1 < H ? y + 1 : 0;        // the next column in (x, y) order
                const u8* brow1 = bin_hw + y1 * W;
                const u8* qx = rgb_hw3 + (y1 * W + x1) * 3;
                const u32 qr = qx[0], qg = qx[1], qb = qx[2];
                const u32 k2 = (gc90_bits(brow1, vbits + x1 * nw, x1, 0, c0, (int)W) << bnd) & 0xffffu;
                u32 w2[12];
                expand16(k2, qr, qg, qb, w2);
#pragma unroll
                for (int k = 0; k < 12; ++k) w[k] |= w2[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { r[k].x = w[4 * k]; r[k].y = w[4 * k + 1]; r[k].z = w[4 * k + 2]; r[k].w = w[4 * k + 3]; }
        }
        if constexpr (C == 3) store48_wave((u32x4_u*)out_slab, gw0, ngroups, r, stage[wv]);
    }
}

// the last nvox % 16 voxels of the slab (and grids with D < 16), voxel by voxel
__global__ __launch_bounds__(256) void k_global_carve90_generic(const u8* __restrict__ bin_hw, const u8* __restrict__ rgb_hw3, u8* __restrict__ out_slab,
                                                                const u32* __restrict__ vbits, int nw, int c0, i64 W, i64 H, i64 D, i64 x_first,
                                                                i64 v_first, i64 nvox, int C) {
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x) {
        const i64 xy = v / D, z = v - xy * D;
        const i64 xr = xy / H, y = xy - xr * H, x = x_first + xr;
        const i64 n = (i64)c0 - z;
        const bool on = bin_hw[y * W + x] && ((vbits[x * nw + (z >> 5)] >> (z & 31)) & 1u) && n >= 0 && n < W && bin_hw[y * W + n];
        const u8* px = rgb_hw3 + (y * W + x) * C;
        for (int c = 0; c < C; ++c) out_slab[C * v + c] = on ? px[c] : (u8)0;
    }
}

__global__ __launch_bounds__(256) void k_color_apply_generic(const u8* __restrict__ carved, const u8* __restrict__ rgb_hw3,
                                                             u8* __restrict__ out, i64 W, i64 H, i64 D, i64 v_first) {
    const i64 nvox = W * H * D;
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x) {
        const i64 xy = v / D;
        const i64 x = xy / H, y = xy - x * H;
        const u8* px = rgb_hw3 + (y * W + x) * 3;
        const bool on = carved[v] == 1;
        out[3 * v] = on ? px[0] : (u8)0;
        out[3 * v + 1] = on ? px[1] : (u8)0;
        out[3 * v + 2] = on ? px[2] : (u8)0;
    }
}

// ------------------------------------------------------------------------------------------------
// part_carve helpers.
//   k_part_occ   : occ[v] = mask_sub[xy] && any(colored[v] > 0)            (reference :152-154)
//   k_keep_or    : keep[v] |= carved[v] && mask_sub[xy]                    (part = sub * carved, :156)
//   k_part_final : out[v] = keep[v] ? colored[v] : 0                       (:158; every job writes
//                  colored[v] itself, so the overlay is the union of the jobs' keep sets)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_part_occ(const u8* __restrict__ colored, const u8* __restrict__ mask_sub,
                                                  u8* __restrict__ occ, i64 nvox, i64 D, i64 v_first = 0) {
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x) {
        const i64 xy = v / D;
        u8 r = 0;
        if (mask_sub[xy]) r = (colored[3 * v] | colored[3 * v + 1] | colored[3 * v + 2]) ? 1 : 0;
        occ[v] = r;
    }
}

__global__ __launch_bounds__(256) void k_keep_or(const u8* __restrict__ carved, const u8* __restrict__ mask_sub,
                                                 u8* __restrict__ keep, i64 nvox, i64 D, int first, i64 v_first = 0) {
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x) {
        const i64 xy = v / D;
        // part = (colored*m)*carved in uint8: non-zero iff m, carved (0/1 here) and colour non-zero
        const u8 k = (mask_sub[xy] && carved[v]) ? 1 : 0;
        keep[v] = first ? k : (u8)(keep[v] | k);
    }
}

__global__ __launch_bounds__(256) void k_part_final(const u8* __restrict__ colored, const u8* __restrict__ keep,
                                                    u8* __restrict__ out, i64 nvox, i64 v_first = 0) {
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x) {
        const bool k = keep[v] != 0;
        out[3 * v] = k ? colored[3 * v] : (u8)0;
        out[3 * v + 1] = k ? colored[3 * v + 1] : (u8)0;
        out[3 * v + 2] = k ? colored[3 * v + 2] : (u8)0;
    }
}

// 16-voxel-per-lane forms of the three kernels above (16-byte aligned buffers, D >= 16, fewer than 2^32 voxels): the volume is a
// flat stream of 16-voxel groups; a group lies in one (x,y) column or -- when D is not a multiple of 16 -- straddles two, so the 2-D
// mask is read once or twice per group and skipped groups cost no grid read.  The stream's last nvox % 16 voxels go through the
// scalar kernels (v_first).
__device__ __forceinline__ u32 group_sel16(const u8* __restrict__ mask_sub, i64 g, const pb3d_magic mD) {
    const u32 v0 = (u32)(16 * g);
    const u32 col0 = pb3d_div(v0, mD), nfirst = mD.d - (v0 - col0 * mD.d);          // voxels of the group in its first column
    u32 sel = mask_sub[col0] ? 0xffffu : 0u;
    if (nfirst < 16u) { const u32 lowm = (1u << nfirst) - 1u; sel = (sel & lowm) | (mask_sub[col0 + 1] ? (0xffffu & ~lowm) : 0u); }
    return sel;
}

__global__ __launch_bounds__(256) void k_part_occ16(const u32x4* __restrict__ colored, const u8* __restrict__ mask_sub,
                                                    u32x4* __restrict__ occ, i64 ngroups, pb3d_magic mD) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        u32 o[4] = {0, 0, 0, 0};
        const u32 sel = group_sel16(mask_sub, g, mD);
        if (sel) {
            u32 w[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) { const u32x4 t = ld_s(colored + 3 * g + k); w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const u32 any = byte_of(w, 3 * i) | byte_of(w, 3 * i + 1) | byte_of(w, 3 * i + 2);
                o[i >> 2] |= (any ? 1u : 0u) << ((i & 3) * 8);
            }
            if (sel != 0xffffu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] &= spread4((sel >> (4 * j)) & 15u);
            }
        }
        u32x4 r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
        st_nt(occ + g, r);
    }
}

__global__ __launch_bounds__(256) void k_keep_or16(const u32x4* __restrict__ carved, const u8* __restrict__ mask_sub,
                                                   u32x4* __restrict__ keep, i64 ngroups, pb3d_magic mD, int first) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        u32x4 k = (u32x4)(0u);
        const u32 sel = group_sel16(mask_sub, g, mD);
        if (sel) {
            const u32x4 c = carved[g];
            // per byte: non-zero -> 1 (carved is 0/1 on this path, but stay exact for any byte value)
            const u32 cw[4] = {c.x, c.y, c.z, c.w};
            u32 kw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 t = cw[j];
                t |= t >> 4; t |= t >> 2; t |= t >> 1;   // fold every byte's bits into its bit 0
                kw[j] = t & 0x01010101u & spread4((sel >> (4 * j)) & 15u);
            }
            k.x = kw[0]; k.y = kw[1]; k.z = kw[2]; k.w = kw[3];
        }
        if (!first) { const u32x4 old = keep[g]; k.x |= old.x; k.y |= old.y; k.z |= old.z; k.w |= old.w; }
        keep[g] = k;
    }
}

__global__ __launch_bounds__(256) void k_part_final16(const u32x4* __restrict__ colored, const u32x4* __restrict__ keep,
                                                      u32x4* __restrict__ out, i64 ngroups) {
    __shared__ u32x4 stage[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (i64 gw0 = (i64)blockIdx.x * blockDim.x + 64 * wv; gw0 < ngroups; gw0 += (i64)gridDim.x * blockDim.x) {   // wave-uniform
        const i64 g = gw0 + lane;
        u32x4 r[3] = {(u32x4)(0u), (u32x4)(0u), (u32x4)(0u)};
        if (g < ngroups) {
            const u32x4 kv = ld_nt(keep + g);
            const u32 kw[4] = {kv.x, kv.y, kv.z, kv.w};
            u32 keep16 = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) keep16 |= (byte_of(kw, i) ? 1u : 0u) << i;
            if (keep16) {
                u32 m[12];
                expand16(keep16, 0xffu, 0xffu, 0xffu, m);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const u32x4 c = ld_s(colored + 3 * g + k);
                    r[k].x = c.x & m[4 * k]; r[k].y = c.y & m[4 * k + 1]; r[k].z = c.z & m[4 * k + 2]; r[k].w = c.w & m[4 * k + 3];
                }
            }
        }
        store48_wave(out, gw0, ngroups, r, stage[wv]);
    }
}

// out[v] = colored[v] where keep[v] (the rest of `out` -- the overlay of the fused 90-degree jobs -- stays)
__global__ __launch_bounds__(256) void k_part_merge(const u8* __restrict__ colored, const u8* __restrict__ keep, u8* __restrict__ out, i64 nvox,
                                                    i64 v_first = 0) {
    for (i64 v = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x)
        if (keep[v]) { out[3 * v] = colored[3 * v]; out[3 * v + 1] = colored[3 * v + 1]; out[3 * v + 2] = colored[3 * v + 2]; }
}

__global__ __launch_bounds__(256) void k_part_merge16(const u32x4* __restrict__ colored, const u32x4* __restrict__ keep, u32x4* __restrict__ out,
                                                      i64 ngroups) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const u32x4 kv = ld_nt(keep + g);
        const u32 kw[4] = {kv.x, kv.y, kv.z, kv.w};
        u32 keep16 = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) keep16 |= (byte_of(kw, i) ? 1u : 0u) << i;
        if (!keep16) continue;                               // untouched groups cost 16 bytes of keep
        u32 m[12];
        expand16(keep16, 0xffu, 0xffu, 0xffu, m);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const u32x4 c = ld_s(colored + 3 * g + k), o = ld_s(out + 3 * g + k);
            u32x4 r;
            r.x = (c.x & m[4 * k]) | (o.x & ~m[4 * k]); r.y = (c.y & m[4 * k + 1]) | (o.y & ~m[4 * k + 1]);
            r.z = (c.z & m[4 * k + 2]) | (o.z & ~m[4 * k + 2]); r.w = (c.w & m[4 * k + 3]) | (o.w & ~m[4 * k + 3]);
            st_s(out + 3 * g + k, r);
        }
    }
}

// Jobs with angle steps other than 90, merged in ONE pass: job k of the list left its carved occupancy in carvedN + k * nvox, S[xy]
// has bit k where mask_sub_k[xy] is set; keep[v] = OR_k (carved_k[v] && S[xy] bit k) -- the per-job keep-OR passes (3 B/voxel each) and
// the keep volume are gone.  MERGE: out[v] = colored[v] where keep (the overlay of the fused 90-degree jobs stays); else the final
// out[v] = keep ? colored[v] : 0.
struct JobList { int j[8]; int n; };

__global__ __launch_bounds__(256) void k_sub_bitset(const u8* __restrict__ mask_sub, JobList jl, i64 npix, u32* __restrict__ S) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (i64)gridDim.x * blockDim.x) {
        u32 a = 0;
        for (int k = 0; k < jl.n; ++k)
            if (mask_sub[(i64)jl.j[k] * npix + i]) a |= 1u << k;
        S[i] = a;
    }
}

// 16 keep bits of group g (the volume as a flat stream of 16-voxel groups; a group lies in one (x,y) column or straddles two)
__device__ __forceinline__ u32 multi_keep16(const u32x4* __restrict__ carvedN, i64 ngroups_stride, int nk, const u32* __restrict__ S, i64 g,
                                            const pb3d_magic mD) {
    const u32 v0 = (u32)(16 * g);
    const u32 col0 = pb3d_div(v0, mD), nfirst = mD.d - (v0 - col0 * mD.d);
    const u32 s0 = S[col0], s1 = nfirst < 16u ? S[col0 + 1] : 0u;
    const u32 lowm = nfirst < 16u ? (1u << nfirst) - 1u : 0xffffu;
    u32 keep16 = 0;
    for (int k = 0; k < nk; ++k) {
        const u32 sel = (((s0 >> k) & 1u) ? lowm : 0u) | (((s1 >> k) & 1u) ? (0xffffu & ~lowm) : 0u);
        if (!sel) continue;
        const u32x4 c = ld_nt(carvedN + (i64)k * ngroups_stride + g);
        const u32 cw[4] = {c.x, c.y, c.z, c.w};
        u32 c16 = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c16 |= (byte_of(cw, i) ? 1u : 0u) << i;
        keep16 |= c16 & sel;
    }
    return keep16;
}

template <bool MERGE>
__global__ __launch_bounds__(256) void k_part_multi16(const u32x4* __restrict__ colored, const u32x4* __restrict__ carvedN, i64 ngroups_stride, int nk,
                                                      const u32* __restrict__ S, pb3d_magic mD, u32x4* __restrict__ out, i64 ngroups) {
    __shared__ u32x4 stage[4][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (i64 gw0 = (i64)blockIdx.x * blockDim.x + 64 * wv; gw0 < ngroups; gw0 += (i64)gridDim.x * blockDim.x) {   // wave-uniform
        const i64 g = gw0 + lane;
        u32x4 r[3] = {(u32x4)(0u), (u32x4)(0u), (u32x4)(0u)};
        u32 keep16 = 0;
        if (g < ngroups) keep16 = multi_keep16(carvedN, ngroups_stride, nk, S, g, mD);
        if (keep16) {
            u32 m[12];
            expand16(keep16, 0xffu, 0xffu, 0xffu, m);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const u32x4 c = ld_s(colored + 3 * g + k);
                if (MERGE) {
                    const u32x4 o = ld_s(out + 3 * g + k);
                    r[k].x = (c.x & m[4 * k]) | (o.x & ~m[4 * k]); r[k].y = (c.y & m[4 * k + 1]) | (o.y & ~m[4 * k + 1]);
                    r[k].z = (c.z & m[4 * k + 2]) | (o.z & ~m[4 * k + 2]); r[k].w = (c.w & m[4 * k + 3]) | (o.w & ~m[4 * k + 3]);
                    st_s(out + 3 * g + k, r[k]);
                } else {
                    r[k].x = c.x & m[4 * k]; r[k].y = c.y & m[4 * k + 1]; r[k].z = c.z & m[4 * k + 2]; r[k].w = c.w & m[4 * k + 3];
                }
            }
        }
        if (!MERGE) store48_wave(out, gw0, ngroups, r, stage[wv]);
    }
}

// ------------------------------------------------------------------------------------------------
// part_carve with 90-degree jobs, PLANE-LOCAL form (round 4; reference utils/voxel_carving_utils.py:139-160).
//   keep[x,y,z] = valid(x,z) && occ[c0 - z, y, x + c2] && (A[x,y] & A[c0 - z, y]) != 0,   out = keep ? colored : 0
// (csrc/rotate_tiled.hip, K5).  The fused tile kernels transpose 128 x 128 BYTE tiles of occupancy through LDS, two barriers per plane
// and 384-byte row pieces on both sides; on rows that are not whole lines (the reference's real shapes) they run at 2.2 - 2.8 TB/s.  But
// only the occupancy BIT has to be transposed, and a Y-plane of bits is a few KB.  One workgroup = (plane y, 128 output rows x0 ..):
//   A  the 128 source columns x0 + c2 .. of EVERY source row of the plane (384 contiguous bytes per row, 48 per thread) -> 16 occupancy
//      bits per thread -> S[n0][128 bits] in LDS; meanwhile the job sets of the plane's image row become bit rows over z (ballots)
//   C  32 x 32 bit blocks of S transposed in registers -> the raw keep rows T[x][z]
//   D  T & validity bits & job-match rows -> K[x][z bits] (LDS)
//   E  the 128 output rows leave as whole rows of 3 D contiguous bytes: 16-byte pieces, colours read only where a piece keeps something
// Every voxel is read once as a source and at most once more as a kept destination, written once: the 6 B/voxel of the fused sweep,
// without a byte transposition, in rows of >= 1 KB on the store side, for any D, any alignment, W != D.
// ------------------------------------------------------------------------------------------------
// 32 x 32 bit transpose in registers (LSB convention): out[c] bit r = in[r] bit c
__device__ __forceinline__ void transpose32(u32 a[32]) {
    u32 m = 0x0000ffffu;
#pragma unroll
    for (int j = 16; j; j >>= 1, m ^= m << j) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const u32 t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j; a[k + j] ^= t;
        }
    }
}

typedef u32 u32_ua __attribute__((aligned(1)));

// C = 3: `colored` is the (W,H,D,3) colour grid; C = 1: a 1-byte LABEL volume (row N3: occupancy = label != 0, a piece is 16 voxels).
template <int C, int UA, int UE>
__global__ __launch_bounds__(256) void k_part90_plane(const u8* __restrict__ colored, const u32* __restrict__ A, const u32* __restrict__ AT,
                                                      const u32* __restrict__ vbits, int nwv, int c0, int c2, i64 W, i64 H, i64 D, int nwz, int njobs,
                                                      pb3d_magic mP, u8* __restrict__ out) {
    extern __shared__ u32 sm_plane[];
    __shared__ u32x4 mtab[192];
    u32* S = sm_plane;                                  // [W][4]: bit b of row n0 = occ[n0, y, x0 + c2 + b]
    unsigned short* S16 = (unsigned short*)S;
    u32* Jb = S + 4 * W;                                // [32][nwz]: bit z of row j = job j at image pixel (c0 - z, y)
    u32* Kl = Jb + 32 * nwz;                            // [128][nwz + 1]: keep bits of output row x0 + r over z
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const i64 y = blockIdx.y, x0 = (i64)blockIdx.x * 128;
    const int pitch = nwz + 1;
    const i64 nvox = W * H * D;
    // ---- the job bit rows: 64 z per ballot
    for (int kk = wv; 64 * kk < D; kk += 4) {
        const i64 z = 64 * (i64)kk + lane, n = (i64)c0 - z;
        const u32 a = (z < D && n >= 0 && n < W) ? AT[y * W + n] : 0u;
        for (int j = 0; j < njobs; ++j) {
            const u64 b = __ballot((a >> j) & 1u);
            if (lane == 0) { Jb[j * nwz + 2 * kk] = (u32)b; if (2 * kk + 1 < nwz) Jb[j * nwz + 2 * kk + 1] = (u32)(b >> 32); }
        }
    }
    for (int r = tid; r < 128; r += 256) Kl[r * pitch + nwz] = 0u;
    // the byte masks of phase E, one 16-byte entry per (channel phase, 6 keep bits): entry of piece phase ph and bits kb6 = bytes 0xff where
    // the voxel that owns the byte is kept (a piece starts at channel ph of its first voxel)
    if (tid < 192) {
        const u32 ph = (u32)tid >> 6, kb6 = (u32)tid & 63u;
        u32 m[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) m[e] = 0u - ((kb6 >> e) & 1u);
        const u32 w0 = (m[0] & 0x00ffffffu) | (m[1] & 0xff000000u), w1 = (m[1] & 0x0000ffffu) | (m[2] & 0xffff0000u),
                  w2 = (m[2] & 0x000000ffu) | (m[3] & 0xffffff00u), w3 = (m[4] & 0x00ffffffu) | (m[5] & 0xff000000u),
                  w4 = m[5] & 0x0000ffffu;
        u32x4 e4;
        e4.x = __builtin_amdgcn_alignbyte(w1, w0, ph); e4.y = __builtin_amdgcn_alignbyte(w2, w1, ph);
        e4.z = __builtin_amdgcn_alignbyte(w3, w2, ph); e4.w = __builtin_amdgcn_alignbyte(w4, w3, ph);
        mtab[tid] = e4;
    }
    // ---- A: occupancy bits of the source columns.  Four items per thread and pass: all their loads are in flight before the first is used
    // (one item at a time, a wave had 48 bytes per lane in flight and the pass ran at the memory's latency, not its bandwidth)
    for (int it0 = tid; it0 < 8 * (int)W; it0 += 256 * UA) {
        u32 w[UA][C == 3 ? 12 : 4];
        int mode[UA];                                                       // 0: nothing to read, 1: whole (loaded here), 2: ragged (read byte-wise below)
        i64 vv[UA], nn2[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int it = it0 + 256 * u;
            const i64 n0 = it >> 3;
            const i64 n2 = x0 + c2 + 16 * (it & 7);                        // first source column of this item's 16
            nn2[u] = n2; vv[u] = (n0 * H + y) * D + n2;
            // 16 voxels that run over a row end are read whole while they stay inside the volume (the foreign ones are masked off below):
            // a byte loop here is executed by every wave that holds ONE such item -- all of them on a 355-wide grid
            mode[u] = (it < 8 * (int)W && n2 + 15 >= 0 && n2 < D) ? ((vv[u] >= 0 && vv[u] + 16 <= nvox) ? 1 : 2) : 0;
            if (mode[u] == 1) {
                const u32x4_u* g = (const u32x4_u*)(colored + C * vv[u]);
#pragma unroll
                for (int q = 0; q < C; ++q) { const u32x4 t = g[q]; w[u][4 * q] = t.x; w[u][4 * q + 1] = t.y; w[u][4 * q + 2] = t.z; w[u][4 * q + 3] = t.w; }
            }
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int it = it0 + 256 * u;
            if (it >= 8 * (int)W) continue;
            u32 bits = 0;
            if (mode[u]) {
                if (mode[u] == 2) {
#pragma unroll
                    for (int q = 0; q < 4 * C; ++q) w[u][q] = 0u;
                    for (int b2 = 0; b2 < 16 * C; ++b2) { const i64 col = nn2[u] + b2 / C; if (col >= 0 && col < D) w[u][b2 >> 2] |= (u32)colored[C * vv[u] + b2] << (8 * (b2 & 3)); }
                }
                // any(colour > 0) of voxel i = (sum of its three bytes) != 0: v_dot4_u32_u8 against 0x00010101 sums three bytes of a dword
                if constexpr (C == 3) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = (3 * i) >> 2, sh = (3 * i) & 3;
                        const u32 v3 = sh == 0 ? w[u][j] : __builtin_amdgcn_alignbyte(j + 1 < 12 ? w[u][j + 1 < 12 ? j + 1 : 0] : 0u, w[u][j], (u32)sh);
                        const u32 sum = __builtin_amdgcn_udot4(v3, 0x00010101u, 0u, false);
                        bits |= (sum < 1u ? sum : 1u) << i;
                    }
                } else bits = nonzero16(w[u]);
                const int ilo = nn2[u] < 0 ? (int)(-nn2[u]) : 0, ihi = D - nn2[u] < 16 ? (int)(D - nn2[u]) : 16;     // the columns that exist: [ilo, ihi)
                bits &= ((1u << ihi) - 1u) & ~((1u << ilo) - 1u);
            }
            S16[it] = (unsigned short)bits;
        }
    }
    __syncthreads();
    // ---- C: 32 x 32 blocks: rows i = source rows c0 - 32 k - i (z = 32 k + i), columns = 32 output rows
    for (int it = tid; it < 4 * nwz; it += 256) {
        const int k = it >> 2, jb = it & 3;
        u32 a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const i64 n0 = (i64)c0 - 32 * k - i;
            a[i] = (n0 >= 0 && n0 < W) ? S[4 * n0 + jb] : 0u;
        }
        transpose32(a);
#pragma unroll
        for (int xx = 0; xx < 32; ++xx) Kl[(32 * jb + xx) * pitch + k] = a[xx];
    }
    __syncthreads();
    // ---- D: validity and job match
    for (int it = tid; it < 128 * nwz; it += 256) {
        const int xl = it / nwz, k = it - xl * nwz;
        const i64 x = x0 + xl;
        u32 kd = 0;
        const u32 t = Kl[xl * pitch + k];
        if (x < W && t) {
            const u32 vb = vbits[x * nwv + k];
            u32 aj = A[x * H + y], M = 0;
            while (aj) { const int j = __ffs((int)aj) - 1; M |= Jb[j * nwz + k]; aj &= aj - 1; }
            kd = t & vb & M;
        }
        Kl[xl * pitch + k] = kd;
    }
    __syncthreads();
    // ---- E: the output rows, 16-byte pieces (piece pc of a row: voxels from 16 pc / 3 on, channel phase pc % 3)
    const int npieces = (int)((C * D + 15) / 16);
    const int nrows = (int)(W - x0 < 128 ? W - x0 : 128);
    const int total = nrows * npieces;
    for (int it0 = tid; it0 < total; it0 += 256 * UE) {                        // four pieces per thread and pass, their loads in flight together
        i64 rb[UE];
        u32 kb6[UE], ph[UE];
        int nb[UE];
        u32x4 src[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const int it = it0 + 256 * u;
            kb6[u] = 0; nb[u] = 0; rb[u] = 0; ph[u] = 0; src[u] = (u32x4)(0u);
            if (it < total) {
                const int xl = (int)pb3d_div((u32)it, mP), pc = it - xl * npieces;
                const i64 x = x0 + xl;
                rb[u] = ((x * H + y) * D) * C + 16 * (i64)pc;              // byte offset of the piece
                const int v0 = C == 3 ? (16 * pc) / 3 : 16 * pc;
                ph[u] = C == 3 ? (u32)(pc % 3) : 0u;
                const u32* kr = Kl + xl * pitch + (v0 >> 5);
                const u32 k0 = kr[0], k1 = (v0 >> 5) + 1 <= nwz ? kr[1] : 0u;
                kb6[u] = (u32)((((u64)k1 << 32) | (u64)k0) >> (v0 & 31)) & (C == 3 ? 0x3fu : 0xffffu);      // keep bits of the piece's 6 / 16 voxels
                nb[u] = 16 * pc + 16 <= C * D ? 16 : (int)(C * D - 16 * pc);      // bytes of this piece (the row's last one may be short)
                if (kb6[u]) {
                    if (nb[u] == 16 || rb[u] + 16 <= nvox * C) src[u] = *(const u32x4_u*)(colored + rb[u]);
                    else {
                        u32 t4[4] = {0, 0, 0, 0};
                        for (int b2 = 0; b2 < nb[u]; ++b2) t4[b2 >> 2] |= (u32)colored[rb[u] + b2] << (8 * (b2 & 3));
                        src[u].x = t4[0]; src[u].y = t4[1]; src[u].z = t4[2]; src[u].w = t4[3];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            if (!nb[u]) continue;
            u32x4 val = (u32x4)(0u);
            if (kb6[u]) {
                u32x4 mk;
                if constexpr (C == 3) mk = mtab[64 * ph[u] + kb6[u]];
                else { mk.x = spread4(kb6[u] & 15u); mk.y = spread4((kb6[u] >> 4) & 15u); mk.z = spread4((kb6[u] >> 8) & 15u); mk.w = spread4(kb6[u] >> 12); }
                val.x = src[u].x & mk.x; val.y = src[u].y & mk.y; val.z = src[u].z & mk.z; val.w = src[u].w & mk.w;
            }
            if (nb[u] == 16) __builtin_nontemporal_store(val, (u32x4_u*)(out + rb[u]));
            else {
                const u32 t4[4] = {val.x, val.y, val.z, val.w};
                u8* op = out + rb[u];
                for (int jj = 0; jj < (nb[u] >> 2); ++jj) *(u32_ua*)(op + 4 * jj) = t4[jj];
                for (int b2 = nb[u] & ~3; b2 < nb[u]; ++b2) op[b2] = (u8)(t4[b2 >> 2] >> (8 * (b2 & 3)));
            }
        }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

extern "C" {

int pb3d_carve_mask_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t W, int64_t H, int64_t D, int C,
                        const uint8_t* d_mask_wh, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_carve_mask: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0 && (C == 1 || C == 3), "pb3d_carve_mask: bad shape (%lld,%lld,%lld,%d)",
                 (long long)W, (long long)H, (long long)D, C);
    const i64 ncols = W * H, col = D * C, nbytes = ncols * col;
    if (nbytes == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid && d_mask_wh && d_out, "pb3d_carve_mask: null buffer");
    PB3D_REQUIRE(d_grid != d_out, "pb3d_carve_mask: in-place carve is not supported");
    const bool fast = (col % 16 == 0) && aligned16(d_grid) && aligned16(d_out) && (nbytes / 16) / kTileVec < (1ll << 31);
    if (fast) {
        const i64 nvec = nbytes / 16;
        const u32 vpc = (u32)(col / 16 < (1ll << 31) ? col / 16 : 0);
        PB3D_REQUIRE(vpc != 0, "pb3d_carve_mask: column too long");
        // (rem0 + lv) / vpc with rem0 + lv < vpc + 768 through one umulhi: exact while (vpc + 768) * vpc < 2^32
        u32 magic = 0;
        if (vpc > 1 && ((u64)vpc + kTileVec) * vpc < (1ull << 32)) magic = (u32)(((1ull << 32) + vpc - 1) / vpc);
        const unsigned blocks = (unsigned)((nvec + kTileVec - 1) / kTileVec);
        hipLaunchKernelGGL(k_carve_tiles, dim3(blocks), dim3(kTileThreads), 0, ctx->stream, (const u32x4*)d_grid,
                           (u32x4*)d_out, d_mask_wh, nvec, ncols, vpc, magic);
    } else if (col >= 16 && col < (1ll << 31) - kTileVec * 16 && (nbytes + kTileVec * 16 - 1) / (kTileVec * 16) < (1ll << 31)) {
        const unsigned blocks = (unsigned)((nbytes + kTileVec * 16 - 1) / (kTileVec * 16));
        hipLaunchKernelGGL(k_carve_flat, dim3(blocks), dim3(kTileThreads), 0, ctx->stream, d_grid, d_out, d_mask_wh, nbytes, ncols, (u32)col,
                           make_magic32((u32)col));
    } else {
        const unsigned blocks = pb3d_stream_blocks(ctx, (nbytes + 15) / 16, 256, 8);
        hipLaunchKernelGGL(k_carve_bytes, dim3(blocks), dim3(256), 0, ctx->stream, d_grid, d_out, d_mask_wh, nbytes, col);
    }
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_occupancy_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t nvox, uint8_t* d_occ) {
    PB3D_REQUIRE(ctx != nullptr && nvox >= 0, "pb3d_occupancy: bad argument");
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_occ, "pb3d_occupancy: null buffer");
    i64 done = 0;
    if (aligned16(d_grid_rgb) && aligned16(d_occ) && nvox >= 16) {
        const i64 ngroups = nvox / 16;
        hipLaunchKernelGGL(k_occupancy16, dim3(pb3d_stream_blocks(ctx, ngroups, 256, 0)), dim3(256), 0, ctx->stream,
                           (const u32x4*)d_grid_rgb, (u32x4*)d_occ, ngroups);
        PB3D_CHECK_LAUNCH();
        done = ngroups * 16;
    }
    if (done < nvox) {
        hipLaunchKernelGGL(k_occupancy_tail, dim3(pb3d_stream_blocks(ctx, nvox - done, 256, 8)), dim3(256), 0, ctx->stream,
                           d_grid_rgb, d_occ, done, nvox);
        PB3D_CHECK_LAUNCH();
    }
    return PB3D_OK;
}

}  // extern "C"

// global_carve(., ., 90) on the slab x in [x0, x1): the stream kernel above (d_vbits: the validity table of the 90-degree step)
int pb3d_launch_gc90_stream(pb3d_ctx* ctx, const u8* d_bin_hw, const u8* d_rgb_hw3, int C, const u32* d_vbits, int nw, int c0, i64 W, i64 H, i64 D, i64 x0,
                            i64 x1, u8* d_out_slab) {
    const i64 nvox = (x1 - x0) * H * D;
    const i64 ngroups = (D >= 16 && W >= 16) ? nvox / 16 : 0;
    if (ngroups) {
        auto kern = C == 3 ? (D % 16 == 0 ? k_global_carve90s<false, 3> : k_global_carve90s<true, 3>) : (D % 16 == 0 ? k_global_carve90s<false, 1> : k_global_carve90s<true, 1>);
        hipLaunchKernelGGL(kern, dim3(pb3d_stream_blocks(ctx, ngroups, 256, 0)), dim3(256), 0, ctx->stream, d_bin_hw, d_rgb_hw3, d_out_slab,
                           d_vbits, nw, c0, W, H, D, x0, ngroups, pb3d_make_magic((u32)(D < (1ll << 31) ? D : 1)), pb3d_make_magic((u32)(H < (1ll << 31) ? H : 1)),
                           (nvox < (1ll << 32) && D < (1ll << 31) && H < (1ll << 31)) ? 1 : 0);
        PB3D_CHECK_LAUNCH();
    }
    if (16 * ngroups < nvox) {
        hipLaunchKernelGGL(k_global_carve90_generic, dim3(pb3d_stream_blocks(ctx, nvox - 16 * ngroups, 256, 8)), dim3(256), 0, ctx->stream, d_bin_hw, d_rgb_hw3,
                           d_out_slab, d_vbits, nw, c0, W, H, D, x0, 16 * ngroups, nvox, C);
        PB3D_CHECK_LAUNCH();
    }
    return PB3D_OK;
}

// part_carve with 90-degree jobs, plane-local form.  d_A: job sets in (x, y) order, d_AT: the same in (y, x) order, njobs: highest job + 1.
// *took = 0: shape outside the path's limits (the caller runs the fused tile kernels).
int pb3d_part_carve90_planes(pb3d_ctx* ctx, const u8* d_colored, int C, i64 W, i64 H, i64 D, const u32* d_A, const u32* d_AT, int njobs, const u32* d_vbits,
                             int nwv, int c0, int c2, u8* d_out, int* took) {
    *took = 0;
    const int nwz = (int)((D + 31) / 32);
    const size_t lds = ((size_t)4 * W + (size_t)32 * nwz + (size_t)128 * (nwz + 1)) * sizeof(u32);
    const i64 npieces = (C * D + 15) / 16;
    if (D < 1 || W < 1 || lds > 150 * 1024 || H > 65535 || W > (1 << 20) || 128 * npieces >= (1ll << 31)) return PB3D_OK;
    if (lds > 60 * 1024 && !ctx->part90_lds_set) {              // (W beyond ~2900: the plane's bits need more than the default 64 KB)
        PB3D_HIP(hipFuncSetAttribute((const void*)k_part90_plane<3, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        PB3D_HIP(hipFuncSetAttribute((const void*)k_part90_plane<1, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ctx->part90_lds_set = true;
    }
    // items in flight per thread in the source pass / the output pass: 2 / 4 (tools/tybench.py --op part, the nine combinations of 1, 2, 4
    // interleaved on one box: best or tied at 512 x 278 x 512, 355 x 512 x 355, 512^3 and 1024^3; profiles/r04_part_carve_plane_kernel_unroll_sweep.jsonl)
    auto kern = C == 3 ? k_part90_plane<3, 2, 4> : k_part90_plane<1, 2, 4>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((W + 127) / 128), (unsigned)H), dim3(256), lds, ctx->stream, d_colored, d_A, d_AT, d_vbits, nwv, c0, c2,
                       W, H, D, nwz, njobs, pb3d_make_magic((u32)npieces), d_out);
    PB3D_CHECK_LAUNCH();
    *took = 1;
    return PB3D_OK;
}

extern "C" {

int pb3d_color_apply_dev(pb3d_ctx* ctx, const uint8_t* d_carved, int64_t W, int64_t H, int64_t D,
                         const uint8_t* d_rgb_hw3, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && W >= 0 && H >= 0 && D >= 0, "pb3d_color_apply: bad shape");
    const i64 nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_carved && d_rgb_hw3 && d_out, "pb3d_color_apply: null buffer");
    const i64 ngroups = D >= 16 ? nvox / 16 : 0;      // whole groups of 16 voxels of the flat stream; the rest goes voxel by voxel
    // one workgroup per 256 groups, not a persistent grid-stride loop: the dispatcher balances a write-heavy stream better (0.84 -> 0.70 ms)
    const unsigned ca_blocks = pb3d_stream_blocks(ctx, ngroups, 256, 0);
    if (ngroups) {
        if (D % 16 == 0)
            hipLaunchKernelGGL(k_color_apply16<false>, dim3(ca_blocks), dim3(256), 0, ctx->stream, d_carved,
                               d_rgb_hw3, d_out, W, H, D, ngroups, pb3d_make_magic((u32)(D < (1ll << 31) ? D : 1)), pb3d_make_magic((u32)(H < (1ll << 31) ? H : 1)),
                               (nvox < (1ll << 32) && D < (1ll << 31) && H < (1ll << 31)) ? 1 : 0);
        else
            hipLaunchKernelGGL(k_color_apply16<true>, dim3(ca_blocks), dim3(256), 0, ctx->stream, d_carved,
                               d_rgb_hw3, d_out, W, H, D, ngroups, pb3d_make_magic((u32)(D < (1ll << 31) ? D : 1)), pb3d_make_magic((u32)(H < (1ll << 31) ? H : 1)),
                               (nvox < (1ll << 32) && D < (1ll << 31) && H < (1ll << 31)) ? 1 : 0);
        PB3D_CHECK_LAUNCH();
    }
    if (16 * ngroups < nvox)
        hipLaunchKernelGGL(k_color_apply_generic, dim3(pb3d_stream_blocks(ctx, nvox - 16 * ngroups, 256, 8)), dim3(256), 0, ctx->stream,
                           d_carved, d_rgb_hw3, d_out, W, H, D, 16 * ngroups);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_part_carve_dev(pb3d_ctx* ctx, const uint8_t* d_colored, int64_t W, int64_t H, int64_t D,
                        const uint8_t* d_mask_sub, const uint8_t* d_mask_carve, const int* job_angle,
                        const int* job_skip, int njobs, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && W >= 0 && H >= 0 && D >= 0 && njobs >= 0, "pb3d_part_carve: bad shape");
    const i64 nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_colored && d_out && (njobs == 0 || (d_mask_sub && d_mask_carve && job_angle && job_skip)),
                 "pb3d_part_carve: null buffer");
    for (int j = 0; j < njobs; ++j)
        PB3D_REQUIRE(job_skip[j] || job_angle[j] > 0, "pb3d_part_carve: job %d has angle %d (must be > 0)", j, job_angle[j]);
    // Every job overlays colored[v] where it keeps, so the result is the union of the jobs' keep sets in any order: all
    // 90-degree jobs go through ONE fused sweep (csrc/rotate_tiled.hip, K5); jobs with other angles follow one by one and are
    // merged into that overlay.
    int n90 = 0, nother = 0;
    for (int j = 0; j < njobs; ++j)
        if (!job_skip[j]) { if (job_angle[j] == 90) ++n90; else ++nother; }
    bool base90 = false;                // d_out already holds the overlay of the 90-degree jobs
    std::vector<int> skip_rest(job_skip, job_skip + njobs);
    if (n90 > 0 && njobs <= 32) {
        std::vector<int> skip90(njobs);
        for (int j = 0; j < njobs; ++j) skip90[j] = job_skip[j] || job_angle[j] != 90;
        const int rc = pb3d_try_part_carve90(ctx, d_colored, 3, W, H, D, d_mask_sub, d_mask_carve, job_angle, skip90.data(), njobs, d_out);
        if (rc != PB3D_EUNSUPPORTED) {
            if (rc != PB3D_OK || nother == 0) return rc;
            base90 = true;
            for (int j = 0; j < njobs; ++j) skip_rest[j] = job_skip[j] || job_angle[j] == 90;
        }
    }
    job_skip = skip_rest.data();
    const bool wide = D >= 16 && nvox < (1ll << 32) && aligned16(d_colored) && aligned16(d_out);   // scratch buffers are 4 KiB aligned
    int nrest_all = 0;
    for (int j = 0; j < njobs; ++j) nrest_all += job_skip[j] ? 0 : 1;
    const bool multi = wide && nrest_all >= 1 && nrest_all <= 8 && nvox % 16 == 0 && ctx->tune_misc[3] != 2;
    // every slot is requested ONCE with the size its path needs: a regrow frees and reallocates (and drops the cached rotation tables,
    // among them the set the first job's prefetch is about to build)
    void *occ, *carved, *tmp, *keep;
    PB3D_TRY(pb3d_scratch(ctx, 4, (size_t)nvox, &occ));
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)nvox * (size_t)(multi ? nrest_all : 1), &carved));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)nvox, &tmp));
    PB3D_TRY(pb3d_scratch(ctx, 7, std::max((size_t)nvox, (size_t)(W * H) * sizeof(u32)), &keep));
    const unsigned blocks = pb3d_stream_blocks(ctx, nvox, 256, 8);
    const i64 ngroups = nvox / 16, vtail = 16 * ngroups;                    // the last nvox % 16 voxels: scalar kernels from vtail on
    const pb3d_magic mD = pb3d_make_magic((u32)(D > 0 ? D : 1));
    const unsigned gblocks = pb3d_stream_blocks(ctx, ngroups > 0 ? ngroups : 1, 256, 8);
    // up to eight such jobs: each leaves its carved occupancy in its own volume and ONE pass merges them (k_part_multi16); tune misc3 = 2
    // keeps the job-by-job keep-OR passes
    JobList jl; jl.n = 0;
    for (int j = 0; j < njobs; ++j)
        if (!job_skip[j] && jl.n < 8) jl.j[jl.n++] = j;
    if (multi) {
        void *carvedN = carved, *S = keep;
        for (int k = 0; k < jl.n; ++k) {
            const int j = jl.j[k];
            const u8* ms = d_mask_sub + (i64)j * W * H;
            const u8* mc = d_mask_carve + (i64)j * W * H;
            hipLaunchKernelGGL(k_part_occ16, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, ms, (u32x4*)occ, ngroups, mD);
            PB3D_CHECK_LAUNCH();
            PB3D_TRY(pb3d_process_grid_binary_dev(ctx, (const u8*)occ, W, H, D, mc, job_angle[j], (u8*)carvedN + (i64)k * nvox, (u8*)tmp));
        }
        hipLaunchKernelGGL(k_sub_bitset, dim3(pb3d_stream_blocks(ctx, W * H, 256, 8)), dim3(256), 0, ctx->stream, d_mask_sub, jl, W * H, (u32*)S);
        PB3D_CHECK_LAUNCH();
        if (base90)
            hipLaunchKernelGGL(k_part_multi16<true>, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, (const u32x4*)carvedN, ngroups, jl.n,
                               (const u32*)S, mD, (u32x4*)d_out, ngroups);
        else
            hipLaunchKernelGGL(k_part_multi16<false>, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, (const u32x4*)carvedN, ngroups, jl.n,
                               (const u32*)S, mD, (u32x4*)d_out, ngroups);
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    bool any = false;
    for (int j = 0; j < njobs; ++j) {
        if (job_skip[j]) continue;
        const u8* ms = d_mask_sub + (i64)j * W * H;
        const u8* mc = d_mask_carve + (i64)j * W * H;
        if (wide) {
            hipLaunchKernelGGL(k_part_occ16, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, ms, (u32x4*)occ, ngroups, mD);
            if (vtail < nvox) hipLaunchKernelGGL(k_part_occ, dim3(1), dim3(256), 0, ctx->stream, d_colored, ms, (u8*)occ, nvox, D, vtail);
        } else
            hipLaunchKernelGGL(k_part_occ, dim3(blocks), dim3(256), 0, ctx->stream, d_colored, ms, (u8*)occ, nvox, D);
        PB3D_CHECK_LAUNCH();
        PB3D_TRY(pb3d_process_grid_binary_dev(ctx, (const u8*)occ, W, H, D, mc, job_angle[j], (u8*)carved, (u8*)tmp));
        if (wide) {
            hipLaunchKernelGGL(k_keep_or16, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)carved, ms, (u32x4*)keep, ngroups, mD, any ? 0 : 1);
            if (vtail < nvox) hipLaunchKernelGGL(k_keep_or, dim3(1), dim3(256), 0, ctx->stream, (const u8*)carved, ms, (u8*)keep, nvox, D, any ? 0 : 1, vtail);
        } else
            hipLaunchKernelGGL(k_keep_or, dim3(blocks), dim3(256), 0, ctx->stream, (const u8*)carved, ms, (u8*)keep, nvox, D, any ? 0 : 1);
        PB3D_CHECK_LAUNCH();
        any = true;
    }
    if (!any) {
        if (!base90) PB3D_HIP(hipMemsetAsync(d_out, 0, (size_t)nvox * 3, ctx->stream));
        return PB3D_OK;
    }
    if (base90) {
        if (wide) {
            hipLaunchKernelGGL(k_part_merge16, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, (const u32x4*)keep, (u32x4*)d_out,
                               ngroups);
            if (vtail < nvox) hipLaunchKernelGGL(k_part_merge, dim3(1), dim3(256), 0, ctx->stream, d_colored, (const u8*)keep, d_out, nvox, vtail);
        } else
            hipLaunchKernelGGL(k_part_merge, dim3(blocks), dim3(256), 0, ctx->stream, d_colored, (const u8*)keep, d_out, nvox);
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    if (wide) {
        hipLaunchKernelGGL(k_part_final16, dim3(gblocks), dim3(256), 0, ctx->stream, (const u32x4*)d_colored, (const u32x4*)keep, (u32x4*)d_out,
                           ngroups);
        if (vtail < nvox) hipLaunchKernelGGL(k_part_final, dim3(1), dim3(256), 0, ctx->stream, d_colored, (const u8*)keep, d_out, nvox, vtail);
    } else
        hipLaunchKernelGGL(k_part_final, dim3(blocks), dim3(256), 0, ctx->stream, d_colored, (const u8*)keep, d_out, nvox);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

}  // extern "C"
