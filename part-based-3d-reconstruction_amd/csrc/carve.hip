// K1 mask-carve sweep, occupancy, colour apply, part overlay -- HBM-bound byte kernels (gfx950).
//
// Data layout (the reference's own): grid (W,H,D[,3]) uint8 in C order, so the D*C bytes of
// one (x,y) column are contiguous and column index xy = x*H + y also indexes the (W,H) mask.
// Every kernel here is a coalesced 16-byte-per-lane sweep; none has a contraction (no MFMA).
#include "pb3d_internal.h"
#include <vector>

#include "lane48.h"

namespace {

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
// ------------------------------------------------------------------------------------------------
// _occupancy: 16 voxels (48 bytes in, 16 bytes out) per thread.
// ------------------------------------------------------------------------------------------------

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
    const bool multi = wide && nrest_all >= 1 && nrest_all <= 8 && nvox % 16 == 0 && ctx->tune_per_job != 1;
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
