// The 90-degree forms of global_carve and part_carve on BITS and STREAMS (round 4): keep bits computed per 16-voxel group
// (k_global_carve90s) and the occupancy bits of a Y-plane transposed in registers (k_part90_plane).  DESIGN.md section 3, "Bits and streams".
#include "pb3d_internal.h"

#include "lane48.h"

namespace {

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

}  // namespace

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
    if (C == 3 && ctx->tune_part90_inflight) {               // development A/B: 10 UA + UE
        const int t = ctx->tune_part90_inflight;
        kern = t == 11 ? k_part90_plane<3, 1, 1> : t == 12 ? k_part90_plane<3, 1, 2> : t == 14 ? k_part90_plane<3, 1, 4> : t == 21 ? k_part90_plane<3, 2, 1>
             : t == 22 ? k_part90_plane<3, 2, 2> : t == 42 ? k_part90_plane<3, 4, 2> : t == 44 ? k_part90_plane<3, 4, 4> : kern;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((W + 127) / 128), (unsigned)H), dim3(256), lds, ctx->stream, d_colored, d_A, d_AT, d_vbits, nwv, c0, c2,
                       W, H, D, nwz, njobs, pb3d_make_magic((u32)npieces), d_out);
    PB3D_CHECK_LAUNCH();
    *took = 1;
    return PB3D_OK;
}
