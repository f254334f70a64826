// K2': rotation steps whose coordinates are (numerically almost) integers -- 0 and 90 degrees on
// grids with W + D even -- as LDS-tiled byte permutations, and K4: the fused global_carve.
//
// For such a step every source coordinate cc lies within 2^-18 of an integer, so SciPy's trilinear
// sum has one weight >= 1 - 2^-17 and the rest <= 2^-18; with uint8 data |acc - v_nearest| < 0.01, the
// "+0.5, truncate" store returns v_nearest exactly, and acc > 0 iff it should (proof in DESIGN.md).
// The border decisions (cc < 0, cc > n-1) are still taken on the SAME f64 values SciPy computes --
// cos(90 deg) = 6.1e-17 drops ~1 % of the border cells and that is reproduced, not patterned.
// The kernel is then a tiled transpose inside each Y-plane: the source tile is staged through LDS
// with coalesced dword row loads, outputs are gathered from LDS and stored as packed dwords.
#include "pb3d_internal.h"

namespace {

struct RotParams {
    double m00, m01, m02, off0;
    double m20, m21, m22, off2;
};

__device__ __forceinline__ double coord(double x, double z, double ma, double mb, double mc, double off) {
    double c = __dadd_rn(0.0, __dmul_rn(x, ma));
    c = __dadd_rn(c, __dmul_rn(0.0, mb));
    c = __dadd_rn(c, __dmul_rn(z, mc));
    return __dadd_rn(c, off);
}

// nearest source voxel of output (x,z), or false when SciPy's bounds test rejects the coordinate
__device__ __forceinline__ bool source_of(const RotParams& p, i64 x, i64 z, i64 W, i64 D, int* n0, int* n2) {
    const double cc0 = coord((double)x, (double)z, p.m00, p.m01, p.m02, p.off0);
    const double cc2 = coord((double)x, (double)z, p.m20, p.m21, p.m22, p.off2);
    if (cc0 < 0.0 || cc0 > (double)(W - 1) || cc2 < 0.0 || cc2 > (double)(D - 1)) return false;
    *n0 = (int)rint(cc0);
    *n2 = (int)rint(cc2);
    return true;
}

constexpr int T = 64;          // tile edge (voxels)
constexpr int PITCH = T + 8;   // LDS row pitch in bytes (dword multiple; z start is aligned down to 4)
constexpr int ROWS = T + 1;
constexpr int MAXLD = (ROWS * (PITCH / 4) + 255) / 256;  // staging dwords per thread

// out[x,y,z] = mask_dst[x,y] && valid(x,z) ? (mask_src[n0,y] ? in[n0,y,n2] : 0) : 0
__global__ __launch_bounds__(256) void k_rotate_perm(const u8* __restrict__ in, u8* __restrict__ out,
                                                     const u8* __restrict__ mask_src, const u8* __restrict__ mask_dst,
                                                     RotParams p, i64 W, i64 H, i64 D, int TY) {
    __shared__ __attribute__((aligned(16))) u8 tile[ROWS * PITCH];
    __shared__ int bb[4];  // min n0, max n0, min n2, max n2
    const int tid = threadIdx.x;
    const i64 x0 = (i64)blockIdx.y * T, z0 = (i64)blockIdx.x * T;
    const i64 y_beg = (i64)blockIdx.z * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    __syncthreads();
    // 16 cells per thread: rows xl = tid/16 + 16k (k = 0..3), z = (tid%16)*4 + q (q = 0..3)
    const int zl = (tid & 15) * 4, xl0 = tid >> 4;
    int n0[16], n2[16];
    int mn0 = 0x7fffffff, mx0 = -1, mn2 = 0x7fffffff, mx2 = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const i64 x = x0 + xl0 + 16 * k, z = z0 + zl + q;
            int a = -1, b = -1;
            if (x < W && z < D && source_of(p, x, z, W, D, &a, &b)) {
                mn0 = a < mn0 ? a : mn0; mx0 = a > mx0 ? a : mx0;
                mn2 = b < mn2 ? b : mn2; mx2 = b > mx2 ? b : mx2;
            } else {
                a = -1;
            }
            n0[4 * k + q] = a; n2[4 * k + q] = b;
        }
    if (mx0 >= 0) {
        atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); atomicMin(&bb[2], mn2); atomicMax(&bb[3], mx2);
    }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1], bz0 = bb[2] & ~3, bz1 = bb[3];
    const bool any_valid = bx1 >= 0;
    const int nrows = any_valid ? bx1 - bx0 + 1 : 0;
    const int nd = any_valid ? (bz1 - bz0) / 4 + 1 : 0;  // dwords per staged row
    // a signed-permutation map sends a T x T tile to a T x T tile: these bounds hold by construction
    // (checked on the host); the guard only keeps a bad launch from writing outside LDS.
    const bool fits = nrows <= ROWS && nd * 4 <= PITCH;
    u32 lo[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) lo[c] = (n0[c] >= 0 && fits) ? (u32)((n0[c] - bx0) * PITCH + (n2[c] - bz0)) : 0xffffffffu;
    // staging assignment: dword i of the tile -> (row, dword-in-row)
    int sr[MAXLD], sc[MAXLD];
#pragma unroll
    for (int j = 0; j < MAXLD; ++j) {
        const int i = tid + 256 * j;
        if (fits && nd > 0 && i < nrows * nd) { sr[j] = i / nd; sc[j] = i - sr[j] * nd; }
        else sr[j] = -1;
    }
    for (i64 y = y_beg; y < y_end; ++y) {
#pragma unroll
        for (int j = 0; j < MAXLD; ++j) {
            if (sr[j] < 0) continue;
            const i64 xs = bx0 + sr[j];
            u32 v = 0;
            if (!mask_src || mask_src[xs * H + y]) v = *(const u32*)(in + (xs * H + y) * D + bz0 + 4 * sc[j]);
            *(u32*)(tile + sr[j] * PITCH + 4 * sc[j]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const i64 x = x0 + xl0 + 16 * k;
            if (x >= W || z0 + zl >= D) continue;
            u32 r = 0;
            if (!mask_dst || mask_dst[x * H + y]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 o = lo[4 * k + q];
                    if (o != 0xffffffffu) r |= (u32)tile[o] << (8 * q);
                }
            }
            *(u32*)(out + (x * H + y) * D + z0 + zl) = r;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Validity table of a permutation-like step: bit z of row x is SciPy's bounds test on the f64
// coordinates of output voxel (x, z).  Computed once per step (W*D cells) so that the sweeping
// kernels below are pure integer work.  Row stride `nw` words; bits for z >= D are zero.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rot_valid(RotParams p, i64 W, i64 D, int nw, u32* __restrict__ bits) {
    const i64 x = blockIdx.y;
    const i64 z = (i64)blockIdx.x * 256 + threadIdx.x;
    int a, b;
    const bool v = x < W && z < D && source_of(p, x, z, W, D, &a, &b);
    const u64 bal = __ballot(v);
    const int lane = threadIdx.x & 63;
    if ((z >> 5) < nw) {
        if (lane == 0) bits[x * nw + (z >> 5)] = (u32)bal;
        if (lane == 32) bits[x * nw + (z >> 5)] = (u32)(bal >> 32);
    }
}

// Per x-row: is its set of valid z an interval?  lo | hi << 16 for [lo, hi) (empty: 0), 0xffffffff otherwise (a kernel then reads
// the bits).  The bounds tests are linear in z for a fixed x, so every row of a 90-degree step is one; the check costs nothing.
__global__ __launch_bounds__(256) void k_rot_ivals(const u32* __restrict__ bits, i64 W, i64 D, int nw, u32* __restrict__ ivals) {
    const i64 x = (i64)blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    int first = -1, last = -1, cnt = 0;
    for (int w = 0; w < nw; ++w) {
        const u32 v = bits[x * nw + w];
        if (!v) continue;
        if (first < 0) first = 32 * w + __ffs((int)v) - 1;
        last = 32 * w + 31 - __clz((int)v);
        cnt += __popc(v);
    }
    ivals[x] = cnt == 0 ? 0u : (cnt == last - first + 1 && last < 65535) ? (u32)first | ((u32)(last + 1) << 16) : 0xffffffffu;
}

// Workgroup -> (z tile, x tile, plane chunk).  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels one), each with
// its own L2.  On grids whose rows are not multiples of 128 bytes every tile row straddles two lines on both sides: the other half
// of a written line belongs to the z-neighbour tile (or the next plane's first tile), the other half of a read line to the
// x-neighbour.  order 0 keeps ALL tiles of one plane chunk on one XCD, z tiles adjacent in dispatch order, so both halves meet in
// that XCD's L2 while the line is still there (partial-line writes merge, shared source lines are fetched once); order 1 is the
// plain x-fastest linear order (neighbours land on different XCDs).
struct TileMap { int nzt, nxt, nyc, order; };
__device__ __forceinline__ bool tile_of_block(const TileMap m, i64* zt, i64* xt, i64* yc) {
    const i64 b = blockIdx.x;
    if (m.order == 1) { *zt = b % m.nzt; *xt = (b / m.nzt) % m.nxt; *yc = b / ((i64)m.nzt * m.nxt); return *yc < m.nyc; }
    const i64 s = b >> 3;
    *zt = s % m.nzt; *xt = (s / m.nzt) % m.nxt; *yc = (s / ((i64)m.nzt * m.nxt)) * 8 + (b & 7);
    return *yc < m.nyc;
}
// planes per workgroup such that the plane chunks spread evenly over the 8 XCDs (a multiple of 8 chunks where possible) and the
// launch has `fill` workgroups per CU -- two rounds of the 4 that are resident.  Measured with tools/tybench.py (variants interleaved
// on one box, process_voxel_grid(occ, 90), ms): fill 4 / 6 / 8 / 12 = 0.460 / 0.462 / 0.460 / 0.473 at 1024^3, 0.083 / 0.093 / 0.0735 /
// 0.082 at 512^3, 0.122 / 0.121 / 0.101 / 0.104 at 437x512x437, 0.093 / 0.082 / 0.080 / 0.072 at 355x512x355; a model of whole
// "rounds" of workgroups did not predict these (exactly one round is never the best: nothing is left to fill the slots of the
// workgroups that finish early).
static inline int planes_per_chunk(i64 H, i64 tiles, int cus, int want_ty, int fill = 0) {
    if (fill <= 0) fill = 8;
    i64 m = (H + 8 * want_ty - 1) / (8 * want_ty);                 // chunks = 8 m
    if (m < 1) m = 1;
    while (tiles * 8 * m < (i64)cus * fill && 8 * (m + 1) <= H) ++m;
    i64 ty = (H + 8 * m - 1) / (8 * m);
    return (int)(ty < 1 ? 1 : ty);
}
static inline unsigned tilemap_blocks(const TileMap& m) {
    return m.order == 1 ? (unsigned)((i64)m.nzt * m.nxt * m.nyc) : (unsigned)(8ll * m.nzt * m.nxt * ((m.nyc + 7) / 8));
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));     // 16-byte access at any byte alignment (rows of odd-sized grids)
typedef u32 u32_u __attribute__((aligned(1)));

// (K4, global_carve(binary, rgb, 90): the per-row piece kernels k_global_carve90v / 90f of rounds 1-3 are gone -- the stream kernel
// k_global_carve90s in csrc/bits90.hip is faster on every shape, profiles/r04_global_carve90_stream_vs_piece_kernels.jsonl.)

// ------------------------------------------------------------------------------------------------
// K2' fast form for the 90-degree map n0 = c0 - z, n2 = x + c2.  Any W, H, D and any pointer alignment: gfx950 serves
// 16-byte global accesses at arbitrary byte addresses at full speed (tools/kbench4.hip), so rows of odd-sized grids
// (355, 437, 123 ... voxels, the reference's real shapes) use the same 16-byte pieces; only the ragged piece at a row's
// end is moved byte-wise.
// 128 x 128 byte tiles, 16-byte global loads and stores (full 128-byte lines on both sides), the
// transpose done in registers: the source tile sits row-major in LDS (16-byte blocks XOR-swizzled by
// the row group so that the column reads are bank-conflict free), each thread reads a 16-row x
// 4-byte block with ds_read_b32 and transposes it with v_perm_b32 into four 16-byte output runs.
// Global loads of plane y+1 are issued before plane y is computed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

#ifndef ROT90_WAVES
#define ROT90_WAVES 4
#endif
// DEPTH planes of global loads are in flight per workgroup (register ring, statically indexed by the unrolled slot loop);
// the LDS tile is double-buffered so a plane costs ONE barrier.  Measured at 1024^3: DEPTH 1 with 4 workgroups/CU 0.51 ms;
// DEPTH 2 (3 workgroups/CU) 0.55; DEPTH 3-4 (2/CU) 0.63 -- resident waves matter more than bytes in flight.  The memory
// system's own bound for this traffic (tools/kbench3.hip: 16 KiB tiles of 128-byte rows 1 MiB apart on both sides, no
// transpose) is 0.455 ms, a linear copy with the same workgroup shape 0.41 ms.
// RAGGED = false: D % 16 == 0 and c2 % 16 == 0, every piece is whole and the byte-wise paths are compiled out (they cost
// 25 % at 1024^3 when merely present)
// (Round 4: the colour-writing and plane-shifted forms of this kernel are gone -- global_carve chains write their colours from the
// bit-sliced volume, csrc/sliced.hip, and rows that are not whole lines take the flat kernels below.)
template <int DEPTH, bool RAGGED>
__global__ __launch_bounds__(256, ROT90_WAVES) void k_rot90(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_src,
                                               const u8* __restrict__ mask_dst, const u32* __restrict__ vbits, int nw, int c0, int c2,
                                               i64 W, i64 H, i64 D, int TY, TileMap tm) {
    __shared__ __attribute__((aligned(16))) u8 tiles[2][128 * 128];
    const int tid = threadIdx.x;
    i64 zt, xt, yc;
    if (!tile_of_block(tm, &zt, &xt, &yc)) return;                          // whole workgroup, before any barrier
    const i64 x0 = xt * 128, z0 = zt * 128;
    const i64 y_beg = yc * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    // staging role: local source row lr = (tid >> 3) + 32 j, 16-byte block cb = tid & 7
    const int cb = tid & 7;
    const i64 rbase0 = (i64)c0 - (z0 + 127);            // source row of local row 0
    const i64 scol = x0 + c2 + 16 * cb;                 // source column of this thread's block
    const int cmode = (scol >= 0 && scol + 15 < D) ? 2 : ((scol + 15 >= 0 && scol < D) ? 1 : 0);   // whole / ragged / outside
    // output role: z-run zg = tid & 7 (16 z), x-group xg = tid >> 3 (4 x)
    const int zg = tid & 7, xg = tid >> 3;
    const int g = 7 - zg;                               // row group holding this thread's 16 source rows
    const u32 rd_off = (u32)(16 * g * 128 + 16 * ((xg >> 2) ^ g) + 4 * (xg & 3));
    const i64 zo = z0 + 16 * zg;                          // first z of this thread's run
    // 16 validity bits of row x for z = zlo .. zlo + 15 (zero outside [0, D): the table is zero there, negative z are shifted out)
    auto vwin = [&](i64 x, i64 zlo) -> u32 {
        if (x >= W || zlo <= -16 || zlo >= D) return 0u;
        const i64 zs = zlo < 0 ? 0 : zlo;
        const u32* vr = vbits + x * nw + (zs >> 5);
        u32 v = (u32)((((u64)vr[1] << 32) | (u64)vr[0]) >> (zs & 31)) & 0xffffu;
        if (zlo < 0) v = (v << (int)(-zlo)) & 0xffffu;
        return v;
    };
    u32 vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vb[i] = vwin(x0 + 4 * xg + i, zo);
    // Everything a plane needs from global memory is issued together, DEPTH planes ahead: the 16-byte source pieces, the
    // source-row mask bytes (applied when the data lands, so the two loads are not dependent) and the
    // destination-row mask bytes.
    u32x4 stg[DEPTH][4];
    u32 msk[DEPTH];    // bit j: source-row mask of piece j ; bits 4..7: destination-row mask of row i
    auto load_plane = [&](u32x4 (&sg)[4], u32& mkout, i64 y) {
        u32 mk = 0;
        const i64 rbase = rbase0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 n0 = rbase + (tid >> 3) + 32 * j;
            sg[j] = (u32x4)(0u);
            if (y < y_end && (RAGGED ? cmode != 0 : cmode == 2) && n0 >= 0 && n0 < W) {
                const u8* sp = in + (n0 * H + y) * D + scol;
                // a ragged piece may be read whole as long as it stays inside the volume: the bytes beyond the row belong to
                // the neighbouring row and are dropped by the validity bits (their source column is outside [0, D))
                if (!RAGGED || cmode == 2 || (sp >= in && sp + 16 <= in + W * H * D)) sg[j] = __builtin_nontemporal_load((const u32x4_u*)sp);
                else {
                    u32 t4[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; ++b)
                        if (scol + b >= 0 && scol + b < D) t4[b >> 2] |= (u32)sp[b] << (8 * (b & 3));
                    sg[j].x = t4[0]; sg[j].y = t4[1]; sg[j].z = t4[2]; sg[j].w = t4[3];
                }
                mk |= (u32)((mask_src ? mask_src[n0 * H + y] : (u8)1) != 0) << j;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            if (y < y_end && x < W && vb[i]) mk |= (u32)((mask_dst ? mask_dst[x * H + y] : (u8)1) != 0) << (4 + i);
        }
        mkout = mk;
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_plane(stg[s], msk[s], y_beg + s);
    int buf = 0;
    for (i64 y0p = y_beg; y0p < y_end; y0p += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const i64 y = y0p + s;
            if (y >= y_end) break;                       // uniform
            u8* tile = tiles[buf];
            buf ^= 1;
            const u32 mkc = msk[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr = (tid >> 3) + 32 * j;
                *(u32x4*)(tile + lr * 128 + 16 * (cb ^ ((lr >> 4) & 7))) = ((mkc >> j) & 1u) ? stg[s][j] : (u32x4)(0u);
            }
            __syncthreads();      // the only barrier of the plane: the other tile buffer was last read before the previous one
            load_plane(stg[s], msk[s], y + DEPTH);
            u32 d[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) d[rr] = *(const u32*)(tile + rd_off + rr * 128);
            u32 o[4][4];  // o[i][w]: output x = 4 xg + i, bytes q = 4w .. 4w+3 ; byte q <- d[15 - q].byte[i]
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 A = d[15 - 4 * w], B = d[14 - 4 * w], Cc = d[13 - 4 * w], E = d[12 - 4 * w];
                const u32 t0 = perm(B, A, 0x05010400u), t1 = perm(B, A, 0x07030602u);
                const u32 u0 = perm(E, Cc, 0x05010400u), u1 = perm(E, Cc, 0x07030602u);
                o[0][w] = perm(u0, t0, 0x05040100u);
                o[1][w] = perm(u0, t0, 0x07060302u);
                o[2][w] = perm(u1, t1, 0x05040100u);
                o[3][w] = perm(u1, t1, 0x07060302u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i64 x = x0 + 4 * xg + i;
                if (x >= W || zo >= D) continue;
                const u32 vbi = vb[i];
                u32x4 r = (u32x4)(0u);
                if ((mkc >> (4 + i)) & 1u) {
                    r.x = o[i][0]; r.y = o[i][1]; r.z = o[i][2]; r.w = o[i][3];
                    if (vbi != 0xffffu) {  // border cells rejected by the f64 bounds test (rare)
                        u32 mw[4];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const u32 b4 = (vbi >> (4 * w)) & 0xfu;
                            mw[w] = ((b4 & 1u) ? 0x000000ffu : 0u) | ((b4 & 2u) ? 0x0000ff00u : 0u) | ((b4 & 4u) ? 0x00ff0000u : 0u) |
                                    ((b4 & 8u) ? 0xff000000u : 0u);
                        }
                        r.x &= mw[0]; r.y &= mw[1]; r.z &= mw[2]; r.w &= mw[3];
                    }
                }
                u8* op = out + (x * H + y) * D + zo;
                if (!RAGGED || zo + 15 < D) __builtin_nontemporal_store(r, (u32x4_u*)op);
                else {
                    const u32 t4[4] = {r.x, r.y, r.z, r.w};
                    const int k = (int)(D - zo);                       // 1..15 bytes: whole dwords, then bytes
                    for (int j = 0; j < (k >> 2); ++j) *(u32_u*)(op + 4 * j) = t4[j];
                    for (int b = k & ~3; b < k; ++b) op[b] = (u8)(t4[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same step on 256 x 256 tiles (D % 16 == 0, c2 % 16 == 0): rows of 256 bytes -- TWO whole lines -- on BOTH sides from one
// instruction stream, the access pattern tools/kbench3.hip measured at 0.417 ms against 0.455 ms for 128-byte rows at 1024^3.
// 1024 threads with k_rot90's roles (staging: 16 lanes per source row, 64 rows per pass, 4 passes; output: 16 z-runs x 64 x-groups
// of 4 rows), ONE 64 KB tile in LDS (two barriers per plane; the next plane's loads are in flight in registers meanwhile), the
// 16-byte blocks of a row XOR-swizzled by the row group so that the column reads hit 64 banks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rot90w(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_src,
                                                 const u8* __restrict__ mask_dst, const u32* __restrict__ vbits, int nw, int c0, int c2,
                                                 i64 W, i64 H, i64 D, int TY, TileMap tm, int blk_on) {
    extern __shared__ __attribute__((aligned(16))) u8 wtile[];          // 256 rows x 256 bytes, then the workgroup's mask flags
    const int tid = threadIdx.x;
    i64 zt, xt, yc;
    if (!tile_of_block(tm, &zt, &xt, &yc)) return;
    const i64 x0 = xt * 256, z0 = zt * 256;
    const i64 y_beg = yc * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    const int cb = tid & 15;                             // staging: 16-byte block of the source row
    const i64 rbase0 = (i64)c0 - (z0 + 255);            // source row of local row 0
    const i64 scol = x0 + c2 + 16 * cb;
    const bool col_ok = scol >= 0 && scol + 15 < D;
    const int zg = tid & 15, xg = tid >> 4;              // output: z-run (16 z), x-group (4 x)
    const int g = 15 - zg;                               // row group holding this thread's 16 source rows
    const u32 rd_off = (u32)(16 * g * 256 + 16 * ((xg >> 2) ^ g) + 4 * (xg & 3));
    const i64 zo = z0 + 16 * zg;
    u32 vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i64 x = x0 + 4 * xg + i;
        u32 v = 0;
        if (x < W && zo < D) { const u32* vr = vbits + x * nw + (zo >> 5); v = (u32)((((u64)vr[1] << 32) | (u64)vr[0]) >> (zo & 31)) & 0xffffu; }
        vb[i] = v;
    }
    // The workgroup's mask flags (blk_on): its 256 source rows and 256 output rows are the same for all of its planes, and the mask
    // bytes of one row's planes are neighbours in memory -- fetched once, eight planes per load, kept as 0 / 1 flags behind the tile:
    // flag of plane y_beg + p at blk[512 p + r], r < 256 a source row, 256 + r an output row.  (Per plane, every thread fetched the
    // bytes of its four source rows and four output rows: 8 more memory instructions per thread, each byte a 64-byte line of its own.)
    u8* blk = wtile + 256 * 256;
    if (blk_on) {
        if (tid < 512) {
            const bool src = tid < 256;
            const i64 row = src ? rbase0 + tid : x0 + (tid - 256);
            const u8* mp = src ? mask_src : mask_dst;
            const bool rok = row >= 0 && row < W;
            for (i64 y = y_beg; y < y_end; y += 8) {
                u32 lo = 0, hi = 0;
                if (rok) {
                    if (!mp) { lo = 0x01010101u; hi = 0x01010101u; }
                    else if (y + 8 <= H) { lo = *(const u32_u*)(mp + row * H + y); hi = *(const u32_u*)(mp + row * H + y + 4); }
                    else
                        for (int b = 0; b < 8 && y + b < H; ++b) { const u32 v = mp[row * H + y + b]; if (b < 4) lo |= v << (8 * b); else hi |= v << (8 * (b - 4)); }
                }
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (y + b < y_end) blk[512 * (y + b - y_beg) + tid] = (u8)((((b < 4 ? lo : hi) >> (8 * (b & 3))) & 0xffu) != 0);
            }
        }
        __syncthreads();
    }
    u32x4 stg[4];
    u32 ms_raw, md_raw;       // the mask bytes as loaded (byte j: source row j, byte i: output row i): a prefetch only ISSUES loads --
                              // looking at a mask byte here would wait for it, and with it (the counter is in order) for the plane's data
    auto load_plane = [&](i64 y) {
        u32 a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 n0 = rbase0 + (tid >> 4) + 64 * j;
            stg[j] = (u32x4)(0u);
            if (y < y_end && col_ok && n0 >= 0 && n0 < W) {
                stg[j] = __builtin_nontemporal_load((const u32x4*)(in + (n0 * H + y) * D + scol));
                if (!blk_on) a |= (mask_src ? (u32)mask_src[n0 * H + y] : 1u) << (8 * j);
            }
        }
        if (!blk_on) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i64 x = x0 + 4 * xg + i;
                if (y < y_end && x < W && vb[i]) b |= (mask_dst ? (u32)mask_dst[x * H + y] : 1u) << (8 * i);
            }
        }
        ms_raw = a; md_raw = b;
    };
    load_plane(y_beg);
    for (i64 y = y_beg; y < y_end; ++y) {
        u32 mkc = 0;
        if (blk_on) {
            const u8* bp = blk + 512 * (y - y_beg);
#pragma unroll
            for (int j = 0; j < 4; ++j) mkc |= (u32)bp[(tid >> 4) + 64 * j] << j;
            const u32 md = *(const u32*)(bp + 256 + 4 * xg);
#pragma unroll
            for (int i = 0; i < 4; ++i) mkc |= (vb[i] ? ((md >> (8 * i)) & 1u) : 0u) << (4 + i);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) mkc |= (u32)(((ms_raw >> (8 * j)) & 0xffu) != 0) << j | (u32)(((md_raw >> (8 * j)) & 0xffu) != 0) << (4 + j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 4) + 64 * j;
            *(u32x4*)(wtile + lr * 256 + 16 * (cb ^ ((lr >> 4) & 15))) = ((mkc >> j) & 1u) ? stg[j] : (u32x4)(0u);
        }
        __syncthreads();
        load_plane(y + 1);
        u32 d[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) d[rr] = *(const u32*)(wtile + rd_off + rr * 256);
        __syncthreads();                                  // the tile is free again: the next plane's data may be written
        u32 o[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32 A = d[15 - 4 * w], B = d[14 - 4 * w], Cc = d[13 - 4 * w], E = d[12 - 4 * w];
            const u32 t0 = perm(B, A, 0x05010400u), t1 = perm(B, A, 0x07030602u);
            const u32 u0 = perm(E, Cc, 0x05010400u), u1 = perm(E, Cc, 0x07030602u);
            o[0][w] = perm(u0, t0, 0x05040100u);
            o[1][w] = perm(u0, t0, 0x07060302u);
            o[2][w] = perm(u1, t1, 0x05040100u);
            o[3][w] = perm(u1, t1, 0x07060302u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            if (x >= W || zo >= D) continue;
            const u32 vbi = vb[i];
            u32x4 r = (u32x4)(0u);
            if ((mkc >> (4 + i)) & 1u) {
                r.x = o[i][0]; r.y = o[i][1]; r.z = o[i][2]; r.w = o[i][3];
                if (vbi != 0xffffu) {
                    u32 mw[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 b4 = (vbi >> (4 * w)) & 0xfu;
                        mw[w] = ((b4 & 1u) ? 0x000000ffu : 0u) | ((b4 & 2u) ? 0x0000ff00u : 0u) | ((b4 & 4u) ? 0x00ff0000u : 0u) | ((b4 & 8u) ? 0xff000000u : 0u);
                    }
                    r.x &= mw[0]; r.y &= mw[1]; r.z &= mw[2]; r.w &= mw[3];
                }
            }
            __builtin_nontemporal_store(r, (u32x4*)(out + (x * H + y) * D + zo));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2' for rows that are not whole lines, FLAT form (grids with H * D % 128 == 0: every real shape of the reference whose longer mask
// side is the height, e.g. Charminar 355 x 512 x 355).  For a fixed x the rows (x, y, :) of all planes follow each other in memory:
// an output x-row is ONE contiguous stream of H * D bytes, f = y * D + z.  The kernel tiles that stream, not the planes: a tile is
// 128 x-rows x one aligned 128-byte SEGMENT s of the stream (f = 128 s .. 128 s + 127), whatever rows of whichever one or two planes
// it holds.  Every store is then a whole aligned line, nothing is clipped at row ends, and a launch has H * D / 128 segment steps per
// x-tile instead of ceil((D + 127) / 128) * H tile-planes (355 x 512 x 355: 1420 against 2048).  Byte j of segment s is voxel
// (y, z) = divmod(128 s + j, D) and comes from source row n0 = c0 - z of plane y: 128 row reads of 128 bytes along x, as in k_rot90;
// LDS layout, transpose and thread roles are k_rot90's (local source row lr holds byte j = 127 - lr).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, ROT90_WAVES) void k_rot90_flat(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_src,
                                                                 const u8* __restrict__ mask_dst, const u32* __restrict__ vbits, int nw, int c0, int c2,
                                                                 i64 W, i64 H, i64 D, int TS, TileMap tm, pb3d_magic mD, i64 nseg) {
    __shared__ __attribute__((aligned(16))) u8 tiles[2][128 * 128];
    const int tid = threadIdx.x;
    i64 zt, xt, sc;
    if (!tile_of_block(tm, &zt, &xt, &sc)) return;                          // whole workgroup, before any barrier
    const i64 x0 = xt * 128;
    const i64 s_beg = sc * TS;
    const i64 s_end = s_beg + TS < nseg ? s_beg + TS : nseg;
    const i64 HD = H * D;
    const int cb = tid & 7;
    const i64 scol = x0 + c2 + 16 * cb;                 // source column of this thread's block
    const int cmode = (scol >= 0 && scol + 15 < D) ? 2 : ((scol + 15 >= 0 && scol < D) ? 1 : 0);   // whole / ragged / outside
    const int zg = tid & 7, xg = tid >> 3;
    const int g = 7 - zg;
    const u32 rd_off = (u32)(16 * g * 128 + 16 * ((xg >> 2) ^ g) + 4 * (xg & 3));
    // 16 validity bits of row x for z = zlo .. zlo + 15 (zero outside [0, D))
    auto vwin = [&](i64 x, i64 zlo) -> u32 {
        if (zlo <= -16 || zlo >= D) return 0u;
        const i64 zs = zlo < 0 ? 0 : zlo;
        const u32* vr = vbits + x * nw + (zs >> 5);
        u32 v = (u32)((((u64)vr[1] << 32) | (u64)vr[0]) >> (zs & 31)) & 0xffffu;
        if (zlo < 0) v = (v << (int)(-zlo)) & 0xffffu;
        return v;
    };
    u32x4 stg[4];
    u32 msk;           // bit j: source-row mask of piece j
    u32 keep[4];       // per output row: the 16 keep bits of this thread's piece (validity AND destination-row mask)
    auto load_seg = [&](i64 s) {
        u32 mk = 0;
        const bool live = s < s_end;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 3) + 32 * j;
            const u32 f = (u32)(128 * s) + (u32)(127 - lr);
            const u32 y = pb3d_div(f, mD), z = f - y * mD.d;
            const i64 n0 = (i64)c0 - (i64)z;
            stg[j] = (u32x4)(0u);
            if (live && (i64)f < HD && cmode != 0 && n0 >= 0 && n0 < W) {          // (f >= H * D: the ragged last segment of a stream that is not whole lines)
                const u8* sp = in + (n0 * H + (i64)y) * D + scol;
                // a ragged piece may be read whole as long as it stays inside the volume: the bytes beyond the row belong to
                // the neighbouring row and are dropped by the validity bits (their source column is outside [0, D))
                if (cmode == 2 || (sp >= in && sp + 16 <= in + W * HD)) stg[j] = __builtin_nontemporal_load((const u32x4_u*)sp);
                else {
                    u32 t4[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; ++b)
                        if (scol + b >= 0 && scol + b < D) t4[b >> 2] |= (u32)sp[b] << (8 * (b & 3));
                    stg[j].x = t4[0]; stg[j].y = t4[1]; stg[j].z = t4[2]; stg[j].w = t4[3];
                }
                mk |= (u32)((mask_src ? mask_src[n0 * H + (i64)y] : (u8)1) != 0) << j;
            }
        }
        msk = mk;
        // this thread's output piece: bytes j = 16 zg .. 16 zg + 15 of the segment = voxels (y, z .. ) and, past a row end, (y + 1, 0 ..)
        const u32 f = (u32)(128 * s) + (u32)(16 * zg);
        const bool pin = live && (i64)f < HD;                               // this thread's piece lies inside the stream (H * D % 16 == 0: whole or absent)
        const u32 y = pin ? pb3d_div(f, mD) : 0u, z = pin ? f - y * mD.d : 0u;
        const int nA = (i64)z + 16 <= D ? 16 : (int)(D - (i64)z);          // bytes of the piece in plane y (D >= 16)
        const u32 lowA = (1u << nA) - 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            u32 k = 0;
            if (pin && x < W) {
                u32 bits = vwin(x, (i64)z);
                if (nA < 16) bits |= vwin(x, (i64)z - D);
                if (bits) {
                    const u32 mA = (mask_dst ? mask_dst[x * H + (i64)y] : (u8)1) != 0 ? lowA : 0u;
                    const u32 mB = (nA < 16 && (mask_dst ? mask_dst[x * H + (i64)y + 1] : (u8)1) != 0) ? (0xffffu & ~lowA) : 0u;
                    k = bits & (mA | mB);
                }
            }
            keep[i] = k;
        }
    };
    load_seg(s_beg);
    int buf = 0;
    for (i64 s = s_beg; s < s_end; ++s) {
        u8* tile = tiles[buf];
        buf ^= 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 3) + 32 * j;
            *(u32x4*)(tile + lr * 128 + 16 * (cb ^ ((lr >> 4) & 7))) = ((msk >> j) & 1u) ? stg[j] : (u32x4)(0u);
        }
        __syncthreads();      // the only barrier of the segment: the other tile buffer was last read before the previous one
        u32 kcur[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kcur[i] = keep[i];
        load_seg(s + 1);
        u32 d[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) d[rr] = *(const u32*)(tile + rd_off + rr * 128);
        u32 o[4][4];  // o[i][w]: output x = 4 xg + i, bytes q = 4w .. 4w+3 ; byte q <- d[15 - q].byte[i]
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32 A = d[15 - 4 * w], B = d[14 - 4 * w], Cc = d[13 - 4 * w], E = d[12 - 4 * w];
            const u32 t0 = perm(B, A, 0x05010400u), t1 = perm(B, A, 0x07030602u);
            const u32 u0 = perm(E, Cc, 0x05010400u), u1 = perm(E, Cc, 0x07030602u);
            o[0][w] = perm(u0, t0, 0x05040100u);
            o[1][w] = perm(u0, t0, 0x07060302u);
            o[2][w] = perm(u1, t1, 0x05040100u);
            o[3][w] = perm(u1, t1, 0x07060302u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            if (x >= W) continue;
            const u32 kb = kcur[i];
            u32x4 r = (u32x4)(0u);
            if (kb) {
                r.x = o[i][0]; r.y = o[i][1]; r.z = o[i][2]; r.w = o[i][3];
                if (kb != 0xffffu) {
                    u32 mw[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 b4 = (kb >> (4 * w)) & 0xfu;
                        mw[w] = ((b4 & 1u) ? 0x000000ffu : 0u) | ((b4 & 2u) ? 0x0000ff00u : 0u) | ((b4 & 4u) ? 0x00ff0000u : 0u) |
                                ((b4 & 8u) ? 0xff000000u : 0u);
                    }
                    r.x &= mw[0]; r.y &= mw[1]; r.z &= mw[2]; r.w &= mw[3];
                }
            }
            if (128 * s + 16 * zg < HD)                                                       // (absent only in a ragged last segment)
                __builtin_nontemporal_store(r, (u32x4*)(out + x * HD + 128 * s + 16 * zg));     // a whole aligned piece (of a whole aligned line when H * D % 128 == 0)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2' with k_rot90w's workgroup (round 3): k_rot90_flat's tiling of the x-rows' (y, z) streams, 1024 threads, one 64 KB tile in LDS.
// A tile is XW = 16 npc x-rows (W split evenly over ceil(W / 256) tiles: 355 -> 2 x 192, so a ragged W does not leave one tile
// mostly empty) x one 256-byte segment of the stream; local source row lr holds byte j = 255 - lr of the segment.
// Nothing in the prefetch may WAIT: tools/kbench6.hip (this data movement without masks) runs 355 x 512 x 355 in 34 us where the
// first form of this kernel took 50 -- its prefetch looked a validity window up and then, depending on it, the destination mask,
// two dependent round trips inside what should only issue loads (the wait for the first also waits for the segment's data: the
// counter is in order).  Here the validity of an x-row is an INTERVAL of z (k_rot_ivals, cached with the bit table; the rows are
// fixed for the workgroup's life, so it sits in registers), and the prefetch keeps the mask bytes raw: they are looked at where
// the segment is consumed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 bits_between(int lo, int hi) {       // bits q in [0, 16) with lo <= q < hi
    lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
    return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
}
__global__ __launch_bounds__(1024) void k_rot90wf(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_src,
                                                  const u8* __restrict__ mask_dst, const u32* __restrict__ vbits, const u32* __restrict__ ivals,
                                                  int nw, int c0, int c2, i64 W, i64 H, i64 D, int TS, TileMap tm, pb3d_magic mD, i64 nseg, int npc, int blk_on) {
    extern __shared__ __attribute__((aligned(16))) u8 wtile[];          // 256 local rows x 256 bytes, then the workgroup's mask block
    __shared__ __attribute__((aligned(16))) u8 msh[1024];               // mask flags (0 / 1) of the segment: [0, 256) source rows, 256 + 256 p + r: output row r, plane ya + p
    const int tid = threadIdx.x;
    i64 zt, xt, sc;
    if (!tile_of_block(tm, &zt, &xt, &sc)) return;                          // whole workgroup, before any barrier
    const int XW = 16 * npc;
    const i64 x0 = xt * XW;
    const i64 s_beg = sc * TS;
    const i64 s_end = s_beg + TS < nseg ? s_beg + TS : nseg;
    const i64 HD = H * D;
    const int cb = tid & 15;
    const i64 scol = x0 + c2 + 16 * cb;                 // source column of this thread's block
    const int cmode = 16 * cb >= XW ? 0 : (scol >= 0 && scol + 15 < D) ? 2 : ((scol + 15 >= 0 && scol < D) ? 1 : 0);   // whole / ragged / outside
    const int zg = tid & 15, xg = tid >> 4;
    const int g = 15 - zg;
    const u32 rd_off = (u32)(16 * g * 256 + 16 * ((xg >> 2) ^ g) + 4 * (xg & 3));
    const bool xrow_ok = 4 * xg < XW;                   // (XW is a multiple of 16: a group of 4 rows is inside the tile or outside)
    const int mgrp = tid >> 8, mrow = tid & 255;        // mask role: group 0 = source row mrow, group p + 1 = output row mrow at plane ya + p
    u32 iv[4];                                          // valid z of this thread's output rows: lo | hi << 16; 0xffffffff: not an interval, use the bits
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i64 x = x0 + 4 * xg + i;
        iv[i] = (xrow_ok && x < W) ? ivals[x] : 0u;
    }
    auto vwin = [&](i64 x, i64 zlo) -> u32 {
        if (zlo <= -16 || zlo >= D) return 0u;
        const i64 zs = zlo < 0 ? 0 : zlo;
        const u32* vr = vbits + x * nw + (zs >> 5);
        u32 v = (u32)((((u64)vr[1] << 32) | (u64)vr[0]) >> (zs & 31)) & 0xffffu;
        if (zlo < 0) v = (v << (int)(-zlo)) & 0xffffu;
        return v;
    };
    // The workgroup's MASK BLOCK: its segments lie in the planes Y0 .. Y1 (a handful: TS * 256 / D), and of the two mask images they
    // touch only the pixels (any source row, those planes) and (the tile's XW output rows, those planes).  Those bytes are fetched
    // ONCE, eight planes of a row per load (they are neighbours in memory), and stay in LDS behind the tile: row r of window w at
    // blk + (w * R + r) * 8, rows [0, W) = mask_src's, W + i = mask_dst's row x0 + i.  Fetched per segment instead, every byte was
    // one 64-byte line through the L1 that the segment's own 64 KB had just emptied -- as much line traffic again as the data
    // (355 x 512 x 355: 44.7 us, 37.7 without masks).  blk_on = 0 (the block would not fit): the bytes come from memory per segment.
    u8* blk = wtile + 256 * 256;
    const u32 Y0 = pb3d_div((u32)(256 * s_beg), mD);
    const i64 flast = (256 * s_end < HD ? 256 * s_end : HD) - 1;
    const u32 Y1 = pb3d_div((u32)flast, mD);
    const int R = (int)W + XW;
    auto fill_block = [&]() {
        const int nwin = (int)(Y1 - Y0) / 8 + 1;
        for (int r = tid; r < R; r += 1024) {
            const bool src = r < W;
            const i64 row = src ? r : x0 + (r - W);
            const u8* mp = src ? mask_src : mask_dst;
            for (int w = 0; w < nwin; ++w) {
                const i64 y = (i64)Y0 + 8 * w;
                u32 lo = 0, hi = 0;
                if (row < W) {
                    if (!mp) { lo = 0x01010101u; hi = 0x01010101u; }
                    else if (y + 8 <= H) { lo = *(const u32_u*)(mp + row * H + y); hi = *(const u32_u*)(mp + row * H + y + 4); }
                    else
                        for (int b = 0; b < 8 && y + b < H; ++b) { const u32 v = mp[row * H + y + b]; if (b < 4) lo |= v << (8 * b); else hi |= v << (8 * (b - 4)); }
                }
                *(u32*)(blk + ((i64)w * R + r) * 8) = lo;
                *(u32*)(blk + ((i64)w * R + r) * 8 + 4) = hi;
            }
        }
    };
    // The loop is software-pipelined by hand: a workgroup is alone on its CU, so whatever stands between "the segment's data has
    // arrived" and "the next segment's loads are issued" is paid in full.  The ADDRESSES of segment s + 2 are computed (the
    // divisions, the bounds tests) while the loads of s + 1 are in flight, the loads of s + 1 are issued right behind the barrier that
    // frees the registers, and the keep bits of s are formed behind them.
    // The masks: a segment needs one byte per source row and one per output row and plane (a 256-byte segment touches at most three
    // planes, D >= 128) -- 1024 bytes, one per THREAD, handed round through LDS as 0 / 1 flags.  (Every thread fetching the bytes of
    // its own four source rows and four output rows was 8 - 12 more memory instructions per thread and segment, each a full pass of
    // the address unit whatever it carries: 9 us of a 47 us launch at 355 x 512 x 355.)
    u32 moff[4];           // pixel n0 * H + y of local row j's source row (its data start at moff * D + scol); ~0: absent
    u32 mmoff;             // this thread's mask byte: offset into mask_src (group 0) / mask_dst; ~0: absent (flag 0)
    int az; u32 ay;        // the output piece's first byte: z (-1: absent) and plane
    u32 aya;               // plane of the segment's first byte
    auto addr_seg = [&](i64 s) {
        const bool live = s < s_end;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 4) + 64 * j;
            const u32 f = (u32)(256 * s) + (u32)(255 - lr);
            const u32 y = pb3d_div(f, mD), z = f - y * mD.d;
            const i64 n0 = (i64)c0 - (i64)z;
            const bool ok = live && (i64)f < HD && cmode != 0 && n0 >= 0 && n0 < W;
            moff[j] = ok ? (u32)(n0 * H + (i64)y) : 0xffffffffu;
        }
        const u32 f = (u32)(256 * s) + (u32)(16 * zg);
        const bool pin = live && xrow_ok && (i64)f < HD;
        const u32 y = pin ? pb3d_div(f, mD) : 0u;
        az = pin ? (int)(f - y * mD.d) : -1; ay = y;
        aya = pb3d_div((u32)(256 * s), mD);
        {
            const u32 fm = (u32)(256 * s) + (mgrp == 0 ? (u32)(255 - mrow) : 0u);
            const u32 ym = pb3d_div(fm, mD), zm = fm - ym * mD.d;
            const i64 n0 = (i64)c0 - (i64)zm, x = x0 + mrow, yp = (i64)ym + mgrp - 1;
            const bool ok = live && (mgrp == 0 ? ((i64)fm < HD && n0 >= 0 && n0 < W) : (mrow < XW && x < W && yp <= (i64)Y1));
            if (blk_on) {
                const i64 yq = (mgrp == 0 ? (i64)ym : yp) - (i64)Y0, r = mgrp == 0 ? n0 : W + mrow;
                mmoff = ok ? (u32)(((yq >> 3) * R + r) * 8 + (yq & 7)) : 0xffffffffu;
            } else mmoff = ok ? (u32)(mgrp == 0 ? n0 * H + (i64)ym : x * H + yp) : 0xffffffffu;
        }
    };
    u32x4 stg[4];
    u32 mraw;              // this thread's mask byte, as loaded (0x100: no mask, all set)
    int pz; u32 py, pya;   // az, ay, aya of the segment in flight
    auto issue_seg = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            stg[j] = (u32x4)(0u);
            if (moff[j] != 0xffffffffu) {
                const i64 doff = (i64)moff[j] * D + scol;
                const u8* sp = in + doff;
                // a ragged block may be read whole as long as it stays inside the volume: the bytes beyond the row belong to the
                // neighbouring row and are dropped by the validity bits (their source column is outside [0, D))
                if (cmode == 2 || (doff >= 0 && doff + 16 <= W * HD)) stg[j] = __builtin_nontemporal_load((const u32x4_u*)sp);
                else {                                                     // the ragged block of the last tile: byte by byte inside [0, D)
                    u32 t4[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; ++b)
                        if (scol + b >= 0 && scol + b < D) t4[b >> 2] |= (u32)sp[b] << (8 * (b & 3));
                    stg[j].x = t4[0]; stg[j].y = t4[1]; stg[j].z = t4[2]; stg[j].w = t4[3];
                }
            }
        }
        if (!blk_on) {
            const u8* mp = mgrp == 0 ? mask_src : mask_dst;
            mraw = mmoff == 0xffffffffu ? 0u : (mp ? (u32)mp[mmoff] : 0x100u);
        }
        pz = az; py = ay; pya = aya;
    };
    auto issue_mask = [&]() { if (blk_on) mraw = mmoff == 0xffffffffu ? 0u : (u32)blk[mmoff]; };
    if (blk_on) { fill_block(); __syncthreads(); }
    addr_seg(s_beg);
    issue_seg();
    issue_mask();
    addr_seg(s_beg + 1);
    for (i64 s = s_beg; s < s_end; ++s) {
        // ---- the segment's data has arrived: into LDS, with the mask flags
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = (tid >> 4) + 64 * j;
            *(u32x4*)(wtile + lr * 256 + 16 * (cb ^ ((lr >> 4) & 15))) = stg[j];
        }
        msh[tid] = (u8)(mraw != 0);
        const int z = pz;
        const int pa = (int)(py - pya);                   // the piece's plane within the segment's (0 .. 2)
        __syncthreads();
        issue_seg();                                      // segment s + 1 (addresses ready)
        issue_mask();
        // ---- keep bits of this thread's four output pieces of segment s: validity, destination rows' masks, source rows' masks
        u32 kcur[4];
        {
            const int nA = z + 16 <= (int)D ? 16 : (int)D - z;
            const u32 lowA = (1u << nA) - 1u;
            const u32x4 sm = *(const u32x4*)(msh + 16 * g);               // flags of source rows 16 g .. 16 g + 15; byte q of the piece comes from row 16 g + 15 - q
            const u32 asc = (((sm.x & 0x01010101u) * 0x01020408u) >> 24) | ((((sm.y & 0x01010101u) * 0x01020408u) >> 24) << 4) |
                            ((((sm.z & 0x01010101u) * 0x01020408u) >> 24) << 8) | ((((sm.w & 0x01010101u) * 0x01020408u) >> 24) << 12);
            const u32 srcbits = __brev(asc) >> 16;
            const u32 mdA = z >= 0 ? *(const u32*)(msh + 256 + 256 * pa + 4 * xg) : 0u;
            const u32 mdB = (z >= 0 && nA < 16) ? *(const u32*)(msh + 256 + 256 * (pa + 1) + 4 * xg) : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32 k = 0;
                if (z >= 0) {
                    u32 bits;
                    if (iv[i] != 0xffffffffu) {
                        const int lo = (int)(iv[i] & 0xffffu), hi = (int)(iv[i] >> 16);
                        bits = (bits_between(lo - z, hi - z) & lowA) | (bits_between(lo - (z - (int)D), hi - (z - (int)D)) & ~lowA & 0xffffu);
                    } else { bits = vwin(x0 + 4 * xg + i, (i64)z); if (nA < 16) bits |= vwin(x0 + 4 * xg + i, (i64)z - D); }
                    const u32 mA = ((mdA >> (8 * i)) & 0xffu) ? lowA : 0u;
                    const u32 mB = ((mdB >> (8 * i)) & 0xffu) ? (0xffffu & ~lowA) : 0u;
                    k = bits & (mA | mB) & srcbits;
                }
                kcur[i] = k;
            }
        }
        u32 d[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) d[rr] = *(const u32*)(wtile + rd_off + rr * 256);
        __syncthreads();                                  // the tile and the flags are free again: the next segment's may be written
        addr_seg(s + 2);
        if (!xrow_ok || z < 0) continue;                                   // (a piece past the stream's end: ragged last segment only)
        u32 o[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32 A = d[15 - 4 * w], B = d[14 - 4 * w], Cc = d[13 - 4 * w], E = d[12 - 4 * w];
            const u32 t0 = perm(B, A, 0x05010400u), t1 = perm(B, A, 0x07030602u);
            const u32 u0 = perm(E, Cc, 0x05010400u), u1 = perm(E, Cc, 0x07030602u);
            o[0][w] = perm(u0, t0, 0x05040100u);
            o[1][w] = perm(u0, t0, 0x07060302u);
            o[2][w] = perm(u1, t1, 0x05040100u);
            o[3][w] = perm(u1, t1, 0x07060302u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i64 x = x0 + 4 * xg + i;
            if (x >= W) continue;
            const u32 kb = kcur[i];
            u32x4 r = (u32x4)(0u);
            if (kb) {
                r.x = o[i][0]; r.y = o[i][1]; r.z = o[i][2]; r.w = o[i][3];
                if (kb != 0xffffu) {
                    u32 mw[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 b4 = (kb >> (4 * w)) & 0xfu;
                        mw[w] = ((b4 & 1u) ? 0x000000ffu : 0u) | ((b4 & 2u) ? 0x0000ff00u : 0u) | ((b4 & 4u) ? 0x00ff0000u : 0u) |
                                ((b4 & 8u) ? 0xff000000u : 0u);
                    }
                    r.x &= mw[0]; r.y &= mw[1]; r.z &= mw[2]; r.w &= mw[3];
                }
            }
            __builtin_nontemporal_store(r, (u32x4*)(out + x * HD + 256 * s + 16 * zg));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5: part_carve with 90-degree jobs, all jobs in ONE sweep.  Every job writes colored[v] itself where
// it keeps (reference :152-158), so the overlay is the union of the jobs' keep sets:
//   keep[x,y,z] = valid(x,z) && occ[c0 - z, y, x + c2] && (A[x,y] & A[c0 - z, y]) != 0
// with occ = any(colored > 0) and A[x,y] = bitset over jobs of (mask_sub_j && mask_carve_j)[x,y]
// (both the source-side and the destination-side carve of a job use its own masks).
// The sweep itself is k_part90_plane in csrc/bits90.hip (round 4: occupancy BITS of a plane transposed, whole output rows); the byte-tile
// kernels k_part90 / k_part90_flat of rounds 1-3 are gone -- slower on every shape (profiles/r04_part_carve_plane_kernel_vs_tile_kernels.jsonl).
// ------------------------------------------------------------------------------------------------
// job_on: bit j = job j takes part (passed by value: a device copy of the flags cost a memcpy and a stream synchronisation per call)
// AT (optional): the same sets in (y, x) order, for the plane-wise pass of the stream form (csrc/bits90.hip, k_part90_keep)
__global__ __launch_bounds__(256) void k_job_bitset(const u8* __restrict__ mask_sub, const u8* __restrict__ mask_carve,
                                                    u32 job_on, int nj, i64 npix, u32* __restrict__ A, u32* __restrict__ AT, i64 W, i64 H) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (i64)gridDim.x * blockDim.x) {
        u32 a = 0;
        for (int j = 0; j < nj; ++j)
            if (((job_on >> j) & 1u) && mask_sub[(i64)j * npix + i] && mask_carve[(i64)j * npix + i]) a |= 1u << j;
        A[i] = a;
        if (AT) { const i64 x = i / H, y = i - x * H; AT[y * W + x] = a; }
    }
}

}  // namespace

// true when the (x,z) part of (M, off) is a signed permutation up to 2^-40 with an offset within
// 2^-20 of an integer: every coordinate of every voxel (< 2^20 per axis) is then within 2^-18 of an
// integer and the step is an exact byte permutation (see the header of this file).
// a permutation-like step that pb3d_launch_rotate_perm can run on these buffers: the 90-degree map has no alignment or
// size condition (k_rot90); the other signed permutations (180, 270 degrees) use the dword kernel k_rotate_perm
bool pb3d_perm_step_ok(const double M[9], const double off[3], i64 W, i64 D, const void* a, const void* b) {
    if (!pb3d_is_perm_step(M, off, W, D)) return false;
    const bool rot90 = nearbyint(M[0]) == 0 && nearbyint(M[2]) == -1 && nearbyint(M[6]) == 1 && nearbyint(M[8]) == 0;
    return rot90 || (D % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 3u) == 0);
}

bool pb3d_is_perm_step(const double M[9], const double off[3], i64 W, i64 D) {
    if (W >= (1ll << 20) || D >= (1ll << 20)) return false;
    const int idx[4] = {0, 2, 6, 8};
    int r[4];
    for (int k = 0; k < 4; ++k) {
        const double v = M[idx[k]], n = nearbyint(v);
        if (fabs(v - n) > 0x1p-40 || fabs(n) > 1.0) return false;
        r[k] = (int)n;
    }
    if (abs(r[0]) + abs(r[1]) != 1 || abs(r[2]) + abs(r[3]) != 1 || abs(r[0]) + abs(r[2]) != 1) return false;
    for (int h = 0; h < 3; h += 2)
        if (fabs(off[h] - nearbyint(off[h])) > 0x1p-20) return false;
    return true;
}

// integer form of a permutation-like step: n0 = r00*x + r02*z + c0, n2 = r20*x + r22*z + c2
struct PermMap {
    int r00, r02, r20, r22, c0, c2;
};

static PermMap perm_map(const double M[9], const double off[3]) {
    PermMap m;
    m.r00 = (int)nearbyint(M[0]); m.r02 = (int)nearbyint(M[2]); m.r20 = (int)nearbyint(M[6]); m.r22 = (int)nearbyint(M[8]);
    m.c0 = (int)nearbyint(off[0]); m.c2 = (int)nearbyint(off[2]);
    return m;
}

static int build_valid_table(pb3d_ctx* ctx, const RotParams& p, i64 W, i64 D, u32** bits, int* nw) {
    const int n = (int)(((D + 63) / 64) * 2 + 2);
    void* buf;
    PB3D_TRY(pb3d_scratch(ctx, 10, (size_t)W * (n + 1) * sizeof(u32), &buf));           // + one interval word per row (pb3d_valid_ivals)
    // the table depends on (matrix, offset, W, D) only and slot 10 is private to it: a repeated step (every 90-degree call on one shape)
    // finds it in place -- at 512-class sizes the memset + table kernel were 10 % of a process_voxel_grid(., ., 90) call
    pb3d_ctx::ValidCache& vc = ctx->valid_cache;
    if (vc.buf == buf && vc.gen == ctx->scratch_gen && vc.W == W && vc.D == D && memcmp(vc.p, &p, sizeof(RotParams)) == 0 && ctx->tune_no_table_cache != 1) {
        *bits = (u32*)buf; *nw = n;
        return PB3D_OK;
    }
    vc.buf = nullptr;
    PB3D_HIP(hipMemsetAsync(buf, 0, (size_t)W * n * sizeof(u32), ctx->stream));
    dim3 grid((unsigned)((D + 255) / 256), (unsigned)W);
    hipLaunchKernelGGL(k_rot_valid, grid, dim3(256), 0, ctx->stream, p, W, D, n, (u32*)buf);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_rot_ivals, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, ctx->stream, (const u32*)buf, W, D, n, (u32*)buf + W * n);
    PB3D_CHECK_LAUNCH();
    vc.buf = buf; vc.gen = ctx->scratch_gen; vc.W = W; vc.D = D;
    static_assert(sizeof(vc.p) == sizeof(RotParams), "ValidCache holds one RotParams");
    memcpy(vc.p, &p, sizeof(RotParams));
    *bits = (u32*)buf; *nw = n;
    return PB3D_OK;
}

// the validity bit table of a permutation-like step for other translation units (csrc/sliced.hip: the chain's last step un-slices)
int pb3d_perm_valid_table(pb3d_ctx* ctx, const double M[9], const double off[3], i64 W, i64 D, u32** bits, int* nw, int* c0, int* c2, bool* rot90) {
    const RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    const PermMap pm = perm_map(M, off);
    *c0 = pm.c0; *c2 = pm.c2;
    *rot90 = pm.r00 == 0 && pm.r02 == -1 && pm.r20 == 1 && pm.r22 == 0;
    return build_valid_table(ctx, p, W, D, bits, nw);
}

int pb3d_launch_rotate_perm(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const double M[9], const double off[3],
                            const u8* d_mask_src, const u8* d_mask_dst, u8* d_out) {
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    const PermMap pm = perm_map(M, off);
    const bool rot90 = pm.r00 == 0 && pm.r02 == -1 && pm.r20 == 1 && pm.r22 == 0;
    PB3D_REQUIRE(rot90 || (D % 4 == 0 && (((uintptr_t)d_in | (uintptr_t)d_out) & 3u) == 0), "pb3d_rotate_perm: needs D %% 4 == 0");
    if (rot90 && W <= 65535 * 128 && H <= 65535) {
        u32* bits; int nw;
        PB3D_TRY(build_valid_table(ctx, p, W, D, &bits, &nw));
        // rows that are not whole lines: the stream of each x-row tiled in whole lines (k_rot90_flat / k_rot90wf) -- streams that are whole lines
        // (H * D % 128 == 0: every real shape of the reference whose longer mask side is the height) or at least whole 16-byte pieces
        // (H * D % 16 == 0, e.g. 500 x 400 x 500: the last segment of an x-row's stream is ragged, the rows of odd x start mid-line);
        // knob rot90_flat: 1 = the row-wise tile kernel instead, 2 = whole-line streams only
        const bool lines_ok = D % 128 != 0 && (H * D) % 128 == 0 && D >= 128 && ctx->tune_rot90_flat != 1;
        const bool flat16 = D % 128 != 0 && (H * D) % 16 == 0 && D >= 128 && ctx->tune_rot90_flat == 0;
        const bool flat = (lines_ok || flat16) && H * D < (1ll << 31) - 256 && (((uintptr_t)d_out) & 127u) == 0;
        const i64 nzt = (D + 127) / 128;
        const i64 tiles = nzt * ((W + 127) / 128);
        const int TY = planes_per_chunk(H, tiles, ctx->cus, 32, ctx->tune_rot90_fill);
        const TileMap tm = {(int)nzt, (int)((W + 127) / 128), (int)((H + TY - 1) / TY), ctx->tune_rot90_order == 1 ? 1 : 0};
        dim3 grid(tilemap_blocks(tm));
#ifndef PB3D_ROT90_DEPTH
#define PB3D_ROT90_DEPTH 1
#endif
        if (flat && W >= 160 && H * D < (1ll << 31) - 1024 && W * H < 0xffffffffll && ctx->tune_rot90_wide != 2) {
            // 256-byte segments, 1024 threads (tune rot90_wide = 2: the 128-row x 128-byte form below)
            if (!ctx->rot90wf_lds_set) {
                PB3D_HIP(hipFuncSetAttribute((const void*)k_rot90wf, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 256 + 48 * 1024));
                ctx->rot90wf_lds_set = true;
            }
            const i64 nx0 = (W + 255) / 256;
            const int npc = (int)((((W + nx0 - 1) / nx0) + 15) / 16);          // 16-byte blocks of a tile's source rows (10 .. 16)
            const i64 nseg = (H * D + 255) / 256, nxt = (W + 16 * npc - 1) / (16 * npc);
            // workgroups per CU: as many (up to 4) as leave a workgroup at least 8 segments (fewer, longer workgroups win at these sizes)
            int fillw = ctx->tune_rot90_fill > 0 ? ctx->tune_rot90_fill : 4;
            if (ctx->tune_rot90_fill <= 0)
                while (fillw > 1 && planes_per_chunk(nseg, nxt, ctx->cus, 32, fillw) < 8) fillw >>= 1;
            const int TS = planes_per_chunk(nseg, nxt, ctx->cus, 32, fillw);
            const TileMap fm = {1, (int)nxt, (int)((nseg + TS - 1) / TS), 0};
            // the workgroup's mask block behind the tile: (W + XW) rows x 8 planes per window, windows for the planes TS segments span
            const i64 nwin = ((i64)TS * 256 / D + 2) / 8 + 1;
            const size_t blk_bytes = (size_t)((W + 16 * npc) * 8 * nwin);
            const int blk_on = blk_bytes <= 48 * 1024 && ctx->tune_rot90_mask_block != 1;
            hipLaunchKernelGGL(k_rot90wf, dim3(tilemap_blocks(fm)), dim3(1024), 256 * 256 + (blk_on ? blk_bytes : 0), ctx->stream, d_in, d_out, d_mask_src, d_mask_dst,
                               (const u32*)bits, (const u32*)bits + W * nw, nw, pm.c0, pm.c2, W, H, D, TS, fm, pb3d_make_magic((u32)D), nseg, npc, blk_on);
        } else if (flat) {
            const i64 nseg = (H * D + 127) / 128, nxt = (W + 127) / 128;
            const int TS = planes_per_chunk(nseg, nxt, ctx->cus, 32, ctx->tune_rot90_fill);
            const TileMap fm = {1, (int)nxt, (int)((nseg + TS - 1) / TS), ctx->tune_rot90_order == 1 ? 1 : 0};
            hipLaunchKernelGGL(k_rot90_flat, dim3(tilemap_blocks(fm)), dim3(256), 0, ctx->stream, d_in, d_out, d_mask_src, d_mask_dst, (const u32*)bits, nw,
                               pm.c0, pm.c2, W, H, D, TS, fm, pb3d_make_magic((u32)D), nseg);
        } else if (D % 16 == 0 && pm.c2 % 16 == 0 && W >= 256 && D >= 256 && ctx->tune_rot90_wide != 2 && (((uintptr_t)d_in | (uintptr_t)d_out) & 15u) == 0) {
            // the 256 x 256-tile form (tune rot90_wide = 2: the 128-tile kernel).  Measured with tools/tybench.py on one box, variants
            // interleaved (ms, 128-tile kernel -> this one): 1024^3 0.464 -> 0.430, 512^3 0.0665 -> 0.0608, 512 x 278 x 512 0.046 -> 0.035.
            // Workgroups per CU: as many (up to 8) as leave a workgroup at least 4 planes.  With the mask flags in LDS (tools/tybench.py,
            // fills interleaved, ms): 1024^3 likes 8 (0.421; 4: 0.439, 2: 0.446, 1: 0.447), 512^3 2 (0.053; 1: 0.077, 4: 0.056, 8: 0.063),
            // 512 x 278 x 512 1 (0.029; 2 and more: 0.033).
            if (!ctx->rot90w_lds_set) {
                PB3D_HIP(hipFuncSetAttribute((const void*)k_rot90w, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 256 + 512 * 64));
                ctx->rot90w_lds_set = true;
            }
            const i64 wt = ((D + 255) / 256) * ((W + 255) / 256);
            int fillw = ctx->tune_rot90_fill > 0 ? ctx->tune_rot90_fill : 8;
            if (ctx->tune_rot90_fill <= 0)
                while (fillw > 1 && planes_per_chunk(H, wt, ctx->cus, 32, fillw) < 4) fillw >>= 1;
            const int TYw = planes_per_chunk(H, wt, ctx->cus, 32, fillw);
            const TileMap wm = {(int)((D + 255) / 256), (int)((W + 255) / 256), (int)((H + TYw - 1) / TYw), 0};
            const int blk_on = TYw <= 64 && ctx->tune_rot90_mask_block != 1;                 // the workgroup's mask flags behind the tile: 512 bytes per plane
            hipLaunchKernelGGL(k_rot90w, dim3(tilemap_blocks(wm)), dim3(1024), 256 * 256 + (blk_on ? 512 * TYw : 0), ctx->stream, d_in, d_out, d_mask_src, d_mask_dst,
                               (const u32*)bits, nw, pm.c0, pm.c2, W, H, D, TYw, wm, blk_on);
        } else if (D % 16 == 0 && pm.c2 % 16 == 0)
            hipLaunchKernelGGL((k_rot90<PB3D_ROT90_DEPTH, false>), grid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_src, d_mask_dst,
                               (const u32*)bits, nw, pm.c0, pm.c2, W, H, D, TY, tm);
        else
            hipLaunchKernelGGL((k_rot90<PB3D_ROT90_DEPTH, true>), grid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_src, d_mask_dst,
                               (const u32*)bits, nw, pm.c0, pm.c2, W, H, D, TY, tm);
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    int TY = 32;
    const i64 tiles = ((D + T - 1) / T) * ((W + T - 1) / T);
    while (TY > 1 && tiles * ((H + TY - 1) / TY) < (i64)ctx->cus * 8) TY >>= 1;
    dim3 grid((unsigned)((D + T - 1) / T), (unsigned)((W + T - 1) / T), (unsigned)((H + TY - 1) / TY));
    PB3D_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, "pb3d_rotate_perm: grid too large");
    hipLaunchKernelGGL(k_rotate_perm, grid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_src, d_mask_dst, p, W, H, D, TY);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

// fused global_carve for angle_interval == 90.  Output slab x in [x0, x1).
int pb3d_launch_global_carve90(pb3d_ctx* ctx, const u8* d_bin_hw, const u8* d_rgb_hw3, int C, i64 h, i64 w, const double M[9],
                               const double off[3], i64 x0, i64 x1, u8* d_out_slab) {
    const i64 W = w, H = h, D = w;
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    const PermMap pm = perm_map(M, off);
    PB3D_REQUIRE(pm.r00 == 0 && pm.r02 == -1, "pb3d_global_carve: unexpected 90-degree map");
    u32* bits; int nw;
    PB3D_TRY(build_valid_table(ctx, p, W, D, &bits, &nw));
    return pb3d_launch_gc90_stream(ctx, d_bin_hw, d_rgb_hw3, C, (const u32*)bits, nw, pm.c0, W, H, D, x0, x1, d_out_slab);
}

// All-90-degree part_carve in one sweep (K5).  Returns PB3D_EUNSUPPORTED (without an error message of
// its own) when the fast-path conditions do not hold; the caller then runs the per-job pipeline.
int pb3d_try_part_carve90(pb3d_ctx* ctx, const u8* d_colored, int C, i64 W, i64 H, i64 D, const u8* d_mask_sub, const u8* d_mask_carve,
                          const int* job_angle, const int* job_skip, int njobs, u8* d_out) {
    if (njobs > 32 || njobs <= 0) return PB3D_EUNSUPPORTED;
    bool any = false;
    for (int j = 0; j < njobs; ++j) {
        if (job_skip[j]) continue;
        if (job_angle[j] != 90) return PB3D_EUNSUPPORTED;
        any = true;
    }
    if (!any) return PB3D_EUNSUPPORTED;
    const i64 shape[3] = {W, H, D};
    double M[9], off[3];
    PB3D_TRY(pb3d_rotinv(90, M));
    PB3D_TRY(pb3d_offset(M, shape, off));
    if (!pb3d_is_perm_step(M, off, W, D)) return PB3D_EUNSUPPORTED;
    const PermMap pm = perm_map(M, off);
    const bool rot90 = pm.r00 == 0 && pm.r02 == -1 && pm.r20 == 1 && pm.r22 == 0;
    if (!(rot90 && W <= 65535 * 128 && H <= 65535)) return PB3D_EUNSUPPORTED;
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    u32* bits; int nw;
    PB3D_TRY(build_valid_table(ctx, p, W, D, &bits, &nw));
    void *A, *AT;
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)(W * H) * sizeof(u32), &A));
    PB3D_TRY(pb3d_scratch(ctx, 46, (size_t)(W * H) * sizeof(u32), &AT));
    u32 on = 0;
    int nj = 0;
    for (int j = 0; j < 32; ++j) {
        const bool live = j < njobs && !job_skip[j];
        on |= (u32)(live ? 1 : 0) << j;
        if (live) nj = j + 1;
    }
    hipLaunchKernelGGL(k_job_bitset, dim3(pb3d_stream_blocks(ctx, W * H, 256, 8)), dim3(256), 0, ctx->stream, d_mask_sub, d_mask_carve, on, njobs, W * H,
                       (u32*)A, (u32*)AT, W, H);
    PB3D_CHECK_LAUNCH();
    int took = 0;
    PB3D_TRY(pb3d_part_carve90_planes(ctx, d_colored, C, W, H, D, (const u32*)A, (const u32*)AT, nj, (const u32*)bits, nw, pm.c0, pm.c2, d_out, &took));
    return took ? PB3D_OK : PB3D_EUNSUPPORTED;          // (a plane of bits that does not fit the LDS: the caller's per-job pipeline)
}
