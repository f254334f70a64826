// Chained rotation steps in a BIT-SLICED occupancy layout ("S32").
//
// process_voxel_grid (reference utils/voxel_carving_utils.py:104-126) rotates the already rotated and carved grid again and again:
// range(0, 91, k) is 19 steps at the notebook's k = 5 (:193, notebook 1 cell 7).  On 0/1 occupancy every step is a table lookup on
// four taps (rot_common.h: lut_of), and a Y-plane never mixes with another (the rotation is about Y), so between two steps the
// volume does not have to be bytes at all:
//
//     S32[g][x][z]  (u32, pitch Dp = D rounded up to 16)      bit q = occ[x][32 g + q][z],   g < G = ceil(H / 32)
//
// is 1/8 byte per voxel, a middle step reads and writes 1/4 B/voxel instead of 2, and neither side converts anything: the staged
// footprint of a tile IS the dword the 32-plane multiplexer tree wants, and the tree's result IS the dword that is stored.  The chain:
//     k_s32_slice (u8 -> S32, the 0-degree carve folded in as a bit mask) -> k_s32_step per rotation -> k_s32_unslice (S32 -> u8)
// plus ONE k_s32_prep launch that compiles every (step, tile) pair's "tile program" (LDS offsets + result tables of its 4096 cells,
// the list of 16-byte units of its rotated footprint) from SciPy's f64 arithmetic.  The programs depend on (angles, W, D) only and
// are cached across calls.  Data that is not 0/1 raises a flag in k_s32_slice: the caller then runs the byte chain (rotate.hip).
#include "rot_common.h"

namespace {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a1 __attribute__((aligned(1)));     // 8-byte access at any byte alignment (rows of odd-sized grids)

constexpr int ST = 64;              // output tile edge (x and z)
constexpr int STHREADS = 256;       // 16 cells per thread: 4 consecutive z in each of 4 rows
constexpr int SROWS = 96;           // footprint rows in LDS (a rotated 64-tile spans <= 64 sqrt 2 + 2 = 92.6)
// LDS row pitch in dwords: ODD, so the 4 rows x 16 column groups a wave reads together hit 64 banks at 0 and 90 degrees.  At the angles in
// between the taps of a wave walk the footprint diagonally and WHICH odd pitch matters: 97 / 99 / 101 / 103 interleaved on one box at 1024^3
// (tools/slicedbench.py, builds with -DPB3D_SPITCH=...): 18 sweeps 1.72 / 1.66 / 1.63 / 1.69 ms, the 45-degree chain 0.586 / 0.540 / 0.535 /
// 0.551 -- 101 it is (96 x 101 dwords = 38.8 KB: still four workgroups per CU; 105 would leave three).
#ifndef PB3D_SPITCH
#define PB3D_SPITCH 101
#endif
constexpr int SPITCH = PB3D_SPITCH;
constexpr int SMAXU = 9;            // 16-byte units per thread: a full bounding box of 95 rows x 24 units
constexpr int SMAXSTEPS = 32;       // steps per table build

// the compiled form of one (step, tile): what a workgroup of k_s32_step needs besides the voxels
struct STile {
    int bx0, bz0, nrows, nunits, fits, pad0, pad1, pad2;
    u32 ugoff[SMAXU * STHREADS];             // unit u: dword offset of its 4 voxels inside a plane group, (bx0 + r) * Dp + bz0 + 4 cu
    unsigned short uloff[SMAXU * STHREADS];  // ... and inside the LDS footprint, r * SPITCH + 4 cu
    u32 cellw[16 * STHREADS];                // [4 i + c][tid]: result table << 16 | LDS dword offset of tap (0,0); 0 = void cell
};
static_assert(sizeof(STile) % 16 == 0, "STile is read with 16-byte loads");

struct StepParams { RotParams p[SMAXSTEPS]; };

// bits[g * W + x]: bit q = mask_wh[x, 32 g + q] != 0 (zero for planes past H); also clears the chain's flag word
__global__ __launch_bounds__(256) void k_s32_maskbits(const u8* __restrict__ mask_wh, i64 W, i64 H, int G, u32* __restrict__ bits, int* __restrict__ flag) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && flag) *flag = 0;
    if (i >= (i64)G * W) return;
    const i64 g = i / W, x = i - g * W, y0 = 32 * g;
    u32 b = 0;
    if (y0 + 32 <= H) {
        // a whole group: its 32 mask bytes are neighbours in memory -- two 16-byte loads at whatever alignment, non-zero bytes gathered into bits
        // (a loop of 32 dependent byte loads made this 128 KB kernel 8 us long at 1024^2: as much as a tenth of a rotation step)
        typedef u32 u32x4_m1 __attribute__((ext_vector_type(4), aligned(1)));
        const u32x4_m1 lo = *(const u32x4_m1*)(mask_wh + x * H + y0), hi = *(const u32x4_m1*)(mask_wh + x * H + y0 + 16);
        const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 t = w[j];
            const u32 z = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;      // bit 7 of a byte: the byte is non-zero
            b |= (((z >> 7) * 0x01020408u) >> 24) << (4 * j);
        }
    } else
        for (int q = 0; q < 32 && y0 + q < H; ++q) b |= (u32)(mask_wh[x * H + y0 + q] != 0) << q;
    bits[i] = b;
}

// u8 (W,H,D) -> S32.  One thread = 16 consecutive z of one (g, x): 32 planes x 16 bytes in (16-byte loads at whatever alignment the row
// has), 16 dwords out.  mbits (optional): the 0-degree carve of process_voxel_grid (reference :111-124, first iteration) as a bit mask.
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
// WHOLE: D % 16 == 0, every thread's 16 voxels lie inside its row -- the loads are unconditional and all in flight together (a plane past
// H re-reads the group's last plane, its bits are masked off).  Written with a branch per plane ("is it inside H", "is the piece
// whole") every load sat in its own exec region with a wait behind it: 32 serial round trips per thread, 252 us at 1024^3.
// NT: the grid is read with nontemporal loads -- at 1024^3 (1 GB read once) the pass went 252 -> ~200 us, at 512-class sizes (the grid
// sits in the 256 MB memory-side cache from the kernel before) they cost 5 %: chosen by size.
template <bool WHOLE, bool NT>
__global__ __launch_bounds__(256) void k_s32_slice(const u8* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ mbits, i64 W, i64 H,
                                                   i64 D, i64 Dp, pb3d_magic mzb, pb3d_magic mw, u32 total, int* __restrict__ flag, pb3d_magic mg, int gfast) {
    const u32 idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    // (gfast: consecutive wavefronts take consecutive plane groups of one x, their 32 KB source blocks following each other in memory,
    // instead of blocks H * D bytes apart -- a development A/B that measured no difference)
    const u32 r2 = pb3d_div(idx, mzb), zb = idx - r2 * mzb.d;
    u32 g, x;
    if (gfast) { x = pb3d_div(r2, mg); g = r2 - x * mg.d; } else { g = pb3d_div(r2, mw); x = r2 - g * mw.d; }
    const u32 row = g * mw.d + x;
    const i64 z = 16 * (i64)zb;
    const int np = (int)(H - 32 * (i64)g < 32 ? H - 32 * (i64)g : 32);
    const u8* base = in + ((i64)x * H + 32 * (i64)g) * D + z;
    const bool whole = WHOLE || z + 16 <= D;
    u32 hib = 0;
    u32 w[4][4];                    // w[k][j]: byte c = planes 8 k .. 8 k + 7 of voxel z + 4 j + c
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32x4 d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pl = 8 * k + q < np ? 8 * k + q : np - 1;
            const u8* p = base + (i64)pl * D;
            if (whole) d[q] = NT ? __builtin_nontemporal_load((const u32x4_a1*)p) : *(const u32x4_a1*)p;
            else {
                u32 t4[4] = {0, 0, 0, 0};
                for (int b = 0; b < 16 && z + b < D; ++b) t4[b >> 2] |= (u32)p[b] << (8 * (b & 3));
                d[q].x = t4[0]; d[q].y = t4[1]; d[q].z = t4[2]; d[q].w = t4[3];
            }
        }
        u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32 live = 8 * k + q < np ? 0xffffffffu : 0u;
            const u32 dx = d[q].x & live, dy = d[q].y & live, dz = d[q].z & live, dw = d[q].w & live;
            a0 |= dx << q; a1 |= dy << q; a2 |= dz << q; a3 |= dw << q;
            hib |= (dx | dy) | (dz | dw);
        }
        w[k][0] = a0; w[k][1] = a1; w[k][2] = a2; w[k][3] = a3;
    }
    const u32 m = mbits ? mbits[row] : 0xffffffffu;
    u32* op = out + (i64)row * Dp + z;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32 v[4];
        tr4x4(w[0][j], w[1][j], w[2][j], w[3][j], v);      // v[c]: byte k = planes 8 k .. 8 k + 7 of voxel z + 4 j + c
        u32x4 o; o.x = v[0] & m; o.y = v[1] & m; o.z = v[2] & m; o.w = v[3] & m;
        *(u32x4*)(op + 4 * j) = o;
    }
    // a value other than 0 / 1: the chain cannot represent it (one relaxed look first: on such data every thread would hit the word)
    if ((hib & 0xfefefefeu) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
}

// S32 -> u8 (W,H,D): the inverse, same thread shape
__global__ __launch_bounds__(256) void k_s32_unslice(const u32* __restrict__ in, u8* __restrict__ out, i64 W, i64 H, i64 D, i64 Dp, pb3d_magic mzb,
                                                     pb3d_magic mw, u32 total, pb3d_magic mg, int gfast) {
    const u32 idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const u32 r2 = pb3d_div(idx, mzb), zb = idx - r2 * mzb.d;
    u32 g, x;
    if (gfast) { x = pb3d_div(r2, mg); g = r2 - x * mg.d; } else { g = pb3d_div(r2, mw); x = r2 - g * mw.d; }
    const u32 row = g * mw.d + x;
    const i64 z = 16 * (i64)zb;
    const int np = (int)(H - 32 * (i64)g < 32 ? H - 32 * (i64)g : 32);
    const u32* ip = in + (i64)row * Dp + z;
    u32 v[4][4];                    // v[j][k]: byte c = planes 8 k .. 8 k + 7 of voxel z + 4 j + c
#pragma unroll
    for (int j = 0; j < 4; ++j) { const u32x4 a = *(const u32x4*)(ip + 4 * j); tr4x4(a.x, a.y, a.z, a.w, v[j]); }
    u8* base = out + ((i64)x * H + 32 * (i64)g) * D + z;
    const bool whole = z + 16 <= D;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q >= np) continue;
        u32x4 r;
        r.x = (v[0][q >> 3] >> (q & 7)) & 0x01010101u; r.y = (v[1][q >> 3] >> (q & 7)) & 0x01010101u;
        r.z = (v[2][q >> 3] >> (q & 7)) & 0x01010101u; r.w = (v[3][q >> 3] >> (q & 7)) & 0x01010101u;
        u8* dp = base + (i64)q * D;
        if (whole) *(u32x4_a1*)dp = r;
        else {
            const u32 t4[4] = {r.x, r.y, r.z, r.w};
            for (int bb = 0; bb < 16 && z + bb < D; ++bb) dp[bb] = (u8)(t4[bb >> 2] >> (8 * (bb & 3)));
        }
    }
}

// The chain's LAST step, when it is the 90-degree one (every angle step that divides 90: reference :111 range(0, 91, k)), un-slices in its own
// stores: out[x, y, z] = valid(x, z) && mask[x, y] ? S[c0 - z][y][x + c2] : 0 -- on 0/1 data a permutation-like step is the nearest tap
// wherever SciPy's bounds test passes (DESIGN.md "Exactness of the permutation-like steps"; the bit table of k_rot_valid holds the
// test's f64 verdicts).  One rotation sweep (0.35 GB at 1024^3) and one launch less per chain.  Workgroup = 32 x-rows x 128 z of one plane
// group: the source block (128 S32 rows x 32 dwords = 128-byte pieces) is staged in LDS and read back transposed; a wave's plane store
// covers 8 rows x 128 bytes.
constexpr int UX = 32, UZ = 128, UPITCH = UX + 1;
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

__global__ __launch_bounds__(256) void k_s32_unslice90(const u32* __restrict__ in, u8* __restrict__ out, const u32* __restrict__ mbits,
                                                       const u32* __restrict__ vbits, int nw, i64 W, i64 H, i64 D, i64 Dp, int c0, int c2, int ntz) {
    __shared__ u32 tile[UZ * UPITCH];
    const int tid = threadIdx.x;
    const int t = (int)blockIdx.x, g = (int)blockIdx.y;
    const i64 x0 = (i64)(t / ntz) * UX, z0 = (i64)(t % ntz) * UZ;
    const u32* src = in + (i64)g * W * Dp;
    // (loads without a branch around each: a row outside the volume re-reads row 0 and is masked; only workgroups at the volume's z-edge
    // take the guarded path)
    const bool inside = x0 + c2 >= 0 && x0 + c2 + UX <= Dp;         // wave-uniform
    u32x4 dq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, r = idx >> 3, cg = idx & 7;
        const i64 sx = (i64)c0 - (z0 + r), sz = x0 + c2 + 4 * cg;
        const bool rowok = sx >= 0 && sx < W;
        const u32* p = src + (rowok ? sx : 0) * Dp + sz;
        if (inside) dq[k] = *(const u32x4_a4*)p;
        else {
            u32 d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sz + e >= 0 && sz + e < Dp) d[e] = p[e];
            dq[k].x = d[0]; dq[k].y = d[1]; dq[k].z = d[2]; dq[k].w = d[3];
        }
        if (!rowok) dq[k] = (u32x4)(0u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, r = idx >> 3, cg = idx & 7;
        u32* dst = tile + r * UPITCH + 4 * cg;
        dst[0] = dq[k].x; dst[1] = dq[k].y; dst[2] = dq[k].z; dst[3] = dq[k].w;
    }
    __syncthreads();
    const int zp = tid & 7, xl = tid >> 3;
    const i64 x = x0 + xl, z = z0 + 16 * zp;
    if (x >= W || z >= D) return;
    const u32 vb = (vbits[x * nw + (z >> 5)] >> (z & 31)) & 0xffffu;        // SciPy's bounds verdict of output cells (x, z .. z + 15)
    const u32 m = mbits[(i64)g * W + x];
    const int np = (int)(H - 32 * (i64)g < 32 ? H - 32 * (i64)g : 32);
    u32 v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32 a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = ((vb >> (4 * j + c)) & 1u) ? tile[(16 * zp + 4 * j + c) * UPITCH + xl] & m : 0u;
        tr4x4(a[0], a[1], a[2], a[3], v[j]);
    }
    u8* base = out + ((i64)x * H + 32 * (i64)g) * D + z;
    const bool whole = z + 16 <= D;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q >= np) continue;
        u32x4 r;
        r.x = (v[0][q >> 3] >> (q & 7)) & 0x01010101u; r.y = (v[1][q >> 3] >> (q & 7)) & 0x01010101u;
        r.z = (v[2][q >> 3] >> (q & 7)) & 0x01010101u; r.w = (v[3][q >> 3] >> (q & 7)) & 0x01010101u;
        u8* dp = base + (i64)q * D;
        if (whole) *(u32x4_a1*)dp = r;
        else {
            const u32 t4[4] = {r.x, r.y, r.z, r.w};
            for (int bb = 0; bb < 16 && z + bb < D; ++bb) dp[bb] = (u8)(t4[bb >> 2] >> (8 * (bb & 3)));
        }
    }
}

// the all-ones grid of global_carve (reference :279) after its 0-degree carve: every voxel of column (x, y) is mask[x, y], so the
// sliced volume is the row's mask word repeated along z (pad columns zero) -- written, never sliced from bytes
__global__ __launch_bounds__(256) void k_s32_fill(u32* __restrict__ out, const u32* __restrict__ mbits, i64 D, i64 Dp, pb3d_magic mzq, u32 total) {
    const u32 idx = blockIdx.x * 256u + threadIdx.x;                 // one thread = 4 dwords
    if (idx >= total) return;
    const u32 row = pb3d_div(idx, mzq), zq = idx - row * mzq.d;
    const u32 m = mbits[row];
    const i64 z = 4 * (i64)zq;
    u32x4 v;
    v.x = z < D ? m : 0u; v.y = z + 1 < D ? m : 0u; v.z = z + 2 < D ? m : 0u; v.w = z + 3 < D ? m : 0u;
    *(u32x4*)(out + (i64)row * Dp + z) = v;
}

// 4 occupancy bytes (0/1) -> the 12 colour bytes of 4 voxels (C0 = R|G<<8|B<<16|R<<24, C1 = G|B<<8|R<<16|G<<24, C2 = B|R<<8|G<<16|B<<24)
__device__ __forceinline__ void rgb4(u32 occ01, u32 C0, u32 C1, u32 C2, u32* o) {
    const u32 e = occ01 * 0xffu;                                                              // bytes 0x00 / 0xff (no carries)
    o[0] = pperm(e, e, 0x01000000u) & C0; o[1] = pperm(e, e, 0x02020101u) & C1; o[2] = pperm(e, e, 0x03030302u) & C2;
}

// S32 -> (W,H,D,3) colours: apply_colored_mask_to_voxel_grid (reference :128-136) folded into the un-slicing -- voxel (x,y,z) gets
// rgb_hw3[y, x] where its bit is set.  Same thread shape as k_s32_unslice: a thread owns 48 contiguous bytes per plane.  Stored
// directly that is 16 bytes every 48 per instruction (each line touched by three instructions: 0.75 ms for the 3.2 GB of 1024^3);
// the lanes' 48 bytes go through a wave-private LDS window and leave in ADDRESS order -- chunk c = 64 k + lane of the wave's 3 KB
// belongs to lane c / 3 -- so every store instruction writes whole lines (the trick of k_color_apply16 / k_rot90<RGBOUT>).
__global__ __launch_bounds__(256) void k_s32_unslice_rgb(const u32* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ rgb_hw3, i64 W, i64 H,
                                                         i64 D, i64 Dp, pb3d_magic mzb, pb3d_magic mw, u32 total) {
    __shared__ u32x4 win[4][192];
    const u32 idx = blockIdx.x * 256u + threadIdx.x;
    const bool live = idx < total;                       // (every lane takes part in the exchange)
    const u32 row = live ? pb3d_div(idx, mzb) : 0u, zb = live ? idx - row * mzb.d : 0u;
    const u32 g = pb3d_div(row, mw), x = row - g * mw.d;
    const i64 z = 16 * (i64)zb;
    const int np = (int)(H - 32 * (i64)g < 32 ? H - 32 * (i64)g : 32);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 v[4][4];
    if (live) {
        const u32* ip = in + (i64)row * Dp + z;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const u32x4 a = *(const u32x4*)(ip + 4 * j); tr4x4(a.x, a.y, a.z, a.w, v[j]); }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0u;
    }
    const i64 vbase = (((i64)x * H + 32 * (i64)g) * D + z) * 3;
    const u8* px = rgb_hw3 + ((32 * (i64)g) * W + x) * 3;
    // a thread whose 16 voxels run past the row's end (D % 16 != 0) stores byte-wise and stays out of the exchange; so does a dead lane
    const bool whole = live && z + 16 <= D;
    // planes this WAVE walks: its lanes may sit in two plane groups (rows of different g), one of them the short last group -- walk the
    // longest, predicate per lane.  The plane loop is unrolled (static indices into v[][]); planes past npmax are skipped wave-uniformly.
    int npmax = live ? np : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) npmax = max(npmax, __shfl_xor(npmax, o));
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q >= npmax) continue;
        const bool mine = whole && q < np;
        u32 o[12];
        if (live && q < np) {
            const u8* c = px + (i64)q * W * 3;
            const u32 R = c[0], G = c[1], B = c[2];
            const u32 C0 = R | (G << 8) | (B << 16) | (R << 24), C1 = G | (B << 8) | (R << 16) | (G << 24), C2 = B | (R << 8) | (G << 16) | (B << 24);
#pragma unroll
            for (int j = 0; j < 4; ++j) rgb4((v[j][q >> 3] >> (q & 7)) & 0x01010101u, C0, C1, C2, o + 3 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) o[j] = 0u;
        }
        const i64 mybase = mine ? vbase + (i64)q * D * 3 : -1;
        if (live && q < np && !whole) {
            u8* dp = out + vbase + (i64)q * D * 3;
            for (int bb = 0; bb < 48 && z * 3 + bb < D * 3; ++bb) dp[bb] = (u8)(o[bb >> 2] >> (8 * (bb & 3)));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { u32x4 t; t.x = o[4 * k]; t.y = o[4 * k + 1]; t.z = o[4 * k + 2]; t.w = o[4 * k + 3]; win[wv][3 * lane + k] = t; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = 64 * k + lane, L = c / 3, part = c - 3 * L;
            const u32x4 t = win[wv][c];
            const i64 lb = __shfl((long long)mybase, L);
            if (lb >= 0) *(u32x4_a1*)(out + lb + 16 * part) = t;
        }
    }
}

// ... and for global_carve (reference :289-292): the last 90-degree step writes the COLOURS.  Same tile as k_s32_unslice90; a lane's 16
// voxels are 48 bytes per plane, the eight lanes of an x-row hold 384 contiguous bytes: they pass through a wave-private LDS window and
// leave in address order (lane zp stores the 16-byte chunks zp, 8 + zp, 16 + zp of its row), so a store instruction writes 8 rows x
// 128 bytes.  Rows that end inside the tile (D % 128 != 0) are stored byte-wise.
__global__ __launch_bounds__(256) void k_s32_unslice90_rgb(const u32* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ rgb_hw3,
                                                           const u32* __restrict__ mbits, const u32* __restrict__ vbits, int nw, i64 W, i64 H, i64 D, i64 Dp,
                                                           int c0, int c2, int ntz) {
    __shared__ u32 tile[UZ * UPITCH];
    __shared__ u32x4 win[4][192];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t = (int)blockIdx.x, g = (int)blockIdx.y;
    const i64 x0 = (i64)(t / ntz) * UX, z0 = (i64)(t % ntz) * UZ;
    const u32* src = in + (i64)g * W * Dp;
    const bool inside = x0 + c2 >= 0 && x0 + c2 + UX <= Dp;         // wave-uniform
    u32x4 dq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, r = idx >> 3, cg = idx & 7;
        const i64 sx = (i64)c0 - (z0 + r), sz = x0 + c2 + 4 * cg;
        const bool rowok = sx >= 0 && sx < W;
        const u32* p = src + (rowok ? sx : 0) * Dp + sz;
        if (inside) dq[k] = *(const u32x4_a4*)p;
        else {
            u32 d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sz + e >= 0 && sz + e < Dp) d[e] = p[e];
            dq[k].x = d[0]; dq[k].y = d[1]; dq[k].z = d[2]; dq[k].w = d[3];
        }
        if (!rowok) dq[k] = (u32x4)(0u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, r = idx >> 3, cg = idx & 7;
        u32* dst = tile + r * UPITCH + 4 * cg;
        dst[0] = dq[k].x; dst[1] = dq[k].y; dst[2] = dq[k].z; dst[3] = dq[k].w;
    }
    __syncthreads();
    const int zp = tid & 7, xl = tid >> 3;
    const i64 x = x0 + xl, z = z0 + 16 * zp;
    const bool rowlive = x < W;                                      // (the eight lanes of a row agree)
    const bool live = rowlive && z < D;
    const u32 vb = live ? (vbits[x * nw + (z >> 5)] >> (z & 31)) & 0xffffu : 0u;
    const u32 m = rowlive ? mbits[(i64)g * W + x] : 0u;
    const int np = (int)(H - 32 * (i64)g < 32 ? H - 32 * (i64)g : 32);
    u32 v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32 a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = ((vb >> (4 * j + c)) & 1u) ? tile[(16 * zp + 4 * j + c) * UPITCH + xl] & m : 0u;
        tr4x4(a[0], a[1], a[2], a[3], v[j]);
    }
    const bool fullrow = z0 + UZ <= D;                               // block-uniform: every lane of a live row holds 16 voxels
    const i64 rowbase = (((i64)(rowlive ? x : 0) * H + 32 * (i64)g) * D + z0) * 3;
    const u8* px = rgb_hw3 + ((32 * (i64)g) * W + (rowlive ? x : 0)) * 3;
    const int grp = (lane & ~7) * 3;                                  // this row's 24 chunks in the wave's window
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q >= np) continue;
        u32 o[12];
        {
            const u8* c = px + (i64)q * W * 3;
            const u32 R = c[0], G = c[1], B = c[2];
            const u32 C0 = R | (G << 8) | (B << 16) | (R << 24), C1 = G | (B << 8) | (R << 16) | (G << 24), C2 = B | (R << 8) | (G << 16) | (B << 24);
#pragma unroll
            for (int j = 0; j < 4; ++j) rgb4((v[j][q >> 3] >> (q & 7)) & 0x01010101u, C0, C1, C2, o + 3 * j);
        }
        if (!fullrow) {
            if (live) {
                u8* dp = out + rowbase + (i64)q * D * 3 + 48 * zp;
                for (int bb = 0; bb < 48 && 3 * z + bb < 3 * D; ++bb) dp[bb] = (u8)(o[bb >> 2] >> (8 * (bb & 3)));
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { u32x4 t4; t4.x = o[4 * k]; t4.y = o[4 * k + 1]; t4.z = o[4 * k + 2]; t4.w = o[4 * k + 3]; win[wv][3 * lane + k] = t4; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int cidx = 8 * k + zp;
            const u32x4 t4 = win[wv][grp + cidx];
            if (rowlive) *(u32x4_a1*)(out + rowbase + (i64)q * D * 3 + 16 * cidx) = t4;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tile programs.  One workgroup per (step, tile): every thread evaluates SciPy's coordinates, weights and result table of its 16
// cells ONCE (f64, no contraction), the workgroup finds the bounding rows / per-row column extents of the taps (a rotated tile is
// a diamond inside its bounding box), lays the rows out in LDS and lists the 16-byte units to stage.
// Thread tid owns cells (x0 + (tid >> 4) + 16 i, z0 + 4 (tid & 15) + c), i, c < 4: a wave's 64 lanes cover 4 rows x 64 z, so each of
// its stores writes four whole 256-byte row segments.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(STHREADS) void k_s32_prep(StepParams sp, i64 W, i64 D, i64 Dp, int ntz, STile* __restrict__ tiles) {
    __shared__ int bb[4];
    __shared__ int rmin[SROWS], rmax[SROWS];
    __shared__ int ustart[SROWS + 1];
    __shared__ int rc0[SROWS];
    const int tid = threadIdx.x;
    const RotParams p = sp.p[blockIdx.y];
    const int t = (int)blockIdx.x;
    STile* o = tiles + (i64)blockIdx.y * gridDim.x + t;
    const i64 x0 = (i64)(t / ntz) * ST, z0 = (i64)(t % ntz) * ST;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    if (tid < SROWS) { rmin[tid] = 0x7fffffff; rmax[tid] = -1; }
    __syncthreads();
    const int tz = tid & 15, jr = tid >> 4;
    int s0[16], s2[16];
    u32 lut[16];                    // bits 0..15 table, 16: x tap 1 used, 17: z tap 1 used
    int mn0 = 0x7fffffff, mx0 = -1, mn2 = 0x7fffffff, mx2 = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * i + c;
            const i64 x = x0 + jr + 16 * i, z = z0 + 4 * tz + c;
            s0[k] = -1; s2[k] = 0; lut[k] = 0;
            if (x < W && z < D) {
                const Cell cl = make_cell(p, x, z, W, D);
                if (cl.s0 >= 0) {
                    s0[k] = cl.s0; s2[k] = cl.s2;
                    lut[k] = lut_of(cl) | (cl.wx1 != 0.0 ? 1u << 16 : 0u) | (cl.wz1 != 0.0 ? 1u << 17 : 0u);
                    const int e0 = cl.s0 + (int)((lut[k] >> 16) & 1u), e2 = cl.s2 + (int)((lut[k] >> 17) & 1u);
                    mn0 = min(mn0, cl.s0); mx0 = max(mx0, e0); mn2 = min(mn2, cl.s2); mx2 = max(mx2, e2);
                }
            }
        }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        mn0 = min(mn0, __shfl_xor(mn0, s)); mx0 = max(mx0, __shfl_xor(mx0, s));
        mn2 = min(mn2, __shfl_xor(mn2, s)); mx2 = max(mx2, __shfl_xor(mx2, s));
    }
    if ((tid & 63) == 0 && mx0 >= 0) { atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); atomicMin(&bb[2], mn2); atomicMax(&bb[3], mx2); }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1], bz0 = bb[2] & ~3, bz1 = bb[3];      // columns start on a 16-byte boundary
    const bool any_valid = bx1 >= 0;
    const int nrows = any_valid ? bx1 - bx0 + 1 : 0;
    // + 1: the second tap of a cell is read from LDS even when its weight is 0 (the table ignores it) -- it must stay inside the array
    bool fits = !any_valid || (nrows + 1 <= SROWS && bz1 - bz0 + 2 <= SPITCH);
    if (fits && any_valid) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (s0[k] < 0) continue;
            const int r = s0[k] - bx0, e2 = s2[k] + (int)((lut[k] >> 17) & 1u);
            atomicMin(&rmin[r], s2[k]); atomicMax(&rmax[r], e2);
            if ((lut[k] >> 16) & 1u) { atomicMin(&rmin[r + 1], s2[k]); atomicMax(&rmax[r + 1], e2); }
        }
    }
    __syncthreads();
    if (tid < 64) {                                    // wave 0: exclusive scan of the per-row unit counts (rows tid and tid + 64)
        int n[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tid + 64 * h;
            n[h] = 0;
            if (r < SROWS) {
                rc0[r] = 0;
                if (r < nrows && fits && rmax[r] >= 0) {
                    rc0[r] = (rmin[r] - bz0) >> 2;
                    n[h] = ((rmax[r] - bz0) >> 2) - ((rmin[r] - bz0) >> 2) + 1;
                }
            }
        }
        int inc0 = n[0], inc1 = n[1];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int a = __shfl_up(inc0, s), b2 = __shfl_up(inc1, s);
            if (tid >= s) { inc0 += a; inc1 += b2; }
        }
        const int tot0 = __shfl(inc0, 63);
        ustart[tid] = inc0 - n[0];
        if (tid + 64 <= SROWS) ustart[tid + 64] = tot0 + inc1 - n[1];
    }
    __syncthreads();
    int nunits = (any_valid && fits) ? ustart[SROWS] : 0;
    if (nunits > SMAXU * STHREADS) { fits = false; nunits = 0; }
    if (fits && tid < nrows) {
        const int u0 = ustart[tid], u1 = ustart[tid + 1];
        for (int u = u0; u < u1; ++u) {
            const int cu = rc0[tid] + (u - u0);
            o->ugoff[u] = (u32)(((i64)bx0 + tid) * Dp + bz0 + 4 * cu);
            o->uloff[u] = (unsigned short)(tid * SPITCH + 4 * cu);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        u32 w = 0;
        if (fits && s0[k] >= 0) w = (lut[k] << 16) | (u32)((s0[k] - bx0) * SPITCH + (s2[k] - bz0));
        o->cellw[k * STHREADS + tid] = w;
    }
    if (tid == 0) { o->bx0 = bx0; o->bz0 = bz0; o->nrows = nrows; o->nunits = nunits; o->fits = fits ? 1 : 0; o->pad0 = o->pad1 = o->pad2 = 0; }
}

// One rotation step S32 -> S32 with the step's mask (reference :116-124 for one angle).  Workgroup = (tile, chunk of plane groups);
// all tiles of a chunk run on ONE XCD (blockIdx % 8), so the lines neighbouring footprints share meet in one L2.
// Per plane group: stage the footprint (<= SMAXU 16-byte loads per thread, plain dword copies -- nothing is converted), evaluate
// 16 cells x 32 planes (four LDS dwords + a 15-select multiplexer tree per cell), store 4 x 16 bytes.
__global__ __launch_bounds__(STHREADS, 4) void k_s32_step(const u32* __restrict__ in, u32* __restrict__ out, const u32* __restrict__ mbits,
                                                           const STile* __restrict__ tiles, i64 W, i64 Dp, int G, int ntz, int ntiles, int gpw,
                                                           int nchunks) {
    __shared__ u32 tile[SROWS * SPITCH];
    const int tid = threadIdx.x;
    const int slot = (int)(blockIdx.x >> 3);
    const int t = slot % ntiles;
    const int chunk = (slot / ntiles) * 8 + (int)(blockIdx.x & 7u);
    if (chunk >= nchunks) return;                      // whole workgroup, before any barrier
    const STile* ti = tiles + t;
    const i64 x0 = (i64)(t / ntz) * ST, z0 = (i64)(t % ntz) * ST;
    const int g_beg = chunk * gpw, g_end = g_beg + gpw < G ? g_beg + gpw : G;
    const int nunits = ti->nunits;
    u32 ug[SMAXU], ul[SMAXU];
#pragma unroll
    for (int j = 0; j < SMAXU; ++j) {
        const int u = tid + STHREADS * j;
        ug[j] = 0xffffffffu; ul[j] = 0;
        if (u < nunits) { ug[j] = ti->ugoff[u]; ul[j] = ti->uloff[u]; }
    }
    u32 cw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cw[i] = ti->cellw[i * STHREADS + tid];
    const int tz = tid & 15, jr = tid >> 4;
    const i64 zc = z0 + 4 * tz;
    const bool zok = zc < Dp;
    // the footprint of plane group g + 1 is in flight (in registers) while group g is evaluated: written into LDS one group at a time, the
    // load -> barrier -> evaluate -> store sequence left the waves waiting 68 % of their cycles (profiles/r03_sliced_chain_sq_counters.txt)
    u32x4 v[SMAXU];
    auto load_group = [&](int g) {
        const u32* src = in + (i64)g * W * Dp;
#pragma unroll
        for (int j = 0; j < SMAXU; ++j)
            if (ug[j] != 0xffffffffu) v[j] = *(const u32x4*)(src + ug[j]);
    };
    load_group(g_beg);
    for (int g = g_beg; g < g_end; ++g) {
#pragma unroll
        for (int j = 0; j < SMAXU; ++j)
            if (ug[j] != 0xffffffffu) { u32* d = tile + ul[j]; d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w; }
        __syncthreads();
        if (g + 1 < g_end) load_group(g + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32 R[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32 w = cw[4 * i + c];
                const u32* tp = tile + (w & 0xffffu);
                R[c] = lut_apply32(w >> 16, tp[0], tp[1], tp[SPITCH], tp[SPITCH + 1]);
            }
            const i64 x = x0 + jr + 16 * i;
            if (x < W && zok) {
                const u32 m = mbits[(i64)g * W + x];
                u32x4 r; r.x = R[0] & m; r.y = R[1] & m; r.z = R[2] & m; r.w = R[3] & m;
                *(u32x4*)(out + ((i64)g * W + x) * Dp + zc) = r;
            }
        }
        __syncthreads();
    }
}

// every step of the chain is a rotation whose 64-tile footprint fits the LDS box by geometry (no device-side fallback is needed)
static bool step_fits(const double M[9]) {
    const double ext0 = fabs(M[0]) + fabs(M[2]), ext2 = fabs(M[6]) + fabs(M[8]);
    return ext0 <= 1.4143 && ext2 <= 1.4143;
}

}  // namespace

// process_voxel_grid(occ, mask, angle_interval) (reference utils/voxel_carving_utils.py:104-126) through the bit-sliced chain.
// *took = 0: the chain does not apply (fewer than two rotation steps, shape limits, tuning, or -- unless known_binary -- data that
// is not 0/1): nothing was written and the caller runs the byte chain.  known_binary = 0 costs one host wait for the slice kernel.
// d_occ == NULL: the source is the all-ones grid of global_carve (nothing is read).  d_rgb_hw3 != NULL: d_out is the (W,H,D,3)
// colour volume of global_carve (reference :289), written by the un-slicing pass.
// Mx / offx != NULL: ONE rotation step with this matrix and offset (pb3d_rotate_carve_dev), no 0-degree carve in front of it.
static int s32_chain(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out, const u8* d_rgb_hw3,
                     int known_binary, int* took, const double* Mx = nullptr, const double* offx = nullptr) {
    *took = 0;
    const int nrot = Mx ? 1 : 90 / angle_interval;
    // (tune sliced = 1: the byte chain with the arithmetic kernel -- the pinned reference of the parity tests)
    if (ctx->tune_sliced == 1 || nrot < 1) return PB3D_OK;
    if (!Mx && nrot == 1 && angle_interval == 90) {
        // a single 90-degree step: the permutation kernels of csrc/rotate_tiled.hip move it at 2 B/voxel (when W + D is even)
        const i64 sh[3] = {W, H, D};
        double M[9], off[3];
        PB3D_TRY(pb3d_rotinv(90, M));
        PB3D_TRY(pb3d_offset(M, sh, off));
        if (pb3d_is_perm_step(M, off, W, D)) return PB3D_OK;
    }
    const i64 Dp = (D + 15) & ~(i64)15;
    const int G = (int)((H + 31) / 32);
    const i64 nzb = Dp / 16;
    if (W >= 32768 || D >= 32768 || (i64)G * W * nzb >= (1ll << 31) || W * Dp >= (1ll << 31)) return PB3D_OK;
    const i64 shape[3] = {W, H, D};
    StepParams sp;
    const int ntz = (int)((D + ST - 1) / ST), ntx = (int)((W + ST - 1) / ST), ntiles = ntz * ntx;
    void *flagp, *mb;
    PB3D_TRY(pb3d_scratch(ctx, 34, 64, &flagp));
    PB3D_TRY(pb3d_scratch(ctx, 35, (size_t)G * W * sizeof(u32), &mb));
    int* flag = (int*)flagp;
    const size_t sbytes = (size_t)G * W * Dp * sizeof(u32);
    void *A = nullptr, *B = nullptr;
    PB3D_TRY(pb3d_dev_alloc(ctx, sbytes, &A));
    int rc = pb3d_dev_alloc(ctx, sbytes, &B);
    if (rc != PB3D_OK) { (void)pb3d_dev_free(ctx, A); return rc; }
    auto body = [&]() -> int {
        if (d_mask_wh) {
            hipLaunchKernelGGL(k_s32_maskbits, dim3((unsigned)(((i64)G * W + 255) / 256)), dim3(256), 0, ctx->stream, d_mask_wh, W, H, G, (u32*)mb, flag);
        } else {        // no mask: every plane bit passes (bits of planes past H stay zero in the volume: the slice pass never sets them)
            PB3D_HIP(hipMemsetAsync(mb, 0xff, (size_t)G * W * sizeof(u32), ctx->stream));
            PB3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        }
        const u32 total = (u32)((i64)G * W * nzb);
        const pb3d_magic mzb = pb3d_make_magic((u32)nzb), mw = pb3d_make_magic((u32)W), mg = pb3d_make_magic((u32)G);
        const int gfast = ctx->tune_s32_order == 1;       // (measured: the order of the wavefronts makes no difference, 0.61 ms either way at 1024^3 / 45)
        if (d_occ) {
            const bool nt = W * H * D >= (i64)256 << 20;
#define PB3D_SLICE(WH, NTL) hipLaunchKernelGGL((k_s32_slice<WH, NTL>), dim3((total + 255u) / 256u), dim3(256), 0, ctx->stream, d_occ, (u32*)A, Mx ? (const u32*)nullptr : (const u32*)mb, W, H, D, Dp, mzb, mw, total, flag, mg, gfast)
            if (D % 16 == 0) { if (nt) PB3D_SLICE(true, true); else PB3D_SLICE(true, false); }
            else { if (nt) PB3D_SLICE(false, true); else PB3D_SLICE(false, false); }
#undef PB3D_SLICE
        } else {
            const u32 tq = (u32)((i64)G * W * (Dp / 4));
            hipLaunchKernelGGL(k_s32_fill, dim3((tq + 255u) / 256u), dim3(256), 0, ctx->stream, (u32*)A, (const u32*)mb, D, Dp, pb3d_make_magic((u32)(Dp / 4)), tq);
        }
        PB3D_CHECK_LAUNCH();
        int* hflag = (int*)((char*)ctx->pinned + ctx->pinned_bytes - 64);
        if (!known_binary) {
            PB3D_HIP(hipMemcpyAsync(hflag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            PB3D_HIP(hipEventRecord(ctx->s32_ev, ctx->stream));
        }
        int gpw = ctx->tune_s32_gpw > 0 ? ctx->tune_s32_gpw : 1;
        if (ctx->tune_s32_gpw <= 0)
            while (gpw < G && (i64)ntiles * ((G + gpw - 1) / gpw) > (i64)ctx->cus * 16) ++gpw;
        const int nchunks = (G + gpw - 1) / gpw;
        const i64 nblk = 8ll * ntiles * ((nchunks + 7) / 8);
        PB3D_REQUIRE(nblk < (1ll << 31), "pb3d_process_grid: grid too large");
        u32 *src = (u32*)A, *dst = (u32*)B;
        bool checked = known_binary != 0;
        // the last step un-slices itself when it is the 90-degree one and a permutation of the cells (W + D even: integer offsets)
        int nrot_tab = nrot;
        u32* vbits = nullptr;
        int vnw = 0, pc0 = 0, pc2 = 0;
        if (!Mx && nrot * angle_interval == 90 && ctx->tune_s32_fuse_last != 1 && H <= 65535ll * 32) {
            double M[9], off[3];
            PB3D_TRY(pb3d_rotinv(90, M));
            PB3D_TRY(pb3d_offset(M, shape, off));
            bool rot90 = false;
            if (pb3d_is_perm_step(M, off, W, D)) {
                PB3D_TRY(pb3d_perm_valid_table(ctx, M, off, W, D, &vbits, &vnw, &pc0, &pc2, &rot90));
                if (rot90) nrot_tab = nrot - 1; else vbits = nullptr;
            }
        }
        for (int s0 = 0; s0 < nrot_tab; s0 += SMAXSTEPS) {
            const int ns = nrot_tab - s0 < SMAXSTEPS ? nrot_tab - s0 : SMAXSTEPS;
            memset(&sp, 0, sizeof(sp));
            for (int k = 0; k < ns; ++k) {
                double M[9], off[3];
                if (Mx) { memcpy(M, Mx, sizeof(M)); memcpy(off, offx, sizeof(off)); }
                else {
                    PB3D_TRY(pb3d_rotinv((s0 + k + 1) * angle_interval, M));
                    PB3D_TRY(pb3d_offset(M, shape, off));
                }
                if (!step_fits(M)) return PB3D_EUNSUPPORTED;            // (every Rinv is a rotation; a caller's own matrix may not be)
                sp.p[k] = RotParams{M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
            }
            // the tile programs of this run of steps: where an earlier call left them, or built now
            void* tp;
            PB3D_TRY(pb3d_scratch(ctx, 32, (size_t)ns * ntiles * sizeof(STile), &tp));
            pb3d_ctx::S32Cache& sc = ctx->s32_cache;
            const bool hit = ctx->tune_no_table_cache != 1 && sc.valid && sc.gen == ctx->scratch_gen && sc.W == W && sc.D == D && sc.ns == ns &&
                             memcmp(sc.p, sp.p, sizeof(RotParams) * (size_t)ns) == 0;
            if (!hit) {
                sc.valid = false;
                hipLaunchKernelGGL(k_s32_prep, dim3((unsigned)ntiles, (unsigned)ns), dim3(STHREADS), 0, ctx->stream, sp, W, D, Dp, ntz, (STile*)tp);
                PB3D_CHECK_LAUNCH();
                static_assert(sizeof(sc.p) == sizeof(sp.p), "S32Cache holds one StepParams");
                memcpy(sc.p, sp.p, sizeof(sp.p));
                sc.W = W; sc.D = D; sc.ns = ns; sc.gen = ctx->scratch_gen; sc.valid = nrot_tab <= SMAXSTEPS;
            }
            for (int k = 0; k < ns; ++k) {
                hipLaunchKernelGGL(k_s32_step, dim3((unsigned)nblk), dim3(STHREADS), 0, ctx->stream, (const u32*)src, dst, (const u32*)mb,
                                   (const STile*)tp + (i64)k * ntiles, W, Dp, G, ntz, ntiles, gpw, nchunks);
                u32* t2 = src; src = dst; dst = t2;
            }
            PB3D_CHECK_LAUNCH();
        }
        if (vbits && d_rgb_hw3) {
            const int utz = (int)((D + UZ - 1) / UZ), utx = (int)((W + UX - 1) / UX);
            hipLaunchKernelGGL(k_s32_unslice90_rgb, dim3((unsigned)(utz * utx), (unsigned)G), dim3(256), 0, ctx->stream, (const u32*)src, d_out, d_rgb_hw3,
                               (const u32*)mb, (const u32*)vbits, vnw, W, H, D, Dp, pc0, pc2, utz);
        } else if (vbits) {
            const int utz = (int)((D + UZ - 1) / UZ), utx = (int)((W + UX - 1) / UX);
            hipLaunchKernelGGL(k_s32_unslice90, dim3((unsigned)(utz * utx), (unsigned)G), dim3(256), 0, ctx->stream, (const u32*)src, d_out, (const u32*)mb,
                               (const u32*)vbits, vnw, W, H, D, Dp, pc0, pc2, utz);
        } else if (d_rgb_hw3)
            hipLaunchKernelGGL(k_s32_unslice_rgb, dim3((total + 255u) / 256u), dim3(256), 0, ctx->stream, (const u32*)src, d_out, d_rgb_hw3, W, H, D, Dp, mzb, mw,
                               total);
        else
            hipLaunchKernelGGL(k_s32_unslice, dim3((total + 255u) / 256u), dim3(256), 0, ctx->stream, (const u32*)src, d_out, W, H, D, Dp, mzb, mw, total, mg, gfast);
        PB3D_CHECK_LAUNCH();
        // The slice kernel's verdict on the data (values other than 0 / 1 -> the caller's byte chain) is read AFTER the whole chain has been
        // queued (round 4; rounds 3 waited for it before the first step: a host round trip in the middle of every chain, with the device
        // idle).  The wait is for the flag's copy, queued right behind the slice kernel -- not for the chain; on such data the steps already
        // queued run on garbage into d_out, which the byte chain then overwrites (the input is never written).
        if (!checked) {
            PB3D_HIP(hipEventSynchronize(ctx->s32_ev));
            checked = true;
            if (*hflag != 0) return PB3D_EUNSUPPORTED;
        }
        return PB3D_OK;
    };
    rc = body();
    (void)pb3d_dev_free(ctx, A);          // (the blocks go back to the context's pool; the stream is in order)
    (void)pb3d_dev_free(ctx, B);
    if (rc == PB3D_EUNSUPPORTED) return PB3D_OK;          // not 0/1 data: *took stays 0
    if (rc == PB3D_OK) *took = 1;
    return rc;
}

int pb3d_process_grid_sliced(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out,
                             int known_binary, int* took) {
    return s32_chain(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, nullptr, known_binary, took);
}

// global_carve(binary, rgb, angle_interval) (reference utils/voxel_carving_utils.py:269-298) for chains of two and more rotation
// steps: mask bits -> S32 -> steps -> colours; the volume exists as bytes only in its final (W,H,D,3) form.
int pb3d_global_carve_sliced(pb3d_ctx* ctx, const u8* d_mask_wh, const u8* d_rgb_hw3, i64 W, i64 H, i64 D, int angle_interval, u8* d_out_rgb,
                             int* took) {
    return s32_chain(ctx, nullptr, W, H, D, d_mask_wh, angle_interval, d_out_rgb, d_rgb_hw3, 1, took);
}

// one rotation step with the caller's matrix (pb3d_rotate_carve_dev) on 0/1 data; *took = 0: not applicable (data, shape, matrix)
int pb3d_rotate_step_sliced(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const double M[9], const double off[3], const u8* d_mask_wh, u8* d_out,
                            int* took) {
    return s32_chain(ctx, d_occ, W, H, D, d_mask_wh, 90, d_out, nullptr, 0, took, M, off);
}
