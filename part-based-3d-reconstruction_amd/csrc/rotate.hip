// K2: rotate-about-Y (trilinear, uint8 rounding) + mask carve, and the process_voxel_grid loop.
//
// Bit-exact restatement of scipy.ndimage.affine_transform(order=1, mode="constant", cval=0) as the
// reference calls it (reference utils/voxel_carving_utils.py:116-123), for matrices whose row 1 is
// [+-0, 1, +-0] with off[1] == 0 (every Rinv(angle) the reference can produce).  Then cc1 == y
// exactly, the y-weights are {1, 0}, and per output voxel the arithmetic is
//     cc_h = (((0 + x*M[h][0]) + y*M[h][1]) + z*M[h][2]) + off[h]       h = 0, 2   (IEEE, NO fma)
//     outside [0, n_h - 1]  -> 0
//     t = cc - floor(cc);  w0 = 1 - t;  w1 = 1 - w0
//     acc = ((v00*wx0)*wz0 + (v01*wx0)*wz1) + (v10*wx1)*wz0 + (v11*wx1)*wz1    (left to right)
//     out = acc > 0 ? (uint8)min(acc + 0.5, 255) : 0
// All doubles are evaluated with __dmul_rn/__dadd_rn so that no contraction can occur; the file is
// also compiled with -ffp-contract=off.  The coordinates depend on (x,z) only, so a thread computes
// them once and re-uses them for every Y-plane of its chunk (the FP64 work is amortised over y).
#include "rot_common.h"

namespace {

// value of one output voxel from plane y of `in`; taps with an exactly-zero weight add +0.0 and are
// skipped (this also keeps the index in range: a tap beyond n-1 only ever occurs with weight 0).
// mask_src (optional): the input is read as carve(in, mask_src) -- a source column (n0, y) with mask_src[n0*H + y] == 0 reads as zeros
// (the 0-degree carve of process_voxel_grid folded into its first rotation step).
__device__ __forceinline__ u32 sample(const u8* __restrict__ in, const Cell& c, i64 y, i64 H, i64 D, const u8* __restrict__ mask_src = nullptr) {
    if (c.s0 < 0) return 0;
    const u8* r0 = in + ((i64)c.s0 * H + y) * D + c.s2;
    const u8* r1 = r0 + H * D;
    double acc = 0.0;
    const bool x1 = c.wx1 != 0.0, z1 = c.wz1 != 0.0;
    const double k0 = (!mask_src || mask_src[(i64)c.s0 * H + y]) ? 1.0 : 0.0;            // v * 1.0 == v, v * 0.0 == +0.0: exact
    const double k1 = (x1 && (!mask_src || mask_src[((i64)c.s0 + 1) * H + y])) ? 1.0 : 0.0;
    // weights are >= 0; a zero wx0/wz0 (never happens: w0 = 1 - t > 0) needs no special case
    acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r0[0], k0), c.wx0), c.wz0));
    if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r0[1], k0), c.wx0), c.wz1));
    if (x1) {
        acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r1[0], k1), c.wx1), c.wz0));
        if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r1[1], k1), c.wx1), c.wz1));
    }
    if (!(acc > 0.0)) return 0;
    acc = __dadd_rn(acc, 0.5);
    if (acc > 255.0) acc = 255.0;
    return (u32)acc;  // truncation
}

// Tile: 4 x-rows (one per wavefront) by 256 z (4 consecutive z per lane), swept over TY planes.
template <bool PACK>
__global__ __launch_bounds__(256) void k_rotate_generic(const u8* __restrict__ in, u8* __restrict__ out,
                                                        const u8* __restrict__ mask_wh, RotParams p, i64 W, i64 H, i64 D,
                                                        int TY, const int* __restrict__ run_if, const u8* __restrict__ mask_src,
                                                        unsigned gx, unsigned gy, unsigned gz, int* __restrict__ clear_flag = nullptr) {
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;      // the flag word of a LATER step (ring of 16, see the launcher)
    if (run_if && *run_if == 0) return;   // second pass of a table-driven step: only when a value > 1 was seen
    // the (gx, gy, gz) block space is walked by however many workgroups were launched: the conditional second pass is launched
    // with a small grid, so that skipping it costs microseconds
    const int lane = threadIdx.x & 63;
    const u64 nblk = (u64)gx * gy * gz;
    for (u64 b = blockIdx.x; b < nblk; b += gridDim.x) {
        const unsigned bx = (unsigned)(b % gx), by = (unsigned)((b / gx) % gy), bz = (unsigned)(b / ((u64)gx * gy));
        const i64 x = (i64)by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const i64 z0 = ((i64)bx * 64 + lane) * 4;
        if (x >= W || z0 >= D) continue;
        const i64 y_beg = (i64)bz * TY;
        const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
        Cell c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (z0 + q < D) c[q] = make_cell(p, x, z0 + q, W, D);
            else { c[q].s0 = -1; c[q].s2 = 0; c[q].wx0 = c[q].wx1 = c[q].wz0 = c[q].wz1 = 0.0; }
        }
        for (i64 y = y_beg; y < y_end; ++y) {
            const bool keep = mask_wh ? mask_wh[x * H + y] != 0 : true;  // wave-uniform
            u32 r[4] = {0, 0, 0, 0};
            if (keep) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = sample(in, c[q], y, H, D, mask_src);
            }
            u8* o = out + (x * H + y) * D + z0;
            if (PACK) {
                *(u32*)o = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (z0 + q < D) o[q] = (u8)r[q];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic-angle step for 0/1 data, LDS-tiled.  Output tile 64 x 64 (x,z) per workgroup, swept over
// TY planes.
//  * the f64 coordinates / weights of a cell depend on (x,z) only: they are evaluated ONCE per
//    workgroup and folded into a 16-bit lookup table per cell: bit b of the table is SciPy's result
//    for the tap pattern b = v00 | v01<<1 | v10<<2 | v11<<3 (partial sums formed in SciPy's tap order
//    from the exact products w_x*w_z; 1.0*w == w and a zero tap adds +0.0, so this IS the reference
//    arithmetic for 0/1 inputs);
//  * the rotated footprint (bounding box of the tile's sources incl. the second taps) is staged into
//    LDS with coalesced 16-byte row loads and every output becomes LDS reads + a table lookup -- pure
//    integer work in the plane loop (k_rotate_bits below does this for eight planes at once);
//  * if ANY staged byte is > 1 the workgroup raises *big_flag: the launcher then lets the arithmetic
//    kernel k_rotate_generic redo the step (it starts only when the flag is set), so 0..255 grids
//    stay exact without a host round trip.
// ------------------------------------------------------------------------------------------------
constexpr int LT = 64;              // tile edge

// ------------------------------------------------------------------------------------------------
// Bit-sliced evaluation: EIGHT Y-planes per pass.  The source position of a cell is the same
// in every plane, so the staged footprint holds, per source voxel, one byte whose bit p is the 0/1 value
// of plane y0 + p.  A cell's four taps are then four bytes, and its 16-entry table is applied to all
// eight planes at once as a 4-level multiplexer tree of bitwise selects (v_bfi_b32):
//     f = mux(t11, mux(t10, mux(t01, mux(t00, L15, L14), ...), ...), ...)      L_k = table bit k, broadcast
// -> 4 LDS byte reads + ~31 bit operations per cell per 8 planes, one barrier pair per 8 planes.
// Packing on the way in: word |= (plane_dword & 0x01010101) << p; unpacking on the way out:
// (R >> p) & 0x01010101 is the output dword of plane p for the thread's 4 consecutive z.
// Values > 1 anywhere raise *big_flag: the launcher lets the arithmetic kernel redo the step (see above).
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ u32 lut_apply8(u32 lut, u32 t00, u32 t01, u32 t10, u32 t11) {
    u32 L[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) L[k] = 0u - ((lut >> k) & 1u);
    u32 g[8], h[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = bsel(t00, L[2 * j + 1], L[2 * j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = bsel(t01, g[2 * j + 1], g[2 * j]);
    const u32 m0 = bsel(t10, h[1], h[0]), m1 = bsel(t10, h[3], h[2]);
    return bsel(t11, m1, m0) & 0xffu;
}

#ifndef BITS_WAVES
#define BITS_WAVES 4
#endif
constexpr int BPITCH = 128;                       // LDS row pitch of the bit-sliced tile (16-byte units)
constexpr int BROWS = 100;
constexpr int BMAXU = (BROWS * (BPITCH / 16) + 255) / 256;   // 16-byte staging units per thread

// AL: D % 16 == 0 and both volumes 16-byte aligned -> every staged piece and every output run is one aligned 16-byte
// access; the byte-wise edge paths exist only in the AL = false instantiation (they cost ~1600 SGPR spill moves per pass
// when compiled into the same loop).
template <bool AL>
__global__ __launch_bounds__(256, BITS_WAVES) void k_rotate_bits(const u8* __restrict__ in, u8* __restrict__ out,
                                                                 const u8* __restrict__ mask_wh, RotParams p, i64 W, i64 H, i64 D, int TY,
                                                                 int* __restrict__ big_flag) {
    __shared__ __attribute__((aligned(16))) u8 tile[BROWS * BPITCH];
    __shared__ int bb[4];
    const int tid = threadIdx.x;
    const i64 x0 = (i64)blockIdx.y * LT, z0 = (i64)blockIdx.x * LT;
    const i64 y_beg = (i64)blockIdx.z * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    __syncthreads();
    const int zl = (tid & 3) * 16, xl0 = tid >> 2;   // 16 cells: row xl0, z = zl + 0..15
    u32 src[16], lut[16];
    int mn0 = 0x7fffffff, mx0 = -1, mn2 = 0x7fffffff, mx2 = -1;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const i64 x = x0 + xl0, z = z0 + zl + c;
        src[c] = 0xffffffffu; lut[c] = 0;
        if (x < W && z < D) {
            const Cell cell = make_cell(p, x, z, W, D);
            if (cell.s0 >= 0) {
                src[c] = ((u32)cell.s0 << 16) | (u32)cell.s2;
                lut[c] = lut_of(cell);
                const int e0 = cell.s0 + (cell.wx1 != 0.0 ? 1 : 0), e2 = cell.s2 + (cell.wz1 != 0.0 ? 1 : 0);
                mn0 = cell.s0 < mn0 ? cell.s0 : mn0; mx0 = e0 > mx0 ? e0 : mx0;
                mn2 = cell.s2 < mn2 ? cell.s2 : mn2; mx2 = e2 > mx2 ? e2 : mx2;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (mx0 >= 0) { atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); atomicMin(&bb[2], mn2); atomicMax(&bb[3], mx2); }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1], bz0 = bb[2] & ~15, bz1 = bb[3];   // columns start on a 16-byte boundary
    const bool any_valid = bx1 >= 0;
    const int nrows = any_valid ? bx1 - bx0 + 1 : 0;
    const int nu = any_valid ? (bz1 - bz0) / 16 + 1 : 0;                  // 16-byte units per staged row
    const bool fits = nrows + 1 <= BROWS && nu * 16 + 4 <= BPITCH + 3 && nu * 16 <= BPITCH;
    if (any_valid && !fits) { if (tid == 0) atomicOr(big_flag, 1); }
    u32 cellw[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        u32 o = 0xffffu;
        if (src[c] != 0xffffffffu && fits) o = (u32)(((int)(src[c] >> 16) - bx0) * BPITCH + ((int)(src[c] & 0xffffu) - bz0));
        cellw[c] = (lut[c] << 16) | o;
    }
    const int nunits = (any_valid && fits) ? nrows * nu : 0;
    const u32 numagic = nu > 1 ? (u32)(((1ull << 32) + nu - 1) / nu) : 0;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32 hib = 0;
    for (i64 yg = y_beg; yg < y_end; yg += 8) {
        const int np = (int)(y_end - yg < 8 ? y_end - yg : 8);
        u32 mbits = 0;          // bit q: mask_wh[x, yg + q] of this thread's output row (issued early, consumed after the barrier)
        {
            const i64 x = x0 + xl0;
            if (x < W) {
#pragma unroll
                for (int q = 0; q < 8; ++q) mbits |= (u32)((q < np) && (!mask_wh || mask_wh[x * H + yg + q])) << q;
            }
        }
        // ---- stage the footprint of 8 planes, bit-sliced: one 16-byte piece of 8 planes per step (8 loads in flight)
#pragma unroll
        for (int j = 0; j < BMAXU; ++j) {
            const int i = tid + 256 * j;
            if (i >= nunits) continue;
            const int r = nu > 1 ? (int)__umulhi((u32)i, numagic) : i;
            const int cu = i - r * nu;
            const i64 col = (i64)bz0 + 16 * cu;
            const u8* sp = in + (((i64)bx0 + r) * H + yg) * D + col;
            u32x4 d[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                d[q] = (u32x4)(0u);
                if (q < np) {
                    const u8* s8 = sp + (i64)q * D;
                    if (AL) d[q] = *(const u32x4*)s8;
                    else {
                        u32 t[4] = {0, 0, 0, 0};
                        for (int b = 0; b < 16; ++b) if (col + b < D) t[b >> 2] |= (u32)s8[b] << (8 * (b & 3));
                        d[q].x = t[0]; d[q].y = t[1]; d[q].z = t[2]; d[q].w = t[3];
                    }
                }
            }
            u32x4 wv = (u32x4)(0u);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                wv.x |= (d[q].x & 0x01010101u) << q; wv.y |= (d[q].y & 0x01010101u) << q;
                wv.z |= (d[q].z & 0x01010101u) << q; wv.w |= (d[q].w & 0x01010101u) << q;
                hib |= d[q].x | d[q].y | d[q].z | d[q].w;
            }
            *(u32x4*)(tile + r * BPITCH + 16 * cu) = wv;
        }
        __syncthreads();
        // ---- evaluate 16 cells x 8 planes, write np planes of this thread's row (16 z = one 16-byte store per plane)
        {
            const i64 x = x0 + xl0;
            if (x < W && z0 + zl < D) {
                u32 R[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const u32 cw = cellw[c], o = cw & 0xffffu;
                    if (o == 0xffffu) continue;
                    R[c >> 2] |= lut_apply8(cw >> 16, tile[o], tile[o + 1], tile[o + BPITCH], tile[o + BPITCH + 1]) << (8 * (c & 3));
                }
                for (int q = 0; q < np; ++q) {
                    const i64 y = yg + q;
                    const u32 keep = ((mbits >> q) & 1u) ? 0x01010101u : 0u;
                    u32x4 r;
                    r.x = (R[0] >> q) & keep; r.y = (R[1] >> q) & keep; r.z = (R[2] >> q) & keep; r.w = (R[3] >> q) & keep;
                    u8* op = out + (x * H + y) * D + z0 + zl;
                    if (AL) *(u32x4*)op = r;
                    else {
                        const u32 t[4] = {r.x, r.y, r.z, r.w};
                        for (int b = 0; b < 16; ++b) if (z0 + zl + b < D) op[b] = (u8)(t[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
        }
        __syncthreads();
    }
    if (hib & 0xfefefefeu) atomicOr(big_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// 32-plane bit-sliced form (aligned volumes, the large-grid path).  k_rotate_bits spends its time in VALU work that
// is the same for every plane (16 table-bit broadcasts + a 15-select mux tree per cell and pass, of which only 8 of
// the 32 bit lanes are used) and in per-workgroup f64 set-up.  Here
//  * the cell table (source offset + 16-bit result table + second-tap flags) is computed ONCE per step by
//    k_rot_cells (W*D cells, 8 bytes each, L2 resident) and only read by the tile workgroups;
//  * the staged footprint holds one DWORD per source voxel, bit q = plane yg + q, so one mux tree serves 32 planes;
//  * bytes <-> bit planes are converted with the 0x01010101 gather of k_rotate_bits per group of 8 planes plus a 4x4
//    byte transpose (8 v_perm per 4 voxels) on the way in and on the way out.
// VALU work per voxel drops ~7x.  Phase ablation at 1024^3, 45 degrees (0.77 ms): footprint staging 0.43 ms, output stores 0.29 ms,
// table evaluation 0.01 ms, set-up 0.01 ms.  The loads are bound at LINE level: a rotated tile's rows are ~64-byte segments
// of 128-byte lines (FETCH_SIZE 3.3 GB for 1.07 GB of input with bounding-box staging); the per-row extents below and the
// XCD-contiguous tile order trim that, a larger tile (longer row segments) is the remaining lever.
// ------------------------------------------------------------------------------------------------
struct CellRec { u32 src, lut; };      // src = s0 << 16 | s2 (0xffffffff: outside); lut bits 0..15 table, 16: x tap 1 used, 17: z tap 1 used

__global__ __launch_bounds__(256) void k_rot_cells(RotParams p, i64 W, i64 D, CellRec* __restrict__ cells, u32* __restrict__ lutmap) {
    // lutmap (optional, 512 words, zeroed by the launcher): bit t is set when some cell's table bits 1..14 equal t -- a rotation
    // produces about a dozen distinct tables, which lets the packed kernel keep a 4-bit index per cell (k_rot8_pack).
    // Four cells per thread (1024 per block): the per-block flush of the bitmap is what this kernel's time consists of.
    __shared__ u32 seen[512];
    if (lutmap) { seen[threadIdx.x] = 0; seen[threadIdx.x + 256] = 0; __syncthreads(); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i64 i = ((i64)blockIdx.x * 4 + q) * 256 + threadIdx.x;
        u32 tbl = 0xffffffffu;
        if (i < W * D) {
            const i64 x = i / D, z = i - x * D;
            const Cell c = make_cell(p, x, z, W, D);
            CellRec r; r.src = 0xffffffffu; r.lut = 0;
            if (c.s0 >= 0) {
                r.src = ((u32)c.s0 << 16) | (u32)c.s2;
                r.lut = lut_of(c) | (c.wx1 != 0.0 ? 1u << 16 : 0u) | (c.wz1 != 0.0 ? 1u << 17 : 0u);
                tbl = (r.lut >> 1) & 0x3fffu;
            }
            cells[i] = r;
        }
        if (lutmap) {
            // a wave holds a handful of distinct tables: one LDS atomic per distinct value, not per lane
            u64 todo = __ballot(tbl != 0xffffffffu);
            while (todo) {
                const int lead = __builtin_ctzll(todo);
                const u32 tv = (u32)__shfl((int)tbl, lead);
                if ((int)(threadIdx.x & 63) == lead) atomicOr(&seen[tv >> 5], 1u << (tv & 31));
                todo &= ~__ballot(tbl == tv);
            }
        }
    }
    if (lutmap) {
        __syncthreads();
        for (int k = threadIdx.x; k < 512; k += 256)         // a dozen distinct tables in all: after the first blocks nothing is new
            if (seen[k] && (__hip_atomic_load(&lutmap[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & seen[k]) != seen[k]) atomicOr(&lutmap[k], seen[k]);
    }
}

// bit q = mask_src[base + q] != 0 for q < np (all ones without a source mask): which planes of one source row are kept by the
// 0-degree carve that process_voxel_grid folds into its first rotation step; masked planes are not even loaded
__device__ __forceinline__ u32 src_plane_bits(const u8* __restrict__ mask_src, i64 base, int np) {
    if (!mask_src) return 0xffffffffu;
    u32 bits = 0;
    for (int q = 0; q < np; ++q) bits |= (u32)(mask_src[base + q] != 0) << q;
    return bits;
}

constexpr int WPITCH = 128;                       // LDS row pitch in voxels (dwords)
constexpr int WROWS = 100;

// Workgroup -> (tile, plane block): neighbouring tiles stage overlapping footprints (the bounding box of a rotated tile is up
// to 2x its area), so they should share an L2.  Workgroups are dealt round-robin over the 8 XCDs; id % 8 therefore labels
// an XCD, and that XCD walks ONE contiguous eighth of the tile list (row-major strips) for plane block after plane block.
constexpr int WTHREADS = 256;                    // (512 threads x 8 cells, 4 waves/SIMD: spills, 0.95 ms vs 0.82 ms at 1024^3)
constexpr int WCELLS = LT * LT / WTHREADS;        // 16 cells per thread: row xl0 = tid / 4, z = 16 * (tid % 4) + 0..15
constexpr int WTPR = LT / WCELLS;                 // threads per tile row
constexpr int WMAXU = (WROWS * (WPITCH / 16) + WTHREADS - 1) / WTHREADS;

template <bool RAGGED>      // false: D % 16 == 0, no piece or run straddles a row end (byte-wise paths compiled out)
__global__ __launch_bounds__(WTHREADS, 2) void k_rotate_bits32(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_wh,
                                                              const CellRec* __restrict__ cells, i64 W, i64 H, i64 D, int TY, int ntz,
                                                              int ntiles, int* __restrict__ big_flag, const u8* __restrict__ mask_src) {
    __shared__ __attribute__((aligned(16))) u32 tile[WROWS * WPITCH];
    __shared__ int bb[4];
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x4 u32x4_a1 __attribute__((aligned(1)));     // any byte alignment (rows of odd-sized grids)
    typedef u32x2 u32x2_a1 __attribute__((aligned(1)));
    const int tid = threadIdx.x;
    const int chunk = (ntiles + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int t = (int)(blockIdx.x & 7u) * chunk + slot % chunk;
    if (t >= ntiles) return;                          // whole workgroup, before any barrier
    const i64 x0 = (i64)(t / ntz) * LT, z0 = (i64)(t % ntz) * LT;
    const i64 y_beg = (i64)(slot / chunk) * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    __syncthreads();
    const int zl = (tid % WTPR) * WCELLS, xl0 = tid / WTPR;
    const i64 x = x0 + xl0;
    const bool row_ok = x < W && z0 + zl < D;        // the run may be ragged (z0 + zl + WCELLS > D): cells past D are void
    u32 src[WCELLS], lut[WCELLS];
    int mn0 = 0x7fffffff, mx0 = -1, mn2 = 0x7fffffff, mx2 = -1;
#pragma unroll
    for (int c = 0; c < WCELLS; ++c) { src[c] = 0xffffffffu; lut[c] = 0; }
    if (row_ok) {
        typedef u32x4 u32x4_u __attribute__((aligned(8)));
        const u32x4_u* cp = (const u32x4_u*)(cells + x * D + z0 + zl);      // WCELLS records (the table is padded by WCELLS)
#pragma unroll
        for (int k = 0; k < WCELLS / 2; ++k) {
            const u32x4 v = cp[k];
            src[2 * k] = v.x; lut[2 * k] = v.y; src[2 * k + 1] = v.z; lut[2 * k + 1] = v.w;
        }
        if (RAGGED && z0 + zl + WCELLS > D) {
#pragma unroll
            for (int c = 0; c < WCELLS; ++c)
                if (z0 + zl + c >= D) { src[c] = 0xffffffffu; lut[c] = 0; }
        }
#pragma unroll
        for (int c = 0; c < WCELLS; ++c) {
            if (src[c] == 0xffffffffu) continue;
            const int s0 = (int)(src[c] >> 16), s2 = (int)(src[c] & 0xffffu);
            const int e0 = s0 + (int)((lut[c] >> 16) & 1u), e2 = s2 + (int)((lut[c] >> 17) & 1u);
            mn0 = s0 < mn0 ? s0 : mn0; mx0 = e0 > mx0 ? e0 : mx0;
            mn2 = s2 < mn2 ? s2 : mn2; mx2 = e2 > mx2 ? e2 : mx2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                 // one LDS atomic per wave instead of one per lane
        mn0 = min(mn0, __shfl_xor(mn0, o)); mx0 = max(mx0, __shfl_xor(mx0, o));
        mn2 = min(mn2, __shfl_xor(mn2, o)); mx2 = max(mx2, __shfl_xor(mx2, o));
    }
    if ((tid & 63) == 0 && mx0 >= 0) { atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); atomicMin(&bb[2], mn2); atomicMax(&bb[3], mx2); }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1], bz0 = bb[2] & ~15, bz1 = bb[3];   // columns start on a 16-byte boundary
    const bool any_valid = bx1 >= 0;
    const int nrows = any_valid ? bx1 - bx0 + 1 : 0;
    const int nu = any_valid ? (bz1 - bz0) / 16 + 1 : 0;                  // 16-voxel units per staged row
    const bool fits = nrows + 1 <= WROWS && nu * 16 <= WPITCH;
    if (any_valid && !fits) { if (tid == 0) atomicOr(big_flag, 1); }
    u32 cellw[WCELLS];                                                     // table << 16 | LDS dword offset (0xffff: outputs 0)
#pragma unroll
    for (int c = 0; c < WCELLS; ++c) {
        u32 o = 0xffffu;
        if (src[c] != 0xffffffffu && fits) o = (u32)(((int)(src[c] >> 16) - bx0) * WPITCH + ((int)(src[c] & 0xffffu) - bz0));
        cellw[c] = (lut[c] << 16) | o;
    }
    // Stage only what the tile reads: per footprint row the 16-voxel units between the leftmost and the rightmost tap of
    // that row (a rotated tile is a diamond inside its bounding box -- up to half of the box is never read).
    __shared__ int rmin[WROWS], rmax[WROWS];
    __shared__ unsigned short ustart[WROWS + 1];
    __shared__ u8 rc0[WROWS], urow[WROWS * (WPITCH / 16)];
    if (tid < WROWS) { rmin[tid] = 0x7fffffff; rmax[tid] = -1; }
    __syncthreads();
    if (fits) {
#pragma unroll
        for (int c = 0; c < WCELLS; ++c) {
            if (src[c] == 0xffffffffu) continue;
            const int r = (int)(src[c] >> 16) - bx0, s2 = (int)(src[c] & 0xffffu);
            const int e2 = s2 + (int)((lut[c] >> 17) & 1u);
            atomicMin(&rmin[r], s2); atomicMax(&rmax[r], e2);
            if ((lut[c] >> 16) & 1u) { atomicMin(&rmin[r + 1], s2); atomicMax(&rmax[r + 1], e2); }
        }
    }
    __syncthreads();
    if (tid < 64) {                                    // wave 0: exclusive scan of the per-row unit counts (rows tid and tid + 64)
        int n[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tid + 64 * h;
            n[h] = 0;
            if (r < nrows && fits && rmax[r] >= 0) {
                rc0[r] = (u8)((rmin[r] - bz0) >> 4);
                n[h] = ((rmax[r] - bz0) >> 4) - ((rmin[r] - bz0) >> 4) + 1;
            } else if (r < WROWS) rc0[r] = 0;
        }
        int inc0 = n[0], inc1 = n[1];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(inc0, o), b2 = __shfl_up(inc1, o);
            if (tid >= o) { inc0 += a; inc1 += b2; }
        }
        const int tot0 = __shfl(inc0, 63);
        ustart[tid] = (unsigned short)(inc0 - n[0]);
        if (tid + 64 <= WROWS) ustart[tid + 64] = (unsigned short)(tot0 + inc1 - n[1]);
        if (tid == 63) ustart[WROWS] = (unsigned short)(tot0 + inc1);
    }
    __syncthreads();
    const int nunits = (any_valid && fits) ? (int)ustart[WROWS] : 0;
    if (tid < nrows && fits) {
        const int u0 = ustart[tid], u1 = ustart[tid + 1];
        for (int u = u0; u < u1; ++u) urow[u] = (u8)tid;
    }
    __syncthreads();
    u32 hib = 0;
    for (i64 yg = y_beg; yg < y_end; yg += 32) {
        const int np = (int)(y_end - yg < 32 ? y_end - yg : 32);
        u32 mbits = 0;          // bit q: mask_wh[x, yg + q] (issued early, consumed after the barrier)
        if (row_ok) {
            if (!mask_wh) mbits = 0xffffffffu;
            else if ((((uintptr_t)mask_wh + (uintptr_t)(x * H + yg)) & 3u) == 0 && np == 32) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    u32 t4 = *(const u32*)(mask_wh + x * H + yg + 4 * k);
                    t4 |= t4 >> 4; t4 |= t4 >> 2; t4 |= t4 >> 1; t4 &= 0x01010101u;   // byte != 0 -> bit 0 of the byte
                    mbits |= ((t4 * 0x01020408u) >> 24) << (4 * k);                   // bits 0, 8, 16, 24 -> bits 0..3
                }
            } else {
                for (int q = 0; q < np; ++q) mbits |= (u32)(mask_wh[x * H + yg + q] != 0) << q;
            }
        }
        // ---- stage the footprint of 32 planes: 16 voxels x 32 planes per unit, 16 planes of 16-byte loads in flight per lane
#pragma unroll 1
        for (int j = 0; j < WMAXU; ++j) {
            const int i = tid + WTHREADS * j;
            if (i >= nunits) break;
            const int r = urow[i];
            const int cu = (int)rc0[r] + (i - (int)ustart[r]);
            // per-lane 32-bit offset + wave-uniform plane base (scalar registers): no per-plane vector address arithmetic
            const u32 voff = (u32)(((i64)bx0 + r) * H * D + (i64)bz0 + 16 * cu);
            const u32 msrc = src_plane_bits(mask_src, ((i64)bx0 + r) * H + yg, np);       // planes of this source row that survive the folded 0-degree carve
            u32x4 wg[4];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                u32x4 d[16];
                // the ragged unit at a row's end may be read whole while it stays inside the volume (the bytes past the row are
                // only ever zero-weight taps); byte-wise only at the very end of the buffer
                const bool whole = !RAGGED || (i64)bz0 + 16 * cu + 15 < D || (i64)voff + (yg + 16 * gg + 15) * D + 16 <= W * H * D;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const u8* pb = in + (yg + 16 * gg + q) * D;      // uniform
                    d[q] = (u32x4)(0u);
                    if (16 * gg + q < np && ((msrc >> (16 * gg + q)) & 1u)) {
                        if (whole) d[q] = *(const u32x4_a1*)(pb + voff);
                        else {
                            u32 t4[4] = {0, 0, 0, 0};
                            for (int b = 0; (i64)bz0 + 16 * cu + b < D; ++b) t4[b >> 2] |= (u32)pb[voff + b] << (8 * (b & 3));
                            d[q].x = t4[0]; d[q].y = t4[1]; d[q].z = t4[2]; d[q].w = t4[3];
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 wv = (u32x4)(0u);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const u32x4 dd = d[8 * h + q];
                        wv.x |= (dd.x & 0x01010101u) << q; wv.y |= (dd.y & 0x01010101u) << q;
                        wv.z |= (dd.z & 0x01010101u) << q; wv.w |= (dd.w & 0x01010101u) << q;
                        hib |= dd.x | dd.y | dd.z | dd.w;
                    }
                    wg[2 * gg + h] = wv;
                }
            }
            u32 v[4];
            u32x4 o4;
            u32* trow = tile + r * WPITCH + 16 * cu;
            tr4x4(wg[0].x, wg[1].x, wg[2].x, wg[3].x, v); o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *(u32x4*)(trow + 0) = o4;
            tr4x4(wg[0].y, wg[1].y, wg[2].y, wg[3].y, v); o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *(u32x4*)(trow + 4) = o4;
            tr4x4(wg[0].z, wg[1].z, wg[2].z, wg[3].z, v); o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *(u32x4*)(trow + 8) = o4;
            tr4x4(wg[0].w, wg[1].w, wg[2].w, wg[3].w, v); o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3]; *(u32x4*)(trow + 12) = o4;
        }
        __syncthreads();
        // ---- evaluate WCELLS cells x 32 planes; write np planes of this thread's WCELLS-byte run
        if (row_ok) {
            u32 G[WCELLS / 4][4];        // G[i][g]: byte c = planes 8g..8g+7 of cell 4i + c
#pragma unroll
            for (int i = 0; i < WCELLS / 4; ++i) {
                u32 R[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 cw = cellw[4 * i + c], o = cw & 0xffffu;
                    R[c] = 0;
                    if (o != 0xffffu) R[c] = lut_apply32(cw >> 16, tile[o], tile[o + 1], tile[o + WPITCH], tile[o + WPITCH + 1]);
                }
                tr4x4(R[0], R[1], R[2], R[3], G[i]);
            }
            const u32 ooff = (u32)(x * H * D + z0 + zl);
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                if (q >= np) break;
                const u32 keep = ((mbits >> q) & 1u) ? 0x01010101u : 0u;
                u32 rr[WCELLS / 4];
#pragma unroll
                for (int i = 0; i < WCELLS / 4; ++i) rr[i] = (G[i][q >> 3] >> (q & 7)) & keep;
                u8* pb = out + (yg + q) * D;                          // uniform
                if (!RAGGED || z0 + zl + WCELLS <= D) {
                    if (WCELLS == 16) { u32x4 r; r.x = rr[0]; r.y = rr[1]; r.z = rr[2 % (WCELLS / 4)]; r.w = rr[3 % (WCELLS / 4)]; *(u32x4_a1*)(pb + ooff) = r; }
                    else { u32x2 r; r.x = rr[0]; r.y = rr[1]; *(u32x2_a1*)(pb + ooff) = r; }
                } else {
                    typedef u32 u32_a1 __attribute__((aligned(1)));
                    const int k = (int)(D - z0 - zl);                  // 1..15 bytes: whole dwords, then bytes
                    for (int j = 0; j < (k >> 2); ++j) *(u32_a1*)(pb + ooff + 4 * j) = rr[j];
                    for (int b = k & ~3; b < k; ++b) pb[ooff + b] = (u8)(rr[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
        __syncthreads();
    }
    if (hib & 0xfefefefeu) atomicOr(big_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// Wide-tile form for large grids: 128 x 128 (x,z) output tiles, 16 planes per pass (one u16 per staged source voxel).
// k_rotate_bits32 fetches 2.7 GB for 1.07 GB of input at 1024^3: its 64-wide tiles cut the source rows into ~64-byte
// segments of 128-byte lines, and its 64-byte output runs are half lines.  Here a footprint row is up to 181 bytes
// (lines touched per tile ~ T^2/128 + sqrt(2) T: 2.4x at T = 128 against 3.8x at T = 64 before the L2) and every
// output run is a whole 128-byte line.  The table evaluation serves 16 planes per mux tree instead of 32 -- it was
// 1 % of the time.  512 threads x 32 cells; LDS 184 x 208 x 2 B (two workgroups per CU).  Measured at 1024^3, 45 degrees:
// FETCH_SIZE 2.04 GB (64-wide tiles: 2.72 GB), 0.66 ms against 0.74 ms; phase ablation: staging 0.35 ms, stores 0.24 ms.
// ------------------------------------------------------------------------------------------------
constexpr int XT = 128, XTHREADS = 512, XCELLS = 32, XTPR = XT / XCELLS;
constexpr int XROWS = 184, XPITCH = 208;
constexpr int XMAXU = (XROWS * (XPITCH / 16) + XTHREADS - 1) / XTHREADS;

// Footprint of every 128 x 128 tile, once per step (it does not depend on the plane): bounding box and, per footprint row,
// the leftmost / rightmost tap column.  ~64 K LDS atomics per tile -- done here once instead of once per (tile, plane block),
// by four workgroups per tile (32 x-rows each, rows relative to the quarter's own first row); the tile kernel merges the four.
struct TileRows { int bb[4]; int rmin[XROWS], rmax[XROWS]; };
constexpr int XQ = 4;

__global__ __launch_bounds__(XTHREADS) void k_rot_tile_rows(const CellRec* __restrict__ cells, i64 W, i64 D, int ntz, TileRows* __restrict__ info) {
    __shared__ int bb[4];
    __shared__ int rmin[XROWS], rmax[XROWS];
    const int tid = threadIdx.x;
    const int t = blockIdx.x / XQ, quarter = blockIdx.x % XQ;
    const i64 x0 = (i64)(t / ntz) * XT, z0 = (i64)(t % ntz) * XT;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    if (tid < XROWS) { rmin[tid] = 0x7fffffff; rmax[tid] = -1; }
    __syncthreads();
    // 512 threads on 32 x-rows x 128 z: 16 threads per row, 8 cells each
    constexpr int QC = XT / 16;
    const int zl = (tid % 16) * QC, xl0 = quarter * (XT / XQ) + tid / 16;
    const i64 x = x0 + xl0;
    int mn0 = 0x7fffffff, mx0 = -1, mn2 = 0x7fffffff, mx2 = -1;
    if (x < W) {
        for (int c = 0; c < QC; ++c) {
            const i64 z = z0 + zl + c;
            if (z >= D) break;
            const CellRec r = cells[x * D + z];
            if (r.src == 0xffffffffu) continue;
            const int s0 = (int)(r.src >> 16), s2 = (int)(r.src & 0xffffu);
            const int e0 = s0 + (int)((r.lut >> 16) & 1u), e2 = s2 + (int)((r.lut >> 17) & 1u);
            mn0 = s0 < mn0 ? s0 : mn0; mx0 = e0 > mx0 ? e0 : mx0;
            mn2 = s2 < mn2 ? s2 : mn2; mx2 = e2 > mx2 ? e2 : mx2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn0 = min(mn0, __shfl_xor(mn0, o)); mx0 = max(mx0, __shfl_xor(mx0, o));
        mn2 = min(mn2, __shfl_xor(mn2, o)); mx2 = max(mx2, __shfl_xor(mx2, o));
    }
    if ((tid & 63) == 0 && mx0 >= 0) { atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); atomicMin(&bb[2], mn2); atomicMax(&bb[3], mx2); }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1];
    const bool fits = bx1 >= 0 && bx1 - bx0 + 2 <= XROWS;
    if (fits && x < W) {
        for (int c = 0; c < QC; ++c) {
            const i64 z = z0 + zl + c;
            if (z >= D) break;
            const CellRec r = cells[x * D + z];
            if (r.src == 0xffffffffu) continue;
            const int rr = (int)(r.src >> 16) - bx0, s2 = (int)(r.src & 0xffffu);
            const int e2 = s2 + (int)((r.lut >> 17) & 1u);
            atomicMin(&rmin[rr], s2); atomicMax(&rmax[rr], e2);
            if ((r.lut >> 16) & 1u) { atomicMin(&rmin[rr + 1], s2); atomicMax(&rmax[rr + 1], e2); }
        }
    }
    __syncthreads();
    TileRows* o = info + (i64)t * XQ + quarter;
    if (tid < 4) o->bb[tid] = fits ? bb[tid] : (tid == 1 && bx1 >= 0 ? 0x7ffffff0 : bb[tid]);   // a quarter that does not fit poisons the tile's box
    if (tid < XROWS) { o->rmin[tid] = rmax[tid] >= 0 ? rmin[tid] : 0; o->rmax[tid] = rmax[tid]; }
}

template <bool RAGGED>
__global__ __launch_bounds__(XTHREADS, 4) void k_rotate_bits16w(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_wh,
                                                                const CellRec* __restrict__ cells, const TileRows* __restrict__ info, i64 W, i64 H,
                                                                i64 D, int TY, int ntz, int ntiles, int* __restrict__ big_flag,
                                                                const u8* __restrict__ mask_src) {
    extern __shared__ __attribute__((aligned(16))) u8 xlds[];
    unsigned short* tile = (unsigned short*)xlds;                                  // XROWS * XPITCH
    int* rmin = (int*)(xlds + XROWS * XPITCH * 2);                                 // XROWS
    int* rmax = rmin + XROWS;                                                      // XROWS
    unsigned short* ustart = (unsigned short*)(rmax + XROWS);                      // XROWS + 1 (+1 pad)
    u8* rc0 = (u8*)(ustart + XROWS + 2);                                           // XROWS
    u8* urow = rc0 + XROWS;                                                        // XROWS * XPITCH / 16
    int* bb = (int*)(xlds + XROWS * XPITCH * 2 + 2 * XROWS * 4 + (XROWS + 2) * 2 + XROWS + XROWS * (XPITCH / 16) + 8);
    bb = (int*)(((uintptr_t)bb + 3) & ~(uintptr_t)3);
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
    typedef u32 u32_a1 __attribute__((aligned(1)));
    const int tid = threadIdx.x;
    const int chunk = (ntiles + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int t = (int)(blockIdx.x & 7u) * chunk + slot % chunk;
    if (t >= ntiles) return;                          // whole workgroup, before any barrier
    const i64 x0 = (i64)(t / ntz) * XT, z0 = (i64)(t % ntz) * XT;
    const i64 y_beg = (i64)(slot / chunk) * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    const TileRows* qi = info + (i64)t * XQ;
    if (tid == 0) {
        int b0 = 0x7fffffff, b1 = -1, b2 = 0x7fffffff, b3 = -1;
        for (int q = 0; q < XQ; ++q)
            if (qi[q].bb[1] >= 0) { b0 = min(b0, qi[q].bb[0]); b1 = max(b1, qi[q].bb[1]); b2 = min(b2, qi[q].bb[2]); b3 = max(b3, qi[q].bb[3]); }
        bb[0] = b0; bb[1] = b1; bb[2] = b2; bb[3] = b3;
    }
    __syncthreads();
    if (tid < XROWS) {                                // row tid of the tile's box = row tid + bb[0] - bb_q[0] of quarter q
        int lo = 0x7fffffff, hi = -1;
        for (int q = 0; q < XQ; ++q) {
            if (qi[q].bb[1] < 0) continue;
            const int rq = tid + bb[0] - qi[q].bb[0];
            if (rq >= 0 && rq < XROWS && qi[q].rmax[rq] >= 0) { lo = min(lo, (int)qi[q].rmin[rq]); hi = max(hi, (int)qi[q].rmax[rq]); }
        }
        rmin[tid] = lo; rmax[tid] = hi;
    }
    __syncthreads();
    const int zl = (tid % XTPR) * XCELLS, xl0 = tid / XTPR;
    const i64 x = x0 + xl0;
    const bool row_ok = x < W && z0 + zl < D;
    const int bx0 = bb[0], bx1 = bb[1], bz0 = bb[2] & ~15, bz1 = bb[3];
    const bool any_valid = bx1 >= 0;
    const int nrows = any_valid ? bx1 - bx0 + 1 : 0;
    const int nu = any_valid ? (bz1 - bz0) / 16 + 1 : 0;
    const bool fits = nrows + 1 <= XROWS && nu * 16 <= XPITCH;
    if (any_valid && !fits) { if (tid == 0) atomicOr(big_flag, 1); }
    u32 cellw[XCELLS];                                // table << 16 | LDS offset (0xffff: outputs 0)
#pragma unroll
    for (int c = 0; c < XCELLS; ++c) cellw[c] = 0xffffu;
    if (row_ok && fits) {
        typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
        const u32x4_a8* cp = (const u32x4_a8*)(cells + x * D + z0 + zl);      // XCELLS records (the table is padded)
#pragma unroll
        for (int k = 0; k < XCELLS / 2; ++k) {
            const u32x4 v = cp[k];
            const bool ok0 = !RAGGED || z0 + zl + 2 * k < D, ok1 = !RAGGED || z0 + zl + 2 * k + 1 < D;
            if (ok0 && v.x != 0xffffffffu) cellw[2 * k] = (v.y << 16) | (u32)(((int)(v.x >> 16) - bx0) * XPITCH + ((int)(v.x & 0xffffu) - bz0));
            if (ok1 && v.z != 0xffffffffu) cellw[2 * k + 1] = (v.w << 16) | (u32)(((int)(v.z >> 16) - bx0) * XPITCH + ((int)(v.z & 0xffffu) - bz0));
        }
    }
    __syncthreads();
    if (tid < 64) {                                    // wave 0: exclusive scan of the per-row unit counts (rows tid, tid+64, tid+128)
        int n[3], inc[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int r = tid + 64 * h;
            n[h] = 0;
            if (r < nrows && fits && rmax[r] >= 0) {
                rc0[r] = (u8)((rmin[r] - bz0) >> 4);
                n[h] = ((rmax[r] - bz0) >> 4) - ((rmin[r] - bz0) >> 4) + 1;
            } else if (r < XROWS) rc0[r] = 0;
            inc[h] = n[h];
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int h = 0; h < 3; ++h) { const int a = __shfl_up(inc[h], o); if (tid >= o) inc[h] += a; }
        }
        const int tot0 = __shfl(inc[0], 63), tot1 = tot0 + __shfl(inc[1], 63);
        ustart[tid] = (unsigned short)(inc[0] - n[0]);
        ustart[tid + 64] = (unsigned short)(tot0 + inc[1] - n[1]);
        if (tid + 128 <= XROWS) ustart[tid + 128] = (unsigned short)(tot1 + inc[2] - n[2]);
        if (tid == 63) ustart[XROWS] = (unsigned short)(tot1 + inc[2]);
    }
    __syncthreads();
    const int nunits = (any_valid && fits) ? (int)ustart[XROWS] : 0;
    if (tid < nrows && fits) {
        const int u0 = ustart[tid], u1 = ustart[tid + 1];
        for (int u = u0; u < u1; ++u) urow[u] = (u8)tid;
    }
    __syncthreads();
    u32 hib = 0;
    for (i64 yg = y_beg; yg < y_end; yg += 16) {
        const int np = (int)(y_end - yg < 16 ? y_end - yg : 16);
        u32 mbits = 0;
        if (row_ok) {
            if (!mask_wh) mbits = 0xffffu;
            else for (int q = 0; q < np; ++q) mbits |= (u32)(mask_wh[x * H + yg + q] != 0) << q;
        }
        // ---- stage the footprint of 16 planes: 16 voxels x 16 planes per unit, 8 planes of 16-byte loads in flight per lane
#pragma unroll 1
        for (int j = 0; j < XMAXU; ++j) {
            const int i = tid + XTHREADS * j;
            if (i >= nunits) break;
            const int r = urow[i];
            const int cu = (int)rc0[r] + (i - (int)ustart[r]);
            const u32 voff = (u32)(((i64)bx0 + r) * H * D + (i64)bz0 + 16 * cu);
            const bool whole = !RAGGED || (i64)bz0 + 16 * cu + 15 < D || (i64)voff + (yg + 15) * D + 16 <= W * H * D;
            const u32 msrc = src_plane_bits(mask_src, ((i64)bx0 + r) * H + yg, np);
            u32x4 wg[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                u32x4 d[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const u8* pb = in + (yg + 8 * g + q) * D;        // uniform
                    d[q] = (u32x4)(0u);
                    if (8 * g + q < np && ((msrc >> (8 * g + q)) & 1u)) {
                        if (whole) d[q] = *(const u32x4_a1*)(pb + voff);
                        else {
                            u32 t4[4] = {0, 0, 0, 0};
                            for (int b = 0; (i64)bz0 + 16 * cu + b < D; ++b) t4[b >> 2] |= (u32)pb[voff + b] << (8 * (b & 3));
                            d[q].x = t4[0]; d[q].y = t4[1]; d[q].z = t4[2]; d[q].w = t4[3];
                        }
                    }
                }
                u32x4 wv = (u32x4)(0u);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    wv.x |= (d[q].x & 0x01010101u) << q; wv.y |= (d[q].y & 0x01010101u) << q;
                    wv.z |= (d[q].z & 0x01010101u) << q; wv.w |= (d[q].w & 0x01010101u) << q;
                    hib |= d[q].x | d[q].y | d[q].z | d[q].w;
                }
                wg[g] = wv;
            }
            // voxel v of the unit: u16 = planes 0..7 (byte v of wg[0]) | planes 8..15 (byte v of wg[1]) << 8
            u32x4 lo, hi;
            lo.x = pperm(wg[1].x, wg[0].x, 0x05010400u); lo.y = pperm(wg[1].x, wg[0].x, 0x07030602u);
            lo.z = pperm(wg[1].y, wg[0].y, 0x05010400u); lo.w = pperm(wg[1].y, wg[0].y, 0x07030602u);
            hi.x = pperm(wg[1].z, wg[0].z, 0x05010400u); hi.y = pperm(wg[1].z, wg[0].z, 0x07030602u);
            hi.z = pperm(wg[1].w, wg[0].w, 0x05010400u); hi.w = pperm(wg[1].w, wg[0].w, 0x07030602u);
            unsigned short* trow = tile + r * XPITCH + 16 * cu;
            *(u32x4*)(trow) = lo; *(u32x4*)(trow + 8) = hi;
        }
        __syncthreads();
        // ---- evaluate 32 cells x 16 planes; write np planes of this thread's 32-byte run
        if (row_ok) {
            u32 G[XCELLS / 4][2];        // G[i][g]: byte c = planes 8g..8g+7 of cell 4i + c
#pragma unroll
            for (int i = 0; i < XCELLS / 4; ++i) {
                u32 R[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 cw = cellw[4 * i + c], o = cw & 0xffffu;
                    R[c] = 0;
                    if (o != 0xffffu) R[c] = lut_apply32(cw >> 16, tile[o], tile[o + 1], tile[o + XPITCH], tile[o + XPITCH + 1]);
                }
                const u32 l01 = pperm(R[1], R[0], 0x05010400u), l23 = pperm(R[3], R[2], 0x05010400u);
                G[i][0] = pperm(l23, l01, 0x05040100u); G[i][1] = pperm(l23, l01, 0x07060302u);
            }
            const u32 ooff = (u32)(x * H * D + z0 + zl);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q >= np) break;
                const u32 keep = ((mbits >> q) & 1u) ? 0x01010101u : 0u;
                u32 rr[XCELLS / 4];
#pragma unroll
                for (int i = 0; i < XCELLS / 4; ++i) rr[i] = (G[i][q >> 3] >> (q & 7)) & keep;
                u8* pb = out + (yg + q) * D;                          // uniform
                if (!RAGGED || z0 + zl + XCELLS <= D) {
                    u32x4 r0, r1;
                    r0.x = rr[0]; r0.y = rr[1]; r0.z = rr[2]; r0.w = rr[3]; r1.x = rr[4]; r1.y = rr[5]; r1.z = rr[6]; r1.w = rr[7];
                    *(u32x4_a1*)(pb + ooff) = r0; *(u32x4_a1*)(pb + ooff + 16) = r1;
                } else {
                    const int k = (int)(D - z0 - zl);                  // 1..31 bytes: whole dwords, then bytes
                    for (int jj = 0; jj < (k >> 2); ++jj) *(u32_a1*)(pb + ooff + 4 * jj) = rr[jj];
                    for (int b = k & ~3; b < k; ++b) pb[ooff + b] = (u8)(rr[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
        __syncthreads();
    }
    if (hib & 0xfefefefeu) atomicOr(big_flag, 1);
}
constexpr size_t kXLds = (size_t)XROWS * XPITCH * 2 + 2 * XROWS * 4 + (XROWS + 2) * 2 + XROWS + XROWS * (XPITCH / 16) + 8 + 4 + 16;

// ------------------------------------------------------------------------------------------------
// Packed-footprint form for the largest grids: 256 x 256 (x,z) output tiles, 8 planes per pass (k_rotate_bits8p).
// The read traffic of a tiled rotation is set at LINE level: a footprint row of a T x T tile is a segment that starts and ends
// inside 128-byte lines, so a tile touches ~ T^2/128 + T (|sin| + |cos|) lines for T^2/128 lines of data -- 2.4x at T = 128 and
// 45 degrees (k_rotate_bits16w, measured 1.9x behind the L2), 1.7x at T = 256.  What kept T at 128 was LDS: the bounding box of a
// rotated 256-tile is 364 x 364 voxels.  Here
//  * the footprint is stored ROW-PACKED: row r occupies the 16-voxel units [ustart[r], ustart[r+1]) of LDS, first column
//    start16[r] -- exactly the rotated square (T^2 + slack = ~72 KB at one byte per voxel: bit q = plane yg + q), two
//    workgroups per CU;
//  * a cell record is ONE dword (footprint row, column within the row, 14 table bits -- entries 0 and 15 of SciPy's result
//    table are constants); the LDS offsets of its two tap rows come from a 384-entry row table in LDS (one ds_read_b32);
//  * everything a tile needs besides the voxels (row table, unit list, cell records) is computed ONCE per step by three small
//    kernels and read through the L2 by every (tile, plane chunk) workgroup;
//  * workgroup -> (tile, plane chunk): all tiles of one plane chunk run on ONE XCD (blockIdx % 8), so the lines two neighbouring
//    footprints share meet in that XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int PT = 256, PTHREADS = 512;
constexpr int PROWS = 384;                          // footprint rows: T (|sin| + |cos|) + 2 <= 365
constexpr int PLDS_DATA = 76 * 1024;                // packed footprint, one byte per voxel
constexpr int PMAXUNITS = PLDS_DATA / 16;           // 16-voxel units
constexpr int PUPT = (PMAXUNITS + PTHREADS - 1) / PTHREADS;   // units per thread (10)
constexpr int PPARTS = 8;                           // set-up: 32 x-rows of a tile per workgroup
constexpr size_t kPLds = (size_t)PLDS_DATA + (PROWS + 1) * 4 + 16 * 4;

struct PPart { int bb[4]; int rmin[PROWS], rmax[PROWS]; };
struct PTile {
    int bx0, nrows, nunits, fits;
    u32 dict[16];                     // the step's distinct 14-bit tables (bits 1..14 of SciPy's result table), ascending; ndict <= 15
    int ndict;
    int atab[PROWS + 1];              // LDS byte of source voxel (row r, column s2) = atab[r] + s2 (rows nobody stages: the row above)
    int start16[PROWS];               // first staged column of row r (multiple of 16)
    u32 voff[PMAXUNITS];              // voxel offset of unit i at plane 0: (bx0 + r) * H * D + start16[r] + 16 k
    unsigned short srow[PMAXUNITS];   // r of unit i
};

// per (tile, 32 x-rows): bounding rows and per-row tap extents, rows relative to the part's own first row
__global__ __launch_bounds__(PTHREADS) void k_rot8_parts(const CellRec* __restrict__ cells, i64 W, i64 D, int ntz, PPart* __restrict__ parts) {
    __shared__ int bb[4];
    __shared__ int rmin[PROWS], rmax[PROWS];
    const int tid = threadIdx.x;
    const int t = blockIdx.x / PPARTS, part = blockIdx.x % PPARTS;
    const i64 x0 = (i64)(t / ntz) * PT, z0 = (i64)(t % ntz) * PT;
    if (tid == 0) { bb[0] = 0x7fffffff; bb[1] = -1; bb[2] = 0x7fffffff; bb[3] = -1; }
    if (tid < PROWS) { rmin[tid] = 0x7fffffff; rmax[tid] = -1; }
    __syncthreads();
    // 512 threads on 32 x-rows x 256 z: 16 threads per row, 16 cells each
    const int zl = (tid & 15) * 16, xl0 = part * (PT / PPARTS) + (tid >> 4);
    const i64 x = x0 + xl0;
    int mn0 = 0x7fffffff, mx0 = -1;
    u32 src[16], lut[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const i64 z = z0 + zl + c;
        src[c] = 0xffffffffu; lut[c] = 0;
        if (x < W && z < D) { const CellRec r = cells[x * D + z]; src[c] = r.src; lut[c] = r.lut; }
        if (src[c] == 0xffffffffu) continue;
        const int s0 = (int)(src[c] >> 16), e0 = s0 + (int)((lut[c] >> 16) & 1u);
        mn0 = s0 < mn0 ? s0 : mn0; mx0 = e0 > mx0 ? e0 : mx0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn0 = min(mn0, __shfl_xor(mn0, o)); mx0 = max(mx0, __shfl_xor(mx0, o)); }
    if ((tid & 63) == 0 && mx0 >= 0) { atomicMin(&bb[0], mn0); atomicMax(&bb[1], mx0); }
    __syncthreads();
    const int bx0 = bb[0], bx1 = bb[1];
    const bool fits = bx1 >= 0 && bx1 - bx0 + 2 <= PROWS;
    if (fits) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (src[c] == 0xffffffffu) continue;
            const int rr = (int)(src[c] >> 16) - bx0, s2 = (int)(src[c] & 0xffffu);
            const int e2 = s2 + (int)((lut[c] >> 17) & 1u);
            atomicMin(&rmin[rr], s2); atomicMax(&rmax[rr], e2);
            if ((lut[c] >> 16) & 1u) { atomicMin(&rmin[rr + 1], s2); atomicMax(&rmax[rr + 1], e2); }
        }
    }
    __syncthreads();
    PPart* o = parts + (i64)t * PPARTS + part;
    if (tid < 4) o->bb[tid] = fits ? bb[tid] : (tid == 1 && bx1 >= 0 ? 0x7ffffff0 : bb[tid]);   // a part that does not fit poisons the tile
    if (tid < PROWS) { o->rmin[tid] = rmax[tid] >= 0 ? rmin[tid] : 0; o->rmax[tid] = rmax[tid]; }
}

// per tile: merge the parts, lay the rows out in LDS, list the staging units
__global__ __launch_bounds__(PTHREADS) void k_rot8_tiles(const PPart* __restrict__ parts, const u32* __restrict__ lutmap, i64 H, i64 D,
                                                         PTile* __restrict__ tiles) {
    __shared__ int sb[2 + PPARTS];
    __shared__ int dcount[PTHREADS / 64 + 1];
    __shared__ u32 dict[16];
    __shared__ int lo[PROWS + 1], hi[PROWS + 1], ust[PROWS + 2];
    __shared__ int wsum[PTHREADS / 64];
    const int tid = threadIdx.x;
    const PPart* qi = parts + (i64)blockIdx.x * PPARTS;
    PTile* ti = tiles + blockIdx.x;
    if (tid < 64) {                     // wave 0: bounding rows over the parts
        int b0 = 0x7fffffff, b1 = -1;
        if (tid < PPARTS && qi[tid].bb[1] >= 0) { b0 = qi[tid].bb[0]; b1 = qi[tid].bb[1]; }
        if (tid < PPARTS) sb[2 + tid] = b1 >= 0 ? b0 : 0x7fffffff;          // first row of part `tid` (none: huge)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { b0 = min(b0, __shfl_xor(b0, o)); b1 = max(b1, __shfl_xor(b1, o)); }
        if (tid == 0) { sb[0] = b0; sb[1] = b1; }
    }
    __syncthreads();
    const int bx0 = sb[0], bx1 = sb[1];
    const int nrows = bx1 >= 0 ? bx1 - bx0 + 1 : 0;
    const bool rows_fit = nrows + 1 <= PROWS;
    // row tid of the tile = row tid + bx0 - bb_q[0] of part q; the 16 loads of a row are independent of each other
    int n_units = 0;
    if (tid <= PROWS) {
        int l = 0x7fffffff, h = -1;
        if (rows_fit && tid < nrows) {
            int lq[PPARTS], hq[PPARTS];
#pragma unroll
            for (int q = 0; q < PPARTS; ++q) {
                const int rq = sb[2 + q] == 0x7fffffff ? -1 : tid + bx0 - sb[2 + q];
                const bool ok = rq >= 0 && rq < PROWS;
                lq[q] = ok ? qi[q].rmin[rq] : 0; hq[q] = ok ? qi[q].rmax[rq] : -1;
            }
#pragma unroll
            for (int q = 0; q < PPARTS; ++q)
                if (hq[q] >= 0) { l = min(l, lq[q]); h = max(h, hq[q]); }
        }
        lo[tid] = h >= 0 ? (l & ~15) : 0;
        hi[tid] = h;
        n_units = h >= 0 ? ((h - (l & ~15)) >> 4) + 1 : 0;
    }
    // exclusive prefix sum of the per-row unit counts (rows 0..PROWS, one per thread)
    int inc = n_units;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if ((tid & 63) >= o) inc += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    if (tid <= PROWS) ust[tid] = base + inc - n_units;
    if (tid == PTHREADS - 1) ust[PROWS + 1] = base + inc;
    __syncthreads();
    // dictionary of the step's tables: word `tid` of the presence bitmap, set bits listed in ascending order (every tile block
    // builds the same list; it is 512 words)
    {
        const u32 w = lutmap[tid];
        const int cnt = __popc(w);
        int pre = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o); if ((tid & 63) >= o) pre += v; }
        if ((tid & 63) == 63) dcount[tid >> 6] = pre;
        __syncthreads();
        int dbase = 0;
        for (int k = 0; k < (tid >> 6); ++k) dbase += dcount[k];
        int pos = dbase + pre - cnt;
        for (u32 m = w; m; m &= m - 1) { if (pos < 16) dict[pos] = (u32)(32 * tid + __builtin_ctz(m)); ++pos; }
        if (tid == PTHREADS - 1) dcount[PTHREADS / 64] = dbase + pre;
        __syncthreads();
    }
    const int ndict = dcount[PTHREADS / 64];
    const int nunits = ust[PROWS + 1];
    const bool fits = rows_fit && nunits <= PMAXUNITS && ndict <= 15;
    if (tid == 0) { ti->bx0 = bx0; ti->nrows = nrows; ti->nunits = fits ? nunits : 0; ti->fits = (fits || nrows == 0) ? 1 : 0; ti->ndict = ndict; }
    if (tid < 16) ti->dict[tid] = tid < ndict ? dict[tid] : 0xffffffffu;
    if (tid <= PROWS) {
        // a row nobody stages is only ever addressed as an UNUSED x tap (weight 0): point it at the row above
        int rr = tid;
        while (rr > 0 && hi[rr] < 0) --rr;
        ti->atab[tid] = hi[rr] >= 0 ? 16 * ust[rr] - lo[rr] : 0;
    }
    if (tid < PROWS) ti->start16[tid] = lo[tid];
    if (fits) {
        for (int i = tid; i < nunits; i += PTHREADS) {            // unit i -> its row: the last r with ust[r] <= i
            int a = 0, b = PROWS;
            while (a < b) { const int m = (a + b + 1) >> 1; if (ust[m] <= i) a = m; else b = m - 1; }
            ti->voff[i] = (u32)(((i64)bx0 + a) * H * D + lo[a] + 16 * (i - ust[a]));
            ti->srow[i] = (unsigned short)a;
        }
    }
}

// One 16-byte record per run of 16 cells along z (the unit a thread of the tile kernel evaluates): consecutive cells of a run
// step through the footprint by (-1 or 0, 0 or +1) -- a rotation by 0..90 degrees moves the source row down and the source
// column up along +z -- so a run is its first cell's position, two step bits per cell and a 4-bit table index per cell:
//   w0 = footprint row r0 | source column s2 << 9 (of the first cell that is not void) ; w1 bit i = row step into cell i, bit 16 + i =
//   column step into cell i ; w2, w3 = 4-bit dictionary index of cells 0..7, 8..15 (15 = void: outputs 0, takes no step)
// 1 byte per cell instead of a position + table dword: a thread keeps its 8 runs in registers for all its passes.  A run that
// does not follow the pattern (or a tile that does not fit) marks the tile unfit: the step is then redone by the arithmetic
// kernel (device-side flag).
struct RunRec { u32 w0, w1, w2, w3; };

// one lane per cell, 16 consecutive lanes per run; a row has nruns = ceil(D / 16) runs, the cells of the last one past D are void
__global__ __launch_bounds__(256) void k_rot8_pack(const CellRec* __restrict__ cells, PTile* __restrict__ tiles, i64 W, i64 D, int ntz, int nruns,
                                                   RunRec* __restrict__ runs) {
    const i64 gr = (i64)blockIdx.x * 16 + (threadIdx.x >> 4);       // run index x * nruns + rz
    const int c = (int)(threadIdx.x & 15);                          // position in the run
    const bool inside = gr < W * nruns;
    const i64 x = inside ? gr / nruns : 0, z = inside ? 16 * (gr - x * nruns) + c : 0;
    const bool cell = inside && z < D;
    PTile* ti = tiles + (x / PT) * ntz + (cell ? z : 0) / PT;
    const bool tile_ok = cell && ti->fits;
    bool live = false, bad = false;
    int r = 0, s2 = 0;
    u32 k = 15;
    if (tile_ok) {
        const CellRec cr = cells[x * D + z];
        if (cr.src != 0xffffffffu) {
            live = true;
            r = (int)(cr.src >> 16) - ti->bx0;
            s2 = (int)(cr.src & 0xffffu);
            const u32 t = (cr.lut >> 1) & 0x3fffu;
            const int nd = ti->ndict;
            k = 0;
            while ((int)k < nd && ti->dict[k] != t) ++k;
            if ((int)k >= nd) { bad = true; k = 15; }
        }
    }
    // the run's live cells must be one contiguous block; steps are taken between consecutive live cells
    const u64 lv = __ballot(live);
    const u32 seg = (u32)(lv >> (threadIdx.x & 48)) & 0xffffu;
    const int first = seg ? __builtin_ctz(seg) : 0;
    if (seg && (((seg >> first) + 1u) & (seg >> first)) != 0u) bad = true;
    const int pr = __shfl_up(r, 1), ps2 = __shfl_up(s2, 1);
    u32 w1 = 0;
    if (live && c > first) {
        const int dr = pr - r, dc = s2 - ps2;
        if (dr < 0 || dr > 1 || dc < 0 || dc > 1) bad = true;
        w1 = ((u32)(dr & 1) << c) | ((u32)(dc & 1) << (16 + c));
    }
    u32 w0 = (live && c == first) ? ((u32)r | ((u32)s2 << 9)) : 0u;
    u32 w2 = c < 8 ? k << (4 * c) : 0u, w3 = c >= 8 ? k << (4 * (c - 8)) : 0u;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        w0 |= (u32)__shfl_xor((int)w0, o); w1 |= (u32)__shfl_xor((int)w1, o);
        w2 |= (u32)__shfl_xor((int)w2, o); w3 |= (u32)__shfl_xor((int)w3, o);
    }
    if (bad) ti->fits = 0;                                          // benign race: every writer stores 0
    if (inside && c == 0) { RunRec rec = {w0, w1, w2, w3}; runs[gr] = rec; }
}

__device__ __forceinline__ u32 lut_apply14(u32 lut14, u32 t00, u32 t01, u32 t10, u32 t11) {
    u32 L[16];
    L[0] = 0u; L[15] = ~0u;                                   // no tap set -> 0 ; all four set -> the weights sum to 1 -> 1
#pragma unroll
    for (int k = 1; k < 15; ++k) L[k] = (u32)__builtin_amdgcn_sbfe((int)lut14, k - 1, 1);
    u32 g[8], h[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = bsel(t00, L[2 * j + 1], L[2 * j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = bsel(t01, g[2 * j + 1], g[2 * j]);
    const u32 m0 = bsel(t10, h[1], h[0]), m1 = bsel(t10, h[3], h[2]);
    return bsel(t11, m1, m0);                                 // bits 0..7: planes ; bits 8.. : garbage (callers pick byte 0)
}

// bit q of the result: byte q of the 8 mask bytes at p is non-zero (q < np)
__device__ __forceinline__ u32 mask8(const u8* __restrict__ p, int np) {
    u32 bits = 0;
    if ((((uintptr_t)p) & 3u) == 0 && np == 8) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            u32 t4 = *(const u32*)(p + 4 * k);
            t4 |= t4 >> 4; t4 |= t4 >> 2; t4 |= t4 >> 1; t4 &= 0x01010101u;
            bits |= ((t4 * 0x01020408u) >> 24) << (4 * k);
        }
    } else {
        for (int q = 0; q < np; ++q) bits |= (u32)(p[q] != 0) << q;
    }
    return bits;
}

// 512 threads, two workgroups per CU (verified resident together: tools/kbench5.hip).  Tried and measured no better: one 1024-thread
// workgroup whose two wave groups alternate stage / evaluate by construction, random start staggers, and a skewed cyclic walk of
// the runs against LDS bank conflicts -- none of them moved the time, because the kernel was bound by VALU ISSUE (below).
// ODD: D % 16 != 0 (the reference's real shapes: 355, 437 ...).  Source units and output runs are then 16 bytes at arbitrary byte
// addresses (gfx950 serves those at full speed), the last run of an output row is stored byte-wise up to D, and the last unit of a
// source row reads up to 15 bytes past the row -- the next row's voxels, which no tap addresses; the launcher takes this path only
// when the input ALLOCATION extends 16 bytes past the volume (hipMemGetAddressRange; the library's own buffers always do).
template <bool SRCMASK, bool ODD>
__global__ __launch_bounds__(PTHREADS, 4) void k_rotate_bits8p(const u8* __restrict__ in, u8* __restrict__ out, const u8* __restrict__ mask_wh,
                                                               const RunRec* __restrict__ runs, const PTile* __restrict__ tiles, i64 W, i64 H, i64 D,
                                                               int TY, int ntz, int ntiles, int nchunks, int* __restrict__ big_flag,
                                                               const u8* __restrict__ mask_src, int abl, int nruns, int nsplit) {
    typedef u32 u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
    extern __shared__ __attribute__((aligned(16))) u8 plds[];
    int* atab = (int*)(plds + PLDS_DATA);              // PROWS + 1 entries
    u32* dict = (u32*)(atab + PROWS + 1);              // 16 entries
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    // nsplit = 2 / 4 (grids whose tiles x plane chunks do not fill the chip): that many workgroups share a (tile, chunk); each stages
    // the whole footprint and evaluates 8 / nsplit of the run slots.  They follow each other on one XCD, so the footprint comes out of
    // its L2 for all but the first.
    const int slot2 = (int)(blockIdx.x >> 3);
    const int sub = slot2 % nsplit, slot = slot2 / nsplit;
    const int t = slot % ntiles;
    const int chunk = (slot / ntiles) * 8 + (int)(blockIdx.x & 7u);
    if (chunk >= nchunks) return;                      // whole workgroup, before any barrier
    const PTile* ti = tiles + t;
    if (!ti->fits) { if (tid == 0) atomicOr(big_flag, 1); return; }
    const i64 x0 = (i64)(t / ntz) * PT, z0 = (i64)(t % ntz) * PT;
    const i64 y_beg = (i64)chunk * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    const int nunits = ti->nunits;
    const i64 bx0 = ti->bx0;
    for (int i = tid; i <= PROWS; i += PTHREADS) atab[i] = ti->atab[i];
    if (tid < 16) dict[tid] = ti->dict[tid];
    u32 uvoff[PUPT], usrow[(PUPT + 1) / 2];           // usrow: footprint row of unit j, two per register (source-mask form only)
#pragma unroll
    for (int j = 0; j < (PUPT + 1) / 2; ++j) usrow[j] = 0;
#pragma unroll
    for (int j = 0; j < PUPT; ++j) {
        const int i = tid + PTHREADS * j;
        uvoff[j] = i < nunits ? ti->voff[i] : 0xffffffffu;
        if (SRCMASK && i < nunits) usrow[j >> 1] |= (u32)ti->srow[i] << (16 * (j & 1));
    }
    // This thread's 8 runs of 16 cells, in registers for the whole life of the workgroup: run slot k is x-row 32 k + tid / 16, cells
    // 16 (tid % 16) .. + 15 (16 lanes cover one 256-byte output row).
    const int zc = 16 * (tid & 15);
    u32x4 run[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const i64 x = x0 + 32 * k + (tid >> 4), z = z0 + zc;
        run[k] = u32x4{0u, 0u, 0xffffffffu, 0xffffffffu};          // outside the grid: 16 void cells at the footprint origin
        if (x < W && z < D) run[k] = *(const u32x4*)(runs + x * nruns + (z >> 4));
    }
    __syncthreads();
    // dictionary entry e as the kernel wants it: table bits 1..7 in byte 0, bits 8..14 in byte 1, table bit 15 (= 1 for every real
    // cell) in byte 2; entry 15 (void cells) = 0: an all-zero table evaluates to 0 whatever the taps are
    if (tid < 16) {
        const u32 raw = dict[tid], t14 = raw & 0x3fffu;
        dict[tid] = (tid == 15 || raw == 0xffffffffu) ? 0u : ((t14 & 0x7fu) | ((t14 >> 7) << 8) | (1u << 16));
    }
    const u32 voff_safe = nunits > 0 ? ti->voff[0] : 0u;          // what lanes without a unit load (and throw away)
    // ODD: the unit that holds the volume's very last voxels runs into the allocation's slack at plane H - 1; whatever lies there must
    // not trip the "value > 1" check (no tap addresses those columns).  Only workgroups of the last plane chunk whose footprint
    // reaches the last source row can hold that unit (risky, uniform); jr = which of this thread's units it is, keep = its bytes inside.
    const bool risky = ODD && y_end == H && bx0 + ti->nrows >= W;
    int jr = -1, keep_bytes = 16;
    if (risky) {
        const i64 total = W * H * D;
#pragma unroll
        for (int j = 0; j < PUPT; ++j)
            if (uvoff[j] != 0xffffffffu && (H - 1) * D + (i64)uvoff[j] + 16 > total) { jr = j; keep_bytes = (int)(total - (H - 1) * D - (i64)uvoff[j]); }
    }
    __syncthreads();
    u32 hib = 0;
    // The kernel is bound by VALU issue, not by HBM or LDS (SQ counters: 10 k vector instructions per wave and pass at one per four
    // cycles were 60 % of its time), so both phases are written for instruction count: unconditional loads, one v_lshl_or per dword
    // to bit-slice, and a table evaluation that serves FOUR cells per instruction.
    auto stage = [&](i64 yg) {
        const int np = (int)(y_end - yg < 8 ? y_end - yg : 8);
        if (nunits == 0 || (abl & 2)) return;
        // 16 voxels x 8 planes per unit, loaded as two halves of 4 planes; the next half's loads are issued before this one is packed.
        // Loads are never predicated: a lane without a unit re-reads unit 0, a plane past the chunk's end re-reads the last one (its
        // bits land in plane slots that are never stored).  Data is 0/1 (anything else raises the flag and the step is redone), so
        // "shift the dword by q and OR" bit-slices four voxels in one instruction.
        u32x4 d[2][4];
        auto load_half = [&](u32x4 (&dd)[4], int hh) {
            const int j = hh >> 1, q0 = 4 * (hh & 1);
            const u32 voff = uvoff[j] != 0xffffffffu ? uvoff[j] : voff_safe;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int qe = q0 + q < np ? q0 + q : np - 1;     // uniform
                dd[q] = ODD ? (u32x4)*(const u32x4a1*)(in + (yg + qe) * D + voff) : *(const u32x4*)(in + (yg + qe) * D + voff);
            }
        };
        load_half(d[0], 0);
        u32x4 wv = (u32x4)(0u);
#pragma unroll
        for (int hh = 0; hh < 2 * PUPT; ++hh) {
            if (hh + 1 < 2 * PUPT) load_half(d[(hh + 1) & 1], hh + 1);
            __builtin_amdgcn_sched_barrier(0);                    // nothing is predicated any more: keep the scheduler from hoisting all 80 loads
            const int j = hh >> 1, q0 = 4 * (hh & 1);
            u32 msrc = 0xffu;
            if (SRCMASK && uvoff[j] != 0xffffffffu) msrc = mask8(mask_src + (u32)((bx0 + ((usrow[j >> 1] >> (16 * (j & 1))) & 0xffffu)) * H) + yg, np);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 dd = d[hh & 1][q];
                if (ODD && risky && j == jr && yg + q0 + q >= H - 1) {             // plane H - 1 (and its re-reads past the chunk's end): the bytes past the volume are zero
                    const int kb = keep_bytes;
                    dd.x &= kb >= 4 ? ~0u : ((1u << (8 * (kb > 0 ? kb : 0))) - 1u);
                    dd.y &= kb >= 8 ? ~0u : (kb > 4 ? (1u << (8 * (kb - 4))) - 1u : 0u);
                    dd.z &= kb >= 12 ? ~0u : (kb > 8 ? (1u << (8 * (kb - 8))) - 1u : 0u);
                    dd.w &= kb >= 16 ? ~0u : (kb > 12 ? (1u << (8 * (kb - 12))) - 1u : 0u);
                }
                if (SRCMASK && !((msrc >> (q0 + q)) & 1u)) dd = (u32x4)(0u);      // the folded 0-degree carve: this source row / plane is dropped
                wv.x |= dd.x << (q0 + q); wv.y |= dd.y << (q0 + q); wv.z |= dd.z << (q0 + q); wv.w |= dd.w << (q0 + q);
                hib |= (dd.x | dd.y) | (dd.z | dd.w);
            }
            asm volatile("" : "+v"(hib));                         // fold the check in HERE: left alone, the scheduler keeps all 80 vectors alive for it
            if (hh & 1) {
                if (uvoff[j] != 0xffffffffu) *(u32x4*)(plds + 16 * (tid + PTHREADS * j)) = wv;
                wv = (u32x4)(0u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto evaluate = [&](i64 yg) {
        const int np = (int)(y_end - yg < 8 ? y_end - yg : 8);
        // ---- evaluate the thread's 8 runs; the run registers rotate by one per slot so that the loop body always names run[0] (no
        //      dynamic register indexing), and are back in place after the eighth
#pragma unroll 1
        for (int qq = 0; qq < 8; ++qq) {
            if (abl & 1) break;
            if (x0 + 32 * qq + 4 * (tid >> 6) >= W || (qq * nsplit) / 8 != sub) {     // all four x-rows of this wave's slot are past the grid (edge tiles), or the slot is another workgroup's
                const u32x4 t0 = run[0];
#pragma unroll
                for (int k = 0; k < 7; ++k) run[k] = run[k + 1];
                run[7] = t0;
                continue;
            }
            // Four cells at a time (rolled loop, one group's registers live): positions by the run's step bits, one row-table read
            // pair and one dictionary read per cell, 16 tap bytes -- then the taps of the four cells are packed bytewise into four
            // dwords and SciPy's table is applied to all four cells x 8 planes at once: mask L_k has byte c = 0xff where cell c's
            // table bit k is set (v_perm with selector bytes 0x0c / 0x0d), and the 15-select multiplexer tree runs once per group.
            const u32x4 rec = run[0];
            int r = (int)(rec.x & 511u);
            u32 s2 = rec.x >> 9;
            u32 G[4];
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const u32 wd = rec.y >> (4 * i), wk = (i < 2 ? rec.z : rec.w) >> (16 * (i & 1));
                u32 o0[4], o1[4], lt[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    r -= (int)((wd >> c) & 1u); s2 += (wd >> (16 + c)) & 1u;            // void cells carry no step bits
                    const int* ap = atab + r;
                    o0[c] = (u32)ap[0] + s2; o1[c] = (u32)ap[1] + s2;
                    lt[c] = dict[(wk >> (4 * c)) & 15u];
                }
                u32 tp[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { tp[c][0] = plds[o0[c]]; tp[c][1] = plds[o0[c] + 1]; tp[c][2] = plds[o1[c]]; tp[c][3] = plds[o1[c] + 1]; }
                u32 T[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) T[t] = tp[0][t] | (tp[1][t] << 8) | (tp[2][t] << 16) | (tp[3][t] << 24);
                const u32 PL = pperm(lt[1], lt[0], 0x0c0c0400u) | (pperm(lt[3], lt[2], 0x0c0c0400u) << 16);     // table bits 1..7 of the 4 cells
                const u32 PH = pperm(lt[1], lt[0], 0x0c0c0501u) | (pperm(lt[3], lt[2], 0x0c0c0501u) << 16);     // table bits 8..14
                const u32 PX = pperm(lt[1], lt[0], 0x0c0c0602u) | (pperm(lt[3], lt[2], 0x0c0c0602u) << 16);     // table bit 15: 1 = real cell
                // L_k: byte c = 0xff where cell c's table bit k is set, in TWO instructions: isolate the bit in every byte (values 0
                // or 1 << j) and let v_perm read it as a byte selector against (S0, S1) = (0x000000ff, 0x00ffff00): selector 0 picks
                // S1.byte0 = 0x00; 1, 2 pick S1.byte1/2 = 0xff; 4 picks S0.byte0 = 0xff; 8 replicates the sign of S1.byte1 = 0xff;
                // 16, 32, 64 (>= 13) are the constant 0xff
                u32 L[16];
                L[0] = 0u; L[15] = pperm(0x000000ffu, 0x00ffff00u, PX);
#pragma unroll
                for (int k = 1; k < 15; ++k) {
                    const int j = k <= 7 ? k - 1 : k - 8;
                    L[k] = pperm(0x000000ffu, 0x00ffff00u, (k <= 7 ? PL : PH) & (0x01010101u << j));
                }
                u32 g[8], h4[4];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = bsel(T[0], L[2 * j + 1], L[2 * j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = bsel(T[1], g[2 * j + 1], g[2 * j]);
                const u32 m0 = bsel(T[2], h4[1], h4[0]), m1 = bsel(T[2], h4[3], h4[2]);
                const u32 Gn = bsel(T[3], m1, m0);                                     // byte c = the 8 planes of cell 4 i + c
                G[0] = G[1]; G[1] = G[2]; G[2] = G[3]; G[3] = Gn;
            }
            const i64 x = x0 + 32 * qq + (tid >> 4);
            const i64 z = z0 + zc;
            if (x < W && z < D) {
                const u32 mbits = mask_wh ? mask8(mask_wh + x * H + yg, np) : 0xffu;
                const u32 mk = mbits * 0x01010101u;                                    // the row's plane mask in every byte
                const u32 A0 = G[0] & mk, A1 = G[1] & mk, A2 = G[2] & mk, A3 = G[3] & mk;
                const u32 ooff = (u32)(x * H * D + z);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q >= np) break;
                    u32x4 rr;
                    rr.x = (A0 >> q) & 0x01010101u; rr.y = (A1 >> q) & 0x01010101u; rr.z = (A2 >> q) & 0x01010101u; rr.w = (A3 >> q) & 0x01010101u;
                    u8* op = out + (yg + q) * D + ooff;
                    if (!ODD) *(u32x4*)op = rr;
                    else if (z + 16 <= D) *(u32x4a1*)op = rr;
                    else {                                                              // the row's last run: its own bytes only
                        const u32 t4[4] = {rr.x, rr.y, rr.z, rr.w};
                        const int nb = (int)(D - z);
                        for (int b = 0; b < nb; ++b) op[b] = (u8)(t4[b >> 2] >> (8 * (b & 3)));
                    }
                }
            }
            const u32x4 t0 = run[0];
#pragma unroll
            for (int k = 0; k < 7; ++k) run[k] = run[k + 1];
            run[7] = t0;
        }
    };
    for (i64 yg = y_beg; yg < y_end; yg += 8) {
        stage(yg);
        __syncthreads();
        evaluate(yg);
        __syncthreads();
    }
    if (hib & 0xfefefefeu) atomicOr(big_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// First rotation step of global_carve (reference utils/voxel_carving_utils.py:279-292) with its source SYNTHESISED: the grid that
// step sees is carve(ones, mask) -- every source voxel (s0, y, s2) equals bm[s0, y], whatever s2 -- so nothing is read from HBM but
// the cell table: the four taps of a cell are (b0, b0, b1, b1) with b0 = bm[s0, y], b1 = bm[s0 + 1, y], and SciPy's 16-entry result
// table collapses to three entries (tap patterns 0011, 1100, 1111).  The masks of 32 planes are held per x-row as one dword in LDS
// (bit q = bm[x, y0 + q]); a cell's 32 planes are three bitwise selects.  RGBOUT: the step is also the LAST one (angle_interval 46..90
// without 90, e.g. 60): the colours rgb[y, x] are written instead of the occupancy -- the whole global_carve is then write-only.
// One thread = one run of 16 cells along z (D % 16 == 0); 16 lanes cover 256 contiguous output bytes (768 with colours).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rgb_of_occ4(u32 occ01, u32 C0, u32 C1, u32 C2, u32* o) {     // 4 occupancy bytes (0/1) -> 12 colour bytes
    const u32 e = occ01 * 0xffu;                                                              // bytes 0x00 / 0xff (no carries)
    o[0] = pperm(e, e, 0x01000000u) & C0; o[1] = pperm(e, e, 0x02020101u) & C1; o[2] = pperm(e, e, 0x03030302u) & C2;
}

// bits[c * W + x]: bit q = mask_wh[x, 32 c + q] != 0 -- the (W,H) mask as one dword per x-row and chunk of 32 planes
__global__ __launch_bounds__(256) void k_mask_planebits(const u8* __restrict__ mask_wh, i64 W, i64 H, u32* __restrict__ bits) {
    const i64 nch = (H + 31) / 32;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nch * W; i += (i64)gridDim.x * blockDim.x) {
        const i64 c = i / W, x = i - c * W, y0 = 32 * c;
        u32 b = 0;
        for (int q = 0; q < 32 && y0 + q < H; ++q) b |= (u32)(mask_wh[x * H + y0 + q] != 0) << q;
        bits[i] = b;
    }
}

template <bool RGBOUT>
__global__ __launch_bounds__(256) void k_first_step(const CellRec* __restrict__ cells, const u32* __restrict__ planebits, const u8* __restrict__ rgb_hw3,
                                                    i64 W, i64 H, i64 D, u8* __restrict__ out) {
    extern __shared__ u32 mb[];                         // W dwords: bit q = mask_wh[x, y0 + q]
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const i64 y0 = (i64)blockIdx.y * 32;
    const int np = (int)(H - y0 < 32 ? H - y0 : 32);
    for (i64 x = threadIdx.x; x < W; x += 256) mb[x] = planebits[(i64)blockIdx.y * W + x];
    __syncthreads();
    // RGBOUT: a lane's 48 colour bytes go through a wave-private LDS window so that every store instruction writes the wave's
    // 16-byte chunks in address order (1 KiB contiguous per instruction where the lanes' runs follow each other; three stores of
    // 16 bytes every 48 touched each line three times)
    __shared__ u32x4 xch[RGBOUT ? 4 * 192 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const i64 nzr = D / 16;
    const i64 g0 = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool live = g0 < W * nzr;
    if (!RGBOUT && !live) return;
    const i64 g = live ? g0 : W * nzr - 1;               // a lane past the grid works on a copy of the last run and stores nothing
    const i64 x = g / nzr, z0 = 16 * (g - x * nzr);
    const u32 dst = mb[x];
    u32 bits[16];
    const u32x4* cp = (const u32x4*)(cells + x * D + z0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u32x4 v = cp[k];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 src = h ? v.z : v.x, lut = h ? v.w : v.y;
            u32 r = 0;
            if (src != 0xffffffffu) {
                const u32 s0 = src >> 16;
                const u32 B0 = mb[s0], B1 = s0 + 1 < (u32)W ? mb[s0 + 1] : 0u;       // beyond the grid the tap has weight 0: the table ignores it
                const u32 E1 = (u32)__builtin_amdgcn_sbfe((int)lut, 3, 1), E2 = (u32)__builtin_amdgcn_sbfe((int)lut, 12, 1),
                          E3 = (u32)__builtin_amdgcn_sbfe((int)lut, 15, 1);
                r = ((B0 & ~B1 & E1) | (~B0 & B1 & E2) | (B0 & B1 & E3)) & dst;
            }
            bits[2 * k + h] = r;
        }
    }
    const u32 ooff = (u32)(x * H * D + z0);
#pragma unroll 1
    for (int gq = 0; gq < 4; ++gq) {                    // 8 planes at a time: byte c of G[i] = planes 8 gq .. 8 gq + 7 of cell 4 i + c
        if (8 * gq >= np) break;
        u32 G[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 a = (bits[4 * i] >> (8 * gq)) & 0xffu, b = (bits[4 * i + 1] >> (8 * gq)) & 0xffu, c = (bits[4 * i + 2] >> (8 * gq)) & 0xffu,
                      d = (bits[4 * i + 3] >> (8 * gq)) & 0xffu;
            G[i] = a | (b << 8) | (c << 16) | (d << 24);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const i64 y = y0 + 8 * gq + q;
            if (8 * gq + q >= np) break;
            u32 oc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) oc[i] = (G[i] >> q) & 0x01010101u;
            if (!RGBOUT) {
                u32x4 r; r.x = oc[0]; r.y = oc[1]; r.z = oc[2]; r.w = oc[3];
                *(u32x4*)(out + y * D + ooff) = r;
            } else {
                const u8* px = rgb_hw3 + (y * W + x) * 3;
                const u32 R = px[0], Gc = px[1], B = px[2];
                const u32 C0 = R | (Gc << 8) | (B << 16) | (R << 24), C1 = Gc | (B << 8) | (R << 16) | (Gc << 24), C2 = B | (R << 8) | (Gc << 16) | (B << 24);
                u32 w[12];
#pragma unroll
                for (int i = 0; i < 4; ++i) rgb_of_occ4(oc[i], C0, C1, C2, w + 3 * i);
#pragma unroll
                for (int k = 0; k < 3; ++k) { u32x4 r; r.x = w[4 * k]; r.y = w[4 * k + 1]; r.z = w[4 * k + 2]; r.w = w[4 * k + 3]; xch[wv * 192 + 3 * lane + k] = r; }
                const u32 myoff = live ? ooff : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int c = 64 * k + lane, L = c / 3, part = c - 3 * L;           // chunk c of the wave's 3 KiB belongs to lane L
                    const u32x4 v = xch[wv * 192 + c];
                    const u32 lo = (u32)__shfl((int)myoff, L);
                    if (lo != 0xffffffffu) *(u32x4*)(out + 3 * (y * D + (i64)lo) + 16 * part) = v;
                }
            }
        }
    }
}

bool is_zero(double v) { return v == 0.0; }

}  // namespace

// The table-driven (0/1 data) form of a generic-angle step; *flag is raised on the device when the data was not 0/1.
//   >= 32 planes: cell table once per step, then the 128 x 128 / 16-plane kernel on large grids or the 64 x 64 / 32-plane one;
//   fewer planes (or a volume of 4 GiB and more): the 8-plane kernel that evaluates its cells itself.
static bool table_step_takes_src_mask(i64 W, i64 H, i64 D) { return H >= 32 && W * H * D < (1ll << 32) - 64; }

// ---- the two table sets ------------------------------------------------------------------------------------------------------
struct TableSet { void *cells, *lutmap, *parts, *tinfo, *runs; };
static const int kSetSlots[2][5] = {{16, 31, 17, 18, 19}, {26, 27, 28, 29, 30}};   // (no slot the auxiliary stream writes is shared with main-stream-only code)

// which tile kernel a table-driven step uses: 1 packed 256-tiles, 2 wide 128-tiles, 3 64-tiles (the parity tests pin each of them on
// the same grids: ctx->tune_rotate_tile = 64 / 128 / 256)
// does the allocation that holds [p, p + bytes) extend at least 16 bytes further?  (one driver query; false when in doubt)
static bool has_read_slack(const void* p, size_t bytes) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return (const char*)p + bytes + 16 <= (const char*)base + size;
}

// odd_ok: rows that are not multiples of 16 bytes may take the packed kernel (the input allocation has 16 bytes of slack)
static int table_kind(const pb3d_ctx* ctx, i64 W, i64 H, i64 D, bool odd_ok) {
    const int pin = ctx->tune_rotate_tile;
    const i64 ptiles = ((D + PT - 1) / PT) * ((W + PT - 1) / PT);
    const bool packed_ok = (D % 16 == 0 || (odd_ok && D >= 16)) && W * H * D < (1ll << 32) - 64 && H >= 8;
    // measured (tools/m4bench.py, 45 degrees): 512^3 0.061 ms against 0.113 (64-tiles) / 0.139 (128-tiles); 512 x 278 x 512 0.053 / 0.070 / 0.125
    if (packed_ok && (pin ? pin == 256 : (W >= 256 && D >= 256 && ptiles * ((H + 7) / 8) >= 64))) return 1;
    const i64 xtiles = ((D + XT - 1) / XT) * ((W + XT - 1) / XT);
    if (pin ? pin == 128 : (W >= 256 && D >= 256 && xtiles * ((H + 63) / 64) >= (i64)ctx->cus * 2)) return 2;
    return 3;
}

static bool cache_hit(const pb3d_ctx* ctx, const pb3d_ctx::RotCache& rc, int kind, const RotParams& p, i64 W, i64 H, i64 D) {
    return ctx->tune_misc[4] != 1 && rc.kind == kind && rc.W == W && rc.H == H && rc.D == D && rc.gen == ctx->scratch_gen &&
           memcmp(rc.p, &p, sizeof(RotParams)) == 0;
}
static void cache_set(pb3d_ctx* ctx, pb3d_ctx::RotCache& rc, int kind, const RotParams& p, i64 W, i64 H, i64 D, void* cells) {
    rc.kind = kind; rc.W = W; rc.H = H; rc.D = D; rc.cells = cells; rc.gen = ctx->scratch_gen; rc.stamp = ++ctx->rot_stamp;
    static_assert(sizeof(rc.p) == sizeof(RotParams), "RotCache holds one RotParams");
    memcpy(rc.p, &p, sizeof(RotParams));
}
// the main stream is about to read or overwrite set k: a table build that may still run on the auxiliary stream comes first
static int join_aux(pb3d_ctx* ctx, int k) {
    pb3d_ctx::RotCache& rc = ctx->rot_cache[k];
    if (rc.pending_aux) { PB3D_HIP(hipStreamWaitEvent(ctx->stream, rc.ready, 0)); rc.pending_aux = false; }
    return PB3D_OK;
}
static int mark_used(pb3d_ctx* ctx, int k) {
    PB3D_HIP(hipEventRecord(ctx->rot_cache[k].used, ctx->stream));
    ctx->rot_cache[k].used_valid = true;
    return PB3D_OK;
}

static int packed_alloc(pb3d_ctx* ctx, int k, i64 W, i64 D, TableSet* t) {
    const i64 ptiles = ((D + PT - 1) / PT) * ((W + PT - 1) / PT);
    PB3D_TRY(pb3d_scratch(ctx, kSetSlots[k][0], (size_t)(W * D + XCELLS) * sizeof(CellRec), &t->cells));
    PB3D_TRY(pb3d_scratch(ctx, kSetSlots[k][1], 512 * sizeof(u32), &t->lutmap));
    PB3D_TRY(pb3d_scratch(ctx, kSetSlots[k][2], (size_t)ptiles * PPARTS * sizeof(PPart), &t->parts));
    PB3D_TRY(pb3d_scratch(ctx, kSetSlots[k][3], (size_t)ptiles * sizeof(PTile), &t->tinfo));
    PB3D_TRY(pb3d_scratch(ctx, kSetSlots[k][4], (size_t)(W * ((D + 15) / 16)) * sizeof(RunRec), &t->runs));
    return PB3D_OK;
}
// cells -> tile parts -> tile footprints -> run records of a packed step, queued on `st`
static int packed_build(pb3d_ctx* ctx, hipStream_t st, const TableSet& t, const RotParams& p, i64 W, i64 H, i64 D) {
    const i64 ptiles = ((D + PT - 1) / PT) * ((W + PT - 1) / PT);
    const int ntz = (int)((D + PT - 1) / PT);
    PB3D_HIP(hipMemsetAsync(t.lutmap, 0, 512 * sizeof(u32), st));
    hipLaunchKernelGGL(k_rot_cells, dim3((unsigned)((W * D + 1023) / 1024)), dim3(256), 0, st, p, W, D, (CellRec*)t.cells, (u32*)t.lutmap);
    hipLaunchKernelGGL(k_rot8_parts, dim3((unsigned)(ptiles * PPARTS)), dim3(PTHREADS), 0, st, (const CellRec*)t.cells, W, D, ntz, (PPart*)t.parts);
    hipLaunchKernelGGL(k_rot8_tiles, dim3((unsigned)ptiles), dim3(PTHREADS), 0, st, (const PPart*)t.parts, (const u32*)t.lutmap, H, D, (PTile*)t.tinfo);
    const int nruns = (int)((D + 15) / 16);
    hipLaunchKernelGGL(k_rot8_pack, dim3((unsigned)((W * nruns + 15) / 16)), dim3(256), 0, st, (const CellRec*)t.cells, (PTile*)t.tinfo, W, D, ntz, nruns,
                       (RunRec*)t.runs);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

static bool generic_step_is_tiled(const double M[9], i64 W, i64 H, i64 D) {
    const double ext0 = fabs(M[0]) + fabs(M[2]), ext2 = fabs(M[6]) + fabs(M[8]);
    return ext0 <= 1.45 && ext2 <= 1.45 && W < 65536 && D < 65536 && W * H * D >= (1ll << 21);
}

// Build the tables of a step that is going to run LATER on the auxiliary stream, while the main stream's kernels run: a chain of
// steps (process_voxel_grid with a small angle interval, the jobs of part_carve) then never waits for a table (55 us per step, as
// much as the step itself at 512 x 278 x 512).  Call it right after the current step has been launched: the set the current step
// reads is the newer one, the other set is rebuilt once the kernels that read IT have finished (event `used`).  No-op when the step
// is not a packed table step or its tables are already there.
int pb3d_prefetch_rotation(pb3d_ctx* ctx, i64 W, i64 H, i64 D, const double M[9], const double off[3]) {
    if (ctx->tune_misc[4] != 0 || W * H * D == 0) return PB3D_OK;                              // misc4 = 1: no reuse, 2: no prefetch
    // (odd D: the tables are built for the packed kernel; a step whose input turns out to have no read slack simply does not use them)
    if (!generic_step_is_tiled(M, W, H, D) || !table_step_takes_src_mask(W, H, D) || table_kind(ctx, W, H, D, true) != 1) return PB3D_OK;
    const RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    for (int k = 0; k < 2; ++k)
        if (cache_hit(ctx, ctx->rot_cache[k], 1, p, W, H, D)) return PB3D_OK;
    const int k = ctx->rot_cache[0].stamp <= ctx->rot_cache[1].stamp ? 0 : 1;
    pb3d_ctx::RotCache& rc = ctx->rot_cache[k];
    rc.kind = 0;
    TableSet t;
    PB3D_TRY(packed_alloc(ctx, k, W, D, &t));
    if (rc.used_valid) PB3D_HIP(hipStreamWaitEvent(ctx->aux_stream, rc.used, 0));             // readers of the old tables first
    PB3D_TRY(packed_build(ctx, ctx->aux_stream, t, p, W, H, D));
    PB3D_HIP(hipEventRecord(rc.ready, ctx->aux_stream));
    rc.pending_aux = true;
    cache_set(ctx, rc, 1, p, W, H, D, t.cells);
    return PB3D_OK;
}

static int launch_table_step(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const RotParams& p, const u8* d_mask_wh, u8* d_out,
                             int* flag, const u8* d_mask_src) {
    const i64 tiles = ((D + LT - 1) / LT) * ((W + LT - 1) / LT);
    if (!table_step_takes_src_mask(W, H, D)) {
        PB3D_REQUIRE(d_mask_src == nullptr, "pb3d_rotate_carve: this step cannot fold a source mask");
        int TYL = 32;
        while (TYL > 1 && tiles * ((H + TYL - 1) / TYL) < (i64)ctx->cus * 8) TYL >>= 1;
        dim3 lgrid((unsigned)((D + LT - 1) / LT), (unsigned)((W + LT - 1) / LT), (unsigned)((H + TYL - 1) / TYL));
        PB3D_REQUIRE(lgrid.y <= 65535u && lgrid.z <= 65535u, "pb3d_rotate_carve: grid too large");
        if (D % 16 == 0 && ((((uintptr_t)d_in) | ((uintptr_t)d_out)) & 15u) == 0)
            hipLaunchKernelGGL(k_rotate_bits<true>, lgrid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TYL, flag);
        else
            hipLaunchKernelGGL(k_rotate_bits<false>, lgrid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TYL, flag);
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    const i64 ptiles = ((D + PT - 1) / PT) * ((W + PT - 1) / PT);
    const i64 xtiles = ((D + XT - 1) / XT) * ((W + XT - 1) / XT);
    // The step tables (cells, tile footprints, run records) depend on (matrix, offset, W, H, D) only.  A caller that repeats a step --
    // part_carve jobs with one angle, the same process_voxel_grid on grid after grid -- finds them where an earlier call left them
    // (two sets of private scratch slots; the stream is in order); a chain of different steps finds them where
    // pb3d_prefetch_rotation built them meanwhile.  tune misc4 = 1 switches the reuse off, 2 the prefetch.
    const bool odd = D % 16 != 0;
    const int kind = table_kind(ctx, W, H, D, odd && ctx->tune_misc[5] != 3 && has_read_slack(d_in, (size_t)(W * H * D)));
    const bool packed = kind == 1, wide = kind == 2;
    if (packed) {
        if (!ctx->packed_lds_set) {
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits8p<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPLds));
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits8p<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPLds));
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits8p<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPLds));
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits8p<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPLds));
            ctx->packed_lds_set = true;
        }
        int k = -1;
        for (int q = 0; q < 2; ++q)
            if (cache_hit(ctx, ctx->rot_cache[q], 1, p, W, H, D)) k = q;
        const bool cached = k >= 0;
        if (!cached) k = ctx->rot_cache[0].stamp <= ctx->rot_cache[1].stamp ? 0 : 1;
        pb3d_ctx::RotCache& rc = ctx->rot_cache[k];
        PB3D_TRY(join_aux(ctx, k));
        TableSet t;
        if (!cached) rc.kind = 0;                           // invalid until this call has queued everything
        PB3D_TRY(packed_alloc(ctx, k, W, D, &t));
        if (!cached) PB3D_TRY(packed_build(ctx, ctx->stream, t, p, W, H, D));
        const int ntz = (int)((D + PT - 1) / PT);
        // planes per workgroup: 8 = one pass (measured at 1024^3, 45 degrees, warm tables: 8 planes 0.46 ms, 16 / 32 planes 0.50 --
        // the more workgroups, the better their stage and evaluate phases interleave across the chip)
        int TYP = ctx->tune_rot8_ty > 0 ? ctx->tune_rot8_ty : 8;
        const int nchunks = (int)((H + TYP - 1) / TYP);
        // fewer (tile, chunk) pairs than two per CU: split the run slots of a pair over 2 or 4 workgroups (tune misc5 = 4: never)
        int nsplit = 1;
        if (ctx->tune_misc[5] != 4)
            while (nsplit < 4 && ptiles * nchunks * nsplit * 2 <= (i64)ctx->cus * 2) nsplit *= 2;
        const i64 nblk = 8 * ptiles * nsplit * ((nchunks + 7) / 8);
        PB3D_REQUIRE(nblk < (1ll << 31), "pb3d_rotate_carve: grid too large");
        auto kern = odd ? (d_mask_src ? k_rotate_bits8p<true, true> : k_rotate_bits8p<false, true>)
                        : (d_mask_src ? k_rotate_bits8p<true, false> : k_rotate_bits8p<false, false>);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(PTHREADS), kPLds, ctx->stream, d_in, d_out, d_mask_wh, (const RunRec*)t.runs,
                           (const PTile*)t.tinfo, W, H, D, TYP, ntz, (int)ptiles, nchunks, flag, d_mask_src, ctx->tune_misc[0], (int)((D + 15) / 16), nsplit);
        PB3D_CHECK_LAUNCH();
        cache_set(ctx, rc, 1, p, W, H, D, t.cells);
        PB3D_TRY(mark_used(ctx, k));
        return PB3D_OK;
    }
    // the other two tile kernels keep their cells (and tile rows) in the slots of set 0
    pb3d_ctx::RotCache& rc = ctx->rot_cache[0];
    PB3D_TRY(join_aux(ctx, 0));
    void* cells;
    PB3D_TRY(pb3d_scratch(ctx, 16, (size_t)(W * D + XCELLS) * sizeof(CellRec), &cells));
    const bool cached = cache_hit(ctx, rc, kind, p, W, H, D) && rc.cells == cells;
    rc.kind = 0;                                            // invalid until this call has queued everything
    if (!cached) {
        hipLaunchKernelGGL(k_rot_cells, dim3((unsigned)((W * D + 1023) / 1024)), dim3(256), 0, ctx->stream, p, W, D, (CellRec*)cells, (u32*)nullptr);
        PB3D_CHECK_LAUNCH();
    }
    auto remember = [&]() { cache_set(ctx, rc, kind, p, W, H, D, cells); return mark_used(ctx, 0); };
    if (wide) {
        if (!ctx->wide_lds_set) {       // > 64 KiB of LDS per workgroup has to be allowed once per device
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits16w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXLds));
            PB3D_HIP(hipFuncSetAttribute((const void*)k_rotate_bits16w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXLds));
            ctx->wide_lds_set = true;
        }
        void* trows;
        PB3D_TRY(pb3d_scratch(ctx, 17, (size_t)xtiles * XQ * sizeof(TileRows), &trows));
        if (!cached) {
            hipLaunchKernelGGL(k_rot_tile_rows, dim3((unsigned)(xtiles * XQ)), dim3(XTHREADS), 0, ctx->stream, (const CellRec*)cells, W, D,
                               (int)((D + XT - 1) / XT), (TileRows*)trows);
            PB3D_CHECK_LAUNCH();
        }
        const int TYX = 64;
        const i64 nblkx = 8 * ((xtiles + 7) / 8) * ((H + TYX - 1) / TYX);
        PB3D_REQUIRE(nblkx < (1ll << 31), "pb3d_rotate_carve: grid too large");
        auto kern = D % XCELLS == 0 ? k_rotate_bits16w<false> : k_rotate_bits16w<true>;     // a thread's run is XCELLS = 32 voxels
        hipLaunchKernelGGL(kern, dim3((unsigned)nblkx), dim3(XTHREADS), kXLds, ctx->stream, d_in, d_out, d_mask_wh, (const CellRec*)cells,
                           (const TileRows*)trows, W, H, D, TYX, (int)((D + XT - 1) / XT), (int)xtiles, flag, d_mask_src);
    } else {
        int TYW = 64;        // multiples of 32 planes per workgroup (measured at 1024^3: 32/64 planes 0.76 ms, 128: 0.79, 256: 0.88)
        while (TYW > 32 && tiles * ((H + TYW - 1) / TYW) < (i64)ctx->cus * 4) TYW >>= 1;
        const i64 nblk = 8 * ((tiles + 7) / 8) * ((H + TYW - 1) / TYW);
        PB3D_REQUIRE(nblk < (1ll << 31), "pb3d_rotate_carve: grid too large");
        auto kern = D % 16 == 0 ? k_rotate_bits32<false> : k_rotate_bits32<true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WTHREADS), 0, ctx->stream, d_in, d_out, d_mask_wh, (const CellRec*)cells, W, H, D,
                           TYW, (int)((D + LT - 1) / LT), (int)tiles, flag, d_mask_src);
    }
    PB3D_CHECK_LAUNCH();
    PB3D_TRY(remember());
    return PB3D_OK;
}

// global_carve's first rotation step (generic angle) from the mask alone; rgb != NULL: the step is also the last one and writes colours.
// PB3D_EUNSUPPORTED (no message) when the shape does not suit the kernel: the caller runs the composed pipeline.
int pb3d_launch_first_step(pb3d_ctx* ctx, i64 W, i64 H, i64 D, const double M[9], const double off[3], const u8* d_mask_wh, const u8* d_rgb_hw3,
                           u8* d_out) {
    if (D % 16 != 0 || W > 16384 || W * H * D >= (1ll << 32) - 64 || (((uintptr_t)d_out) & 15u)) return PB3D_EUNSUPPORTED;
    if (!(is_zero(M[3]) && M[4] == 1.0 && is_zero(M[5]) && is_zero(M[1]) && is_zero(M[7]) && is_zero(off[1]))) return PB3D_EUNSUPPORTED;
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    void* cells;
    PB3D_TRY(pb3d_scratch(ctx, 16, (size_t)(W * D + XCELLS) * sizeof(CellRec), &cells));
    PB3D_TRY(join_aux(ctx, 0));
    ctx->rot_cache[0].kind = 0;                         // slots 16 / 17 are about to hold another step's cells
    hipLaunchKernelGGL(k_rot_cells, dim3((unsigned)((W * D + 1023) / 1024)), dim3(256), 0, ctx->stream, p, W, D, (CellRec*)cells, (u32*)nullptr);
    PB3D_CHECK_LAUNCH();
    dim3 grid((unsigned)((W * (D / 16) + 255) / 256), (unsigned)((H + 31) / 32));
    PB3D_REQUIRE(grid.y <= 65535u, "pb3d_global_carve: grid too large");
    void* pbits;
    PB3D_TRY(pb3d_scratch(ctx, 17, (size_t)(W * ((H + 31) / 32)) * sizeof(u32), &pbits));
    hipLaunchKernelGGL(k_mask_planebits, dim3(pb3d_stream_blocks(ctx, W * ((H + 31) / 32), 256, 8)), dim3(256), 0, ctx->stream, d_mask_wh, W, H, (u32*)pbits);
    PB3D_CHECK_LAUNCH();
    if (d_rgb_hw3) hipLaunchKernelGGL(k_first_step<true>, grid, dim3(256), (size_t)W * 4, ctx->stream, (const CellRec*)cells, (const u32*)pbits, d_rgb_hw3, W, H, D, d_out);
    else hipLaunchKernelGGL(k_first_step<false>, grid, dim3(256), (size_t)W * 4, ctx->stream, (const CellRec*)cells, (const u32*)pbits, d_rgb_hw3, W, H, D, d_out);
    PB3D_CHECK_LAUNCH();
    // these kernels read slots 16 / 17 on the main stream: a later prefetch on the auxiliary stream must see them as the set's last
    // readers and must not pick set 0 as the "older" one while they run
    PB3D_TRY(mark_used(ctx, 0));
    ctx->rot_cache[0].stamp = ++ctx->rot_stamp;
    return PB3D_OK;
}

bool pb3d_generic_step_takes_src_mask(const double M[9], i64 W, i64 H, i64 D) {
    return generic_step_is_tiled(M, W, H, D) && table_step_takes_src_mask(W, H, D);
}

int pb3d_prefetch_first_step(pb3d_ctx* ctx, i64 W, i64 H, i64 D, int angle_interval) {
    if (angle_interval <= 0 || angle_interval > 90 || W * H * D == 0) return PB3D_OK;
    const i64 shape[3] = {W, H, D};
    double M[9], off[3];
    PB3D_TRY(pb3d_rotinv(angle_interval, M));
    PB3D_TRY(pb3d_offset(M, shape, off));
    if (pb3d_is_perm_step(M, off, W, D)) return PB3D_OK;
    return pb3d_prefetch_rotation(ctx, W, H, D, M, off);
}

int pb3d_launch_rotate_generic(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const double M[9],
                               const double off[3], const u8* d_mask_wh, u8* d_out, const u8* d_mask_src) {
    PB3D_REQUIRE(is_zero(M[3]) && M[4] == 1.0 && is_zero(M[5]) && is_zero(M[1]) && is_zero(M[7]) && is_zero(off[1]),
                 "pb3d_rotate_carve: matrix is not a rotation about Y (row 1 must be [0,1,0], M[0][1]=M[2][1]=0, off[1]=0)");
    PB3D_REQUIRE(W < (1ll << 30) && D < (1ll << 30), "pb3d_rotate_carve: axis too long");
    if (W * H * D == 0) return PB3D_OK;
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    // Large grids: table-driven tiled kernel first; the arithmetic kernel follows and runs only if the
    // first one met a value > 1 (device-side flag, no host round trip).  Small grids (launch-bound):
    // the arithmetic kernel alone.  The footprint of a 64 x 64 tile must fit the LDS box: true for
    // rotations (row sums of |M| <= sqrt 2).
    const bool tiled = generic_step_is_tiled(M, W, H, D);
    int *flag = nullptr, *flag_clear = nullptr;
    if (tiled) {
        void* f;
        PB3D_TRY(pb3d_scratch(ctx, 15, 64, &f));
        // The "a value > 1 was seen" flag of step g is word g % 16 of a ring; the conditional second pass of step g clears the word
        // of step g + 8 (no kernel in flight uses it: the stream is in order), so a step needs no memset of its own.  The ring is
        // zeroed when the slot is new (another user of the slot may have grown it).
        if (ctx->flag_ring != f) { PB3D_HIP(hipMemsetAsync(f, 0, 64, ctx->stream)); ctx->flag_ring = f; ctx->flag_gen = 0; }
        flag = (int*)f + (ctx->flag_gen & 15);
        flag_clear = (int*)f + ((ctx->flag_gen + 8) & 15);
        ++ctx->flag_gen;
        PB3D_TRY(launch_table_step(ctx, d_in, W, H, D, p, d_mask_wh, d_out, flag, d_mask_src));
    }
    int TY = tiled ? 64 : 16;      // after a table-driven step the grid only reads the flag: keep that launch small
    // keep at least ~8 blocks per CU in flight for small grids
    const i64 tiles_xz = ((D + 255) / 256) * ((W + 3) / 4);
    while (TY > 1 && tiles_xz * ((H + TY - 1) / TY) < (i64)ctx->cus * 8) TY >>= 1;
    const unsigned gx = (unsigned)((D + 255) / 256), gy = (unsigned)((W + 3) / 4), gz = (unsigned)((H + TY - 1) / TY);
    const u64 nblk = (u64)gx * gy * gz;
    // after a table-driven step this launch normally only reads the flag: a few workgroups per CU walk the block space if it runs
    const u64 cap = tiled ? (u64)ctx->cus * 4 : 0x7fffffffull;
    const unsigned launch = (unsigned)(nblk < cap ? nblk : cap);
    const bool pack = (D % 4 == 0) && (((uintptr_t)d_out & 3u) == 0);
    if (pack)
        hipLaunchKernelGGL(k_rotate_generic<true>, dim3(launch), dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TY, flag, d_mask_src, gx, gy, gz, flag_clear);
    else
        hipLaunchKernelGGL(k_rotate_generic<false>, dim3(launch), dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TY, flag, d_mask_src, gx, gy, gz, flag_clear);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

static int process_grid_impl(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                             int angle_interval, uint8_t* d_out, uint8_t* d_tmp, int known_binary);

extern "C" {

int pb3d_rotate_carve_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const double M[9], const double off[3], const uint8_t* d_mask_wh, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && M && off, "pb3d_rotate_carve: null argument");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_rotate_carve: bad shape");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_occ && d_out && d_occ != d_out, "pb3d_rotate_carve: null or aliased buffer");
    const bool row1 = M[3] == 0.0 && M[4] == 1.0 && M[5] == 0.0 && M[1] == 0.0 && M[7] == 0.0 && off[1] == 0.0;
    if (row1 && pb3d_perm_step_ok(M, off, W, D, d_occ, d_out))
        return pb3d_launch_rotate_perm(ctx, d_occ, W, H, D, M, off, nullptr, d_mask_wh, d_out);
    return pb3d_launch_rotate_generic(ctx, d_occ, W, H, D, M, off, d_mask_wh, d_out, nullptr);
}

int pb3d_process_grid_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const uint8_t* d_mask_wh, int angle_interval, uint8_t* d_out, uint8_t* d_tmp) {
    return process_grid_impl(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, d_tmp, 0);
}

}  // extern "C"

int pb3d_process_grid_binary_dev(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out,
                                 u8* d_tmp) {
    return process_grid_impl(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, d_tmp, 1);
}

static int process_grid_impl(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                             const uint8_t* d_mask_wh, int angle_interval, uint8_t* d_out, uint8_t* d_tmp, int known_binary) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_process_grid: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_process_grid: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_process_grid: angle_interval must be a positive integer (got %d)", angle_interval);
    const i64 nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_occ && d_mask_wh && d_out && d_tmp, "pb3d_process_grid: null buffer");
    PB3D_REQUIRE(d_out != d_occ && d_tmp != d_occ && d_tmp != d_out, "pb3d_process_grid: buffers must not alias");
    const int nsteps = 90 / angle_interval + 1;  // len(range(0, 91, k))
    const i64 shape[3] = {W, H, D};
    // Step 0 is Rinv(0) = I with zero offset: coordinates are the integers themselves, weights {1,0},
    // acc == v and (uint8)(v + 0.5) == v, i.e. the step is the mask carve alone.
    double M0[9], off0[3];
    PB3D_TRY(pb3d_rotinv(0, M0));
    PB3D_TRY(pb3d_offset(M0, shape, off0));
    bool ident = off0[0] == 0.0 && off0[1] == 0.0 && off0[2] == 0.0;
    for (int k = 0; k < 9; ++k) ident = ident && M0[k] == ((k % 4 == 0) ? 1.0 : 0.0);
    PB3D_REQUIRE(ident, "pb3d_process_grid: internal error, Rinv(0) is not the identity");
    if (nsteps == 1) return pb3d_carve_mask_dev(ctx, d_occ, W, H, D, 1, d_mask_wh, d_out);
    {   // chains of rotation steps stay bit-sliced between the steps (csrc/sliced.hip); data that is not 0 / 1 comes back here
        int took = 0;
        PB3D_TRY(pb3d_process_grid_sliced(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, known_binary, &took));
        if (took) return PB3D_OK;
    }
    double M1[9], off1[3];
    PB3D_TRY(pb3d_rotinv(angle_interval, M1));
    PB3D_TRY(pb3d_offset(M1, shape, off1));
    // The second step takes the 0-degree carve as its SOURCE mask (one pass over the volume less): a permutation-like step
    // (90 degrees, W + D even) always, a generic-angle step when it runs through the cell-table kernels.
    const bool perm1 = pb3d_perm_step_ok(M1, off1, W, D, d_occ, d_out) && pb3d_perm_step_ok(M1, off1, W, D, d_occ, d_tmp);
    const bool fuse_first = perm1 || (!pb3d_is_perm_step(M1, off1, W, D) && pb3d_generic_step_takes_src_mask(M1, W, H, D));
    const int nlaunch = fuse_first ? nsteps - 1 : nsteps;
    const u8* src = d_occ;
    int li = 0;
    for (int s = fuse_first ? 1 : 0; s < nsteps; ++s, ++li) {
        u8* dst = ((nlaunch - 1 - li) % 2 == 0) ? d_out : d_tmp;  // ping-pong so that the last step lands in d_out
        if (s == 0) {
            PB3D_TRY(pb3d_carve_mask_dev(ctx, src, W, H, D, 1, d_mask_wh, dst));
        } else {
            double M[9], off[3];
            PB3D_TRY(pb3d_rotinv(s * angle_interval, M));
            PB3D_TRY(pb3d_offset(M, shape, off));
            if (pb3d_perm_step_ok(M, off, W, D, src, dst))
                PB3D_TRY(pb3d_launch_rotate_perm(ctx, src, W, H, D, M, off, (fuse_first && s == 1) ? d_mask_wh : nullptr, d_mask_wh, dst));
            else
                PB3D_TRY(pb3d_launch_rotate_generic(ctx, src, W, H, D, M, off, d_mask_wh, dst, (fuse_first && s == 1) ? d_mask_wh : nullptr));
        }
        if (s + 1 < nsteps) {                               // the next step's tables are built while this step's kernel runs
            double Mn[9], offn[3];
            PB3D_TRY(pb3d_rotinv((s + 1) * angle_interval, Mn));
            PB3D_TRY(pb3d_offset(Mn, shape, offn));
            if (!pb3d_is_perm_step(Mn, offn, W, D)) PB3D_TRY(pb3d_prefetch_rotation(ctx, W, H, D, Mn, offn));
        }
        src = dst;
    }
    return PB3D_OK;
}
