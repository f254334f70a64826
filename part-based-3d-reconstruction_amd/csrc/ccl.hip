// N1: 3-D connected components (6-connectivity) of up to eight colours in ONE labelling sequence -- scipy.ndimage.label(mask) with
// the default structure, as called at reference utils/voxel_carving_utils.py:175 (once per part colour from :338) and :254:
// the components of a colour are numbered in raster order of their first voxel.
//
// Round-4 form (round 2 built the bit-mask / row-frame design; SQ counters of that version, profiles/r04_ccl_sq_counters_before.txt:
// k_ccl_init issued 3.1e8 vector + 3.9e8 scalar instructions per 1024^3 launch -- its lane-per-voxel forest initialisation, a serial
// walk over the sixteen windows of a chunk, was the kernel; k_ccl_finish the same in the other direction; the statistics ran on 2048
// waves that waited 67 % of their cycles).  Now:
//   * the colour grid is read ONCE for all requested colours (k_ccl_init): membership is one bit per voxel and colour, in row-padded
//     bit arrays (word k * nwords + row * P + t, bits past A2 zero) -- a run = a maximal run of ONE colour, so the colours never
//     interact and one forest in the caller's int32 label array serves them all;
//   * union-find nodes are WINDOW SEGMENTS: the maximal runs of a colour inside a 64-voxel window, named by their first voxel.  A
//     segment that continues a run from the window before is linked to that window's last segment by the init pass itself.  Every
//     later pass is therefore local to a window: no carries along a row, no serial loop, and the forest is touched at segment
//     starts only (4 B per segment instead of 4 B per member voxel, twice);
//   * links along the two slow axes are made at the first voxel of every overlap of two runs (k_ccl_merge), the two root searches
//     of a union advance together (two loads in flight per step: the pass is bound by dependent-load latency);
//   * the smaller index always becomes the root, so root == first voxel of the component in raster order and label == rank of the
//     root among the roots OF ITS COLOUR (k_ccl_roots / k_ccl_scan / k_ccl_number: popcount scans per colour);
//   * the last pass (k_ccl_finish) resolves one walk per segment -- a lane per window, 64 windows in flight per wave -- and writes
//     the labels as 16-byte vectors, four voxels per lane; it gathers the per-component statistics (bounding box, count, coordinate
//     sums: closed form per segment) on the way, in registers per lane, then per block in LDS, then into one of sixteen shadow copies
//     of the record table (same-address global atomics serialise across the chip); k_ccl_fold reduces the copies.
// While the forest is being built a segment start holds ~parent (negative), a finished voxel holds its label (positive), so a walker
// that meets a positive value has met its answer; non-member voxels end as 0 (members_only: they are not written at all).
// HBM traffic at 1024^3: C B/voxel read once; full labels: 4 B/voxel written once; everything else is bit masks.
#include <vector>

#include "pb3d_internal.h"

namespace {

constexpr int kMaxColors = PB3D_CCL_MAX_COLORS;
struct CclColors { u32 c[kMaxColors]; };

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- forest on complemented indices: parent[v] == ~v at a root -------------------------------------------------------------
// Union with both root searches advancing together (path halving on each side; every store is an ancestor, so the forest stays valid
// under concurrent unions).
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        int pa = ~ld(&parent[a]), pb = ~ld(&parent[b]);
        while (pa != a || pb != b) {
            const int ga = pa != a ? ~ld(&parent[pa]) : pa;
            const int gb = pb != b ? ~ld(&parent[pb]) : pb;
            if (pa != a) { if (ga != pa) st(&parent[a], ~ga); a = pa; pa = ga; }
            if (pb != b) { if (gb != pb) st(&parent[b], ~gb); b = pb; pb = gb; }
        }
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }        // the larger root goes under the smaller: ~b > ~a
        const int old = atomicMax(&parent[a], ~b);
        if (old == ~a) return;
        a = ~old;                                            // someone re-parented a meanwhile: retry from there
    }
}

__device__ __forceinline__ u64 le_mask(int i) { return i >= 63 ? ~0ull : ((2ull << i) - 1ull); }     // bits 0 .. i
__device__ __forceinline__ int hi_bit(u64 x) { return 63 - __clzll((long long)x); }                  // x != 0
typedef u32 u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef int i32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

// ---- membership bits of all colours + the segment forest: a wavefront per 1024 voxels of whole rows ---------------------------------
// A wave takes R = 2^lgR whole rows at a time when a row fits 1024 / R voxels (R = 1: a row in chunks of 1024 voxels): lane l reads
// 16 voxels of its row (16-byte loads at whatever alignment the row has) and compares them with every colour; lanes 0..15 collect the
// wave's sixteen 64-bit windows per colour with cross-lane reads (window l = the masks of lanes 4 l .. 4 l + 3, whichever row that
// is), store them, and initialise the forest at their windows' segment starts.  Rows of 512 voxels (Taj at max_dim 512) left half
// of the lanes idle in the one-row-per-wave form, and the pass is bound by vector issue once several colours are compared.
// C = 1: the grid is a 1-byte LABEL volume (row N3) and the colours are label values.  The shadow statistics records of the last
// pass are cleared on the way (nrec > 0).
constexpr int kChunkVox = 1024;

template <int C, int KT>
__global__ __launch_bounds__(256) void k_ccl_init(const u8* __restrict__ grid, i64 rows, int A2, int P, int K, CclColors cols, i64 nwords,
                                                  u64* __restrict__ bits, int* __restrict__ parent, int lgR, int zero_is_colour, i64 nrec,
                                                  char* __restrict__ recs) {
    if (nrec > 0) {         // every shadow record: lo = +inf, hi = -1, count and sums 0
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < nrec * 4; q += (i64)gridDim.x * blockDim.x) {
            const int part = (int)(q & 3);
            i32x4 v;
            if (part == 0) { v.x = v.y = v.z = 0x7fffffff; v.w = -1; }
            else if (part == 1) { v.x = v.y = -1; v.z = v.w = 0; }
            else { v.x = v.y = v.z = v.w = 0; }
            ((i32x4*)recs)[q] = v;
        }
    }
    const int lane = threadIdx.x & 63;
    const i64 nbytes = C * rows * (i64)A2;
    const int R = 1 << lgR;
    const int lr = lane & ((64 >> lgR) - 1);                   // this lane's 16 voxels: 16 lr .. of row rw0 + lane / (64 / R)
    const int tw = lane & ((16 >> lgR) - 1);                   // lanes 0..15: window tw of row rw0 + lane / (16 / R)
    constexpr int NW = C == 1 ? 4 : 12;                        // dwords a lane loads per task
    // A task = 1024 voxels: R whole rows, or one chunk of a row longer than that (its chunks in order: run carries).  One task per wave
    // and as many workgroups as there are tasks: a persistent grid that requests the NEXT task's voxels before comparing this one's was
    // measured SLOWER (692 against 615 us at 1024^3, 54 against 48 us for four colours at Taj 512) -- as for the headline carve kernel,
    // the dispatcher's many small workgroups hide the latency better than a software pipeline does.
    auto load_task = [&](i64 rw0, int c0, u32* w) {
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = 0u;
        const i64 vrow = rw0 + (lane >> (6 - lgR));
        const int v = c0 + 16 * lr;
        if (vrow >= rows || v >= A2) return;
        const i64 boff = C * (vrow * (i64)A2 + v);
        if (boff + 4 * NW <= nbytes) {                               // (past the row's end these are the next row's voxels: masked off below)
            const u32x4a1* p = (const u32x4a1*)(grid + boff);
#pragma unroll
            for (int k = 0; k < NW / 4; ++k) { const u32x4a1 t = p[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
        } else {                                                     // the grid's last bytes: byte by byte
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                u32 t = 0;
                for (int b = 0; b < 4; ++b) { const i64 o = boff + 4 * k + b; if (o < nbytes) t |= (u32)grid[o] << (8 * b); }
                w[k] = t;
            }
        }
    };
    u32 wcur[NW + 1];
    wcur[NW] = 0u;
    u32 carry_prev[KT], carry_node[KT];
    for (i64 rw0 = ((i64)blockIdx.x * 4 + (threadIdx.x >> 6)) * R; rw0 < rows; rw0 += (i64)gridDim.x * 4 * R) {
#pragma unroll
        for (int k = 0; k < KT; ++k) { carry_prev[k] = 0; carry_node[k] = 0; }
        for (int c0 = 0; c0 < A2; c0 += kChunkVox) {
            load_task(rw0, c0, wcur);
        {
            const i64 vrow = rw0 + (lane >> (6 - lgR));
            const i64 wrow = rw0 + ((lane & 15) >> (4 - lgR));
            // ---- the 16 voxels of this lane as 24-bit (8-bit) values
            const int v = c0 + 16 * lr;
            const bool have = vrow < rows && v < A2;
            u32 vox[16];
            u32 nz = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) nz |= wcur[k];
            if (C == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) vox[i] = have ? (wcur[i >> 2] >> (8 * (i & 3))) & 0xffu : 0xffffffffu;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int j = (3 * i) >> 2, sh = (3 * i) & 3;
                    vox[i] = have ? __builtin_amdgcn_alignbyte(wcur[j + 1], wcur[j], (u32)sh) & 0x00ffffffu : 0xffffffffu;      // (no voxel: no colour)
                }
            }
            const u32 tail = have ? (v + 16 > A2 ? (1u << (A2 - v)) - 1u : 0xffffu) : 0u;
            const int t = (c0 >> 6) + tw;                           // (window role, lanes 0..15)
            const bool wvalid = lane < 16 && wrow < rows && t < P;
            // carved grids are mostly empty: a wave whose 1024 voxels are all zero has no member (unless zero IS a requested colour)
            const bool empty = !zero_is_colour && __ballot(nz != 0u) == 0ull;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k >= K) break;
                if (empty) {
                    if (wvalid) bits[(i64)k * nwords + wrow * P + t] = 0ull;
                    carry_prev[k] = 0; carry_node[k] = 0;
                    continue;
                }
                const u32 ck = cols.c[k];
                // m16 bit i = (vox[i] == ck): built from the top bit down, m = m + m + (compare's carry) -- two instructions per voxel
                u32 m16 = 0;
#pragma unroll
                for (int i = 15; i >= 0; --i)
                    asm volatile("v_cmp_eq_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m16) : "v"(vox[i]), "s"(ck) : "vcc");
                m16 &= tail;
                // ---- the wave's windows of colour k, one per lane 0..15
                const u64 w = (u64)(u32)__shfl((int)m16, 4 * lane) | ((u64)(u32)__shfl((int)m16, 4 * lane + 1) << 16) |
                              ((u64)(u32)__shfl((int)m16, 4 * lane + 2) << 32) | ((u64)(u32)__shfl((int)m16, 4 * lane + 3) << 48);
                const u64 seg = w & ~(w << 1);                            // first voxel of every segment of this window
                const int hi = seg ? hi_bit(seg) : 0;                     // ... the last segment's
                const u32 top = (u32)(w >> 63);
                u32 prev = (u32)__shfl_up((int)top, 1);
                if (tw == 0) prev = lgR ? 0u : carry_prev[k];             // (a row's first window of this chunk)
                // A segment that continues the run of the window before hangs under that RUN's first segment -- not under the window
                // before it: linked window by window, a run of 1024 voxels is a chain sixteen nodes deep, which the last pass walked one
                // dependent load at a time.  The run's first segment sits in the nearest window below that the run does not pass
                // through (ft: the bit-0 segment continues AND is the window's last one): one ballot + one cross-lane read.
                const bool cont = prev && (w & 1ull);
                const bool ft = cont && hi == 0;
                const u32 below = (u32)__ballot(!ft) & 0xffffu & ((1u << (lane & 15)) - 1u);
                const u32 v0 = (u32)wrow * (u32)A2 + 64u * (u32)t;
                const u32 mynode = v0 + (u32)hi;                          // this window's last segment
                u32 runstart = (u32)__shfl((int)mynode, below ? 31 - __clz((int)below) : 0);
                if (!below) runstart = carry_node[k];                     // (the run came in from the chunk before: rows longer than 1024)
                if (wvalid) {
                    bits[(i64)k * nwords + wrow * P + t] = w;
                    u64 sg = seg;
                    while (sg) {
                        const int i = __ffsll((unsigned long long)sg) - 1;
                        sg &= sg - 1;
                        const u32 par = (i == 0 && cont) ? runstart : v0 + (u32)i;
                        parent[v0 + (u32)i] = ~(int)par;
                    }
                }
                carry_node[k] = (u32)__builtin_amdgcn_readlane((int)(ft ? runstart : mynode), 15);
                carry_prev[k] = (u32)__builtin_amdgcn_readlane((int)top, 15);
            }
        }
        }
    }
}

// ---- links along the two slow axes ---------------------------------------------------------------------------------------------
// A link (v, v + s) is made only at the FIRST voxel of every overlap of two runs (an overlap at bit 0 that continues one of the window
// before is the same overlap).  Tile form (rows of at most 2048 voxels): a workgroup stages 64 levels along the link direction (rows of a
// plane for DIR 0, planes for DIR 1) x up to 32 windows per level (+ one halo level) in LDS with coalesced loads; a wave then takes a
// window column, one lane per level.  Neighbouring levels whose windows hold ONE segment each and overlap -- every row pair inside a
// solid part -- form chains along the lanes, found with one ballot: every member is united with the chain's FIRST node instead of
// with its neighbour.  Linked pairwise by thousands of lanes at the same moment, a solid grew paths as deep as it is tall (the
// plinth of Taj 512: 40 rows x 512 planes), which the later searches -- and the last pass -- then walked one dependent load at a time;
// the pass is bound by exactly that latency (SQ counters: the waves wait 87 % of their cycles).
constexpr int kTileLev = 64, kTileCol = 64;

// DIR 1 also stages the row ABOVE its rows: a plane-to-plane link at row r is implied -- and skipped -- when somewhere along the overlap
// the row above has members in both planes too.  Only the top row of every vertical stack of a solid links across planes: 1 / height
// of the unions (at 1024^3 the pass wrote 35 MB of forest entries before, profiles/r04_ccl_merge_tile_counters.txt).
template <int DIR>
__global__ __launch_bounds__(256) void k_ccl_merge_tile(const u64* __restrict__ bits_all, i64 nwords, int A0, int A1, int A2, int P, int RT, int* parent) {
    extern __shared__ u64 tile_mem[];                                     // (kTileLev + 1) levels x pitch words
    const u64* __restrict__ bits = bits_all + (i64)blockIdx.z * nwords;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pitch = (DIR == 0 ? P : (RT + 1) * P) + 1;
#define tile(l, c) tile_mem[(l) * pitch + (c)]
    // level l of the tile <-> rows a1 = 64 bx + l of plane by (DIR 0) / planes a0 = 64 bx + l at rows RT by .. (DIR 1)
    const int lev0 = 64 * (int)blockIdx.x;
    const int nlevav = DIR == 0 ? (A1 - lev0 < kTileLev + 1 ? A1 - lev0 : kTileLev + 1) : (A0 - lev0 < kTileLev + 1 ? A0 - lev0 : kTileLev + 1);
    const int r0 = DIR == 0 ? 0 : RT * (int)blockIdx.y;                   // first row this block links
    const int rlo = DIR == 0 ? 0 : (r0 > 0 ? r0 - 1 : 0);                 // first row it stages
    const int rhi = DIR == 0 ? 1 : (r0 + RT < A1 ? r0 + RT : A1);
    const int ncol = DIR == 0 ? P : (rhi - rlo) * P;
    const int cfirst = DIR == 0 ? 0 : (r0 - rlo) * P;
    const i64 lev_stride = DIR == 0 ? (i64)P : (i64)A1 * P;               // words from a level to the next
    const i64 w00 = DIR == 0 ? ((i64)blockIdx.y * A1 + lev0) * P : ((i64)lev0 * A1 + rlo) * P;
    u32 any = 0;
    // (eight loads in flight per thread: issued one per loop iteration the staging alone took eight memory round trips per workgroup)
    for (int i0 = 0; i0 < nlevav * ncol; i0 += 8 * 256) {
        u64 w[8];
        int lc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 256 * k + (int)threadIdx.x;
            const int l = i / ncol, c = i - l * ncol;
            lc[k] = i < nlevav * ncol ? l * pitch + c : -1;
            w[k] = lc[k] >= 0 ? bits[w00 + l * lev_stride + c] : 0ull;
            any |= (u32)(w[k] != 0ull && c >= cfirst);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (lc[k] >= 0) tile_mem[lc[k]] = w[k];
    }
    if (!__syncthreads_or((int)any)) return;
    const u32 step = DIR == 0 ? (u32)A2 : (u32)A1 * (u32)A2;
    for (int c = cfirst + wv; c < ncol; c += 4) {
        const int rr = DIR == 0 ? 0 : c / P, t = DIR == 0 ? c : c - rr * P;
        const bool have = lane < nlevav && lane < kTileLev, haveN = have && lane + 1 < nlevav;
        const u64 M = have ? tile(lane, c) : 0ull, N = haveN ? tile(lane + 1, c) : 0ull;
        const u64 pm = (have && t) ? tile(lane, c - 1) >> 63 : 0ull, q = (haveN && t) ? tile(lane + 1, c - 1) >> 63 : 0ull;
        const u64 cc = M & N;
        u64 reps = cc & ~((cc << 1) | (pm & q));
        const u64 segM = M & ~(M << 1), segN = N & ~(N << 1);
        const bool one = __popcll(segM) == 1 && __popcll(segN) == 1;
        if (DIR == 1 && reps && rr > 0) {
            // an overlap that holds a position z where the row above has members in BOTH planes is implied: (x,r,z) ~ (x,r-1,z) and
            // (x+1,r,z) ~ (x+1,r-1,z) are row-to-row links (DIR 0, all made), (x,r-1,z) ~ (x+1,r-1,z) is the link of the row above
            // (made, or implied in turn)
            const u64 up = tile(lane, c - P) & tile(lane + 1, c - P) & cc;      // (haveN holds: reps != 0)
            if (up) {
                u64 todo = reps;
                while (todo) {
                    const u64 low = todo & (~todo + 1ull);                      // this overlap's first voxel
                    todo ^= low;
                    const u64 run = (cc ^ (cc + low)) & cc;                     // ... and all of it inside the window
                    if (run & up) reps ^= low;
                }
            }
        }
        const u32 rowM = DIR == 0 ? (u32)blockIdx.y * (u32)A1 + (u32)(lev0 + lane) : (u32)(lev0 + lane) * (u32)A1 + (u32)(rlo + rr);
        const u32 base = rowM * (u32)A2 + 64u * (u32)t;
        const bool single = reps != 0ull && one;
        const u64 B = __ballot(single);
        const u64 z = ~B & (((u64)1 << lane) - 1ull);
        const int sl = z ? hi_bit(z) + 1 : 0;                                // first lane of this lane's chain
        const int nodeM = segM ? (int)(base + (u32)hi_bit(segM)) : 0;
        const int tgt = __shfl(nodeM, sl);
        bool first = single;
        if (single) reps = 0ull;
        while (first || reps) {
            int a, b;
            if (first) { a = tgt; b = (int)(base + step + (u32)hi_bit(segN)); first = false; }
            else {
                const int i = __ffsll((unsigned long long)reps) - 1;
                reps &= reps - 1;
                const u64 le = le_mask(i);
                a = (int)(base + (u32)hi_bit(segM & le)); b = (int)(base + step + (u32)hi_bit(segN & le));
            }
            uf_union(parent, a, b);
        }
    }
}
#undef tile

// generic form (rows longer than 2048 voxels): one lane per 64-voxel window, blockIdx.y = colour, neighbours linked pairwise
__global__ __launch_bounds__(256) void k_ccl_merge(const u64* __restrict__ bits_all, i64 nwords, pb3d_magic mP, pb3d_magic m1, int A0, int A1, int A2,
                                                   int* parent) {
    const u64* __restrict__ bits = bits_all + (i64)blockIdx.y * nwords;
    const i64 P = mP.d;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < nwords; idx += (i64)gridDim.x * blockDim.x) {
        const u64 M = bits[idx];
        if (!M) continue;
        const u32 row = pb3d_div((u32)idx, mP), t = (u32)idx - row * mP.d;
        const u32 x0 = pb3d_div(row, m1), x1 = row - x0 * m1.d;
        const u32 base = row * (u32)A2 + 64u * t;
        const bool has0 = x1 + 1 < (u32)A1, has1 = x0 + 1 < (u32)A0;
        const i64 n0 = idx + P, n1 = idx + (i64)A1 * P;                  // the same window one row / one plane on
        // all six words this lane may need are requested before any is used
        const u64 N0 = has0 ? bits[n0] : 0ull, N1 = has1 ? bits[n1] : 0ull;
        const u64 pm = t ? bits[idx - 1] >> 63 : 0ull;
        const u64 q0 = (t && has0) ? bits[n0 - 1] >> 63 : 0ull, q1 = (t && has1) ? bits[n1 - 1] >> 63 : 0ull;
        const u64 segM = M & ~(M << 1);
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const u64 N = dir == 0 ? N0 : N1;
            const u64 c = M & N;
            if (!c) continue;
            const u32 s = dir == 0 ? (u32)A2 : (u32)A1 * (u32)A2;
            const u64 segN = N & ~(N << 1);
            u64 reps = c & ~((c << 1) | (pm & (dir == 0 ? q0 : q1)));
            while (reps) {
                const int i = __ffsll((unsigned long long)reps) - 1;
                reps &= reps - 1;
                const u64 le = le_mask(i);
                uf_union(parent, (int)(base + (u32)hi_bit(segM & le)), (int)(base + s + (u32)hi_bit(segN & le)));
            }
        }
    }
}

// ---- roots: run starts that are still their own parent; 1024 windows per block, a lane per window ----------------------------------
// Roots are rare (one per component): only windows that hold one get their root mask written, a flag word per 64 windows says which
// (the dense mask array was 134 MB written and read again at 1024^3, almost all of it zeros).
constexpr int kWinPerBlock = 1024;

__global__ __launch_bounds__(256) void k_ccl_roots(const u64* __restrict__ bits_all, i64 nwords, pb3d_magic mP, int A2, const int* parent,
                                                   u64* __restrict__ rootbits_all, u64* __restrict__ rootflag_all, u32* __restrict__ chunk_count) {
    __shared__ u32 wsum[4];
    const u64* __restrict__ bits = bits_all + (i64)blockIdx.y * nwords;
    u64* __restrict__ rootbits = rootbits_all + (i64)blockIdx.y * nwords;
    u64* __restrict__ rootflag = rootflag_all + (i64)blockIdx.y * gridDim.x * (kWinPerBlock / 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i64 idx = (i64)blockIdx.x * kWinPerBlock + 256 * q + threadIdx.x;
        u64 roots = 0;
        if (idx < nwords) {
            const u64 w = bits[idx];
            if (w) {
                const u32 row = pb3d_div((u32)idx, mP), t = (u32)idx - row * mP.d;
                const u32 base = row * (u32)A2 + 64u * t;
                const u64 pm = t ? bits[idx - 1] >> 63 : 0ull;
                u64 cand = w & ~((w << 1) | pm);                     // (a segment that continues a run is never a root)
                while (cand) {
                    const int i = __ffsll((unsigned long long)cand) - 1;
                    cand &= cand - 1;
                    const int v = (int)(base + (u32)i);
                    if (ld(&parent[v]) == ~v) roots |= 1ull << i;
                }
                if (roots) rootbits[idx] = roots;
            }
        }
        const u64 fl = __ballot(roots != 0ull);
        if (lane == 0) rootflag[(i64)blockIdx.x * (kWinPerBlock / 64) + 4 * q + wv] = fl;
        cnt += (u32)__popcll(roots);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) wsum[wv] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[(i64)blockIdx.y * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of a colour's chunk counts, one block per colour (16 384 chunks at 1024^3); total[colour] = number of components
__global__ __launch_bounds__(1024) void k_ccl_scan(const u32* __restrict__ counts_all, i64 nchunks, u32* __restrict__ chunk_base_all, i64* __restrict__ total) {
    __shared__ u32 part[1024];
    const u32* __restrict__ counts = counts_all + (i64)blockIdx.x * nchunks;
    u32* __restrict__ chunk_base = chunk_base_all + (i64)blockIdx.x * nchunks;
    const i64 per = (nchunks + 1023) / 1024;
    const i64 b = (i64)threadIdx.x * per, e = b + per < nchunks ? b + per : nchunks;
    u32 s = 0;
    for (i64 i = b; i < e; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64) {                                      // one wavefront scans the 1024 partial sums, 16 per lane
        u32 loc[16], sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { loc[k] = part[16 * threadIdx.x + k]; sum += loc[k]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if ((int)threadIdx.x >= off) inc += t; }
        u32 run = inc - sum;
#pragma unroll
        for (int k = 0; k < 16; ++k) { part[16 * threadIdx.x + k] = run; run += loc[k]; }
        if (threadIdx.x == 63) total[blockIdx.x] = (i64)run;
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (i64 i = b; i < e; ++i) { chunk_base[i] = run; run += counts[i]; }
}

// parent[root] = its 1-based rank among the roots of its colour, in raster order (positive: from here on the entry is a finished label)
__global__ __launch_bounds__(256) void k_ccl_number(i64 nwords, pb3d_magic mP, int A2, const u64* __restrict__ rootbits_all, const u64* __restrict__ rootflag_all,
                                                    const u32* __restrict__ chunk_base, int* parent) {
    __shared__ u32 wsum[4];
    const u64* __restrict__ rootbits = rootbits_all + (i64)blockIdx.y * nwords;
    const u64* __restrict__ rootflag = rootflag_all + (i64)blockIdx.y * gridDim.x * (kWinPerBlock / 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // thread i takes the four consecutive windows 4 i .. 4 i + 3 of the chunk (raster order = thread order): one nibble of a flag word
    const u32 nib = (u32)(rootflag[(i64)blockIdx.x * (kWinPerBlock / 64) + (threadIdx.x >> 4)] >> (4 * (threadIdx.x & 15))) & 15u;
    u64 r[4];
    u32 mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i64 idx = (i64)blockIdx.x * kWinPerBlock + 4 * threadIdx.x + q;
        r[q] = ((nib >> q) & 1u) ? rootbits[idx] : 0ull;
        mine += (u32)__popcll(r[q]);
    }
    u32 inc = mine;
    for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 label = chunk_base[(i64)blockIdx.y * gridDim.x + blockIdx.x] + inc - mine;
    for (int k = 0; k < wv; ++k) label += wsum[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u64 m = r[q];
        if (!m) continue;
        const u32 idx = (u32)((i64)blockIdx.x * kWinPerBlock + 4 * threadIdx.x + q);
        const u32 row = pb3d_div(idx, mP), t = idx - row * mP.d;
        const u32 base = row * (u32)A2 + 64u * t;
        while (m) {
            const int i = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            st(&parent[base + (u32)i], (int)++label);
        }
    }
}

// ---- labels + statistics: a lane per window resolves the walks, then four voxels per lane are stored ------------------------------
// A walker may pass through entries other wavefronts are finishing at the same moment: it then reads either the old ~parent (still a
// valid path) or the label itself (the answer).
//
// Statistics record of component L (1-based) of colour k: 64 bytes {int lo[3], hi[3], pad[2]; u64 count, sum[3]} at
// ((copy * K + k) * dcap + L - 1); labels above dcap are not recorded (the caller then runs the separate pass).
constexpr int kSeg = 8;                 // segment labels of a window that are kept in LDS (windows with more take the slow path)
constexpr int kSlots = 32;              // statistics slots of a block
constexpr int kMaxCopies = 64;          // shadow copies of the record table (fewer when the table is large)

struct SegAcc {                         // one lane's run of segments of one component, in registers
    int key;                            // k * (dcap + 1) + L, 0 = empty
    int lo[3], hi[3];
    unsigned long long cs[4];
};

__device__ __forceinline__ void acc_flush(const SegAcc& a, int* slab, int (*slo)[3], int (*shi)[3], unsigned long long (*scs)[4], int K, int dcap,
                                          char* __restrict__ recs, int copy) {
    if (!a.key) return;
    int slot = a.key & (kSlots - 1), found = -1;
    for (int j = 0; j < kSlots; ++j) {
        const int old = atomicCAS(&slab[slot], 0, a.key);
        if (old == 0 || old == a.key) { found = slot; break; }
        slot = (slot + 1) & (kSlots - 1);
    }
    if (found >= 0) {
        for (int d = 0; d < 3; ++d) { atomicMin(&slo[found][d], a.lo[d]); atomicMax(&shi[found][d], a.hi[d]); }
        for (int d = 0; d < 4; ++d) atomicAdd(&scs[found][d], a.cs[d]);
    } else {                            // more than kSlots components under one block: straight to the block's shadow copy
        const int k = (a.key - 1) / (dcap + 1), L = a.key - k * (dcap + 1);
        char* rec = recs + ((i64)(copy * K + k) * dcap + (L - 1)) * 64;
        int* bb = (int*)rec;
        unsigned long long* cs = (unsigned long long*)(rec + 32);
        for (int d = 0; d < 3; ++d) { atomicMin(&bb[d], a.lo[d]); atomicMax(&bb[3 + d], a.hi[d]); }
        for (int d = 0; d < 4; ++d) atomicAdd(&cs[d], a.cs[d]);
    }
}

template <int KT, bool SPARSE, bool STATS>
__global__ __launch_bounds__(256) void k_ccl_finish(const u64* __restrict__ bits, i64 nwords, int K, pb3d_magic mP, pb3d_magic m1, int A2, int* parent,
                                                    int dcap, char* __restrict__ recs, int ncopies) {
    __shared__ u64 s_wany[4][64], s_st[4][64], s_info[4][64];
    __shared__ int s_tab[4][64][kSeg];
    __shared__ int slab[kSlots];
    __shared__ int slo[kSlots][3], shi[kSlots][3];
    __shared__ unsigned long long scs[kSlots][4];
    if (STATS) {
        if (threadIdx.x < kSlots) {
            slab[threadIdx.x] = 0;
            for (int a = 0; a < 3; ++a) { slo[threadIdx.x][a] = 0x7fffffff; shi[threadIdx.x][a] = -1; }
            for (int a = 0; a < 4; ++a) scs[threadIdx.x][a] = 0ull;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int copy = (int)(blockIdx.x & (unsigned)(ncopies - 1));
    SegAcc acc;
    acc.key = 0;
    for (i64 w0 = ((i64)blockIdx.x * 4 + wv) * 64; w0 < nwords; w0 += (i64)gridDim.x * 256) {
        // ---- phase A, one lane per WINDOW: all the walks of 64 windows are in flight together
        const i64 widx = w0 + lane;
        const bool valid = widx < nwords;
        const u32 row = valid ? pb3d_div((u32)widx, mP) : 0u, t = valid ? (u32)widx - row * mP.d : 0u;
        const u32 v0 = row * (u32)A2 + 64u * t;
        const int nval = valid ? (A2 - 64 * (int)t < 64 ? A2 - 64 * (int)t : 64) : 0;
        u64 wk[KT], wany = 0, stall = 0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            wk[k] = (k < K && valid) ? bits[(i64)k * nwords + widx] : 0ull;
            wany |= wk[k];
            stall |= wk[k] & ~(wk[k] << 1);
        }
        s_wany[wv][lane] = wany; s_st[wv][lane] = stall; s_info[wv][lane] = ((u64)(u32)nval << 32) | (u64)v0;
        if (wany) {
            u32 a0 = 0, a1 = 0;
            if (STATS) { a0 = pb3d_div(row, m1); a1 = row - a0 * m1.d; }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k >= K) break;
                u64 s = wk[k] & ~(wk[k] << 1);
                while (s) {
                    const int i = __ffsll((unsigned long long)s) - 1;
                    s &= s - 1;
                    int p = ld(&parent[v0 + (u32)i]);
                    if (p < 0) {
                        do p = ld(&parent[~p]); while (p < 0);
                        st(&parent[v0 + (u32)i], p);            // (its final value; the slow path below and other walkers read it)
                    }
                    const int slot = __popcll(stall & ((1ull << i) - 1ull));
                    if (slot < kSeg) s_tab[wv][lane][slot] = p;
                    if (STATS && p <= dcap) {
                        const u64 x = ~(wk[k] >> i);
                        const int len = x ? __ffsll((unsigned long long)x) - 1 : 64;
                        const int a2 = 64 * (int)t + i;
                        const int key = k * (dcap + 1) + p;
                        const unsigned long long cnt = (unsigned long long)len;
                        if (acc.key != key) {
                            acc_flush(acc, slab, slo, shi, scs, K, dcap, recs, copy);
                            acc.key = key;
                            acc.lo[0] = acc.hi[0] = (int)a0; acc.lo[1] = acc.hi[1] = (int)a1; acc.lo[2] = a2; acc.hi[2] = a2 + len - 1;
                            acc.cs[0] = cnt; acc.cs[1] = (unsigned long long)a0 * cnt; acc.cs[2] = (unsigned long long)a1 * cnt;
                            acc.cs[3] = (unsigned long long)(2 * a2 + len - 1) * cnt / 2ull;
                        } else {
                            acc.lo[0] = min(acc.lo[0], (int)a0); acc.hi[0] = max(acc.hi[0], (int)a0);
                            acc.lo[1] = min(acc.lo[1], (int)a1); acc.hi[1] = max(acc.hi[1], (int)a1);
                            acc.lo[2] = min(acc.lo[2], a2); acc.hi[2] = max(acc.hi[2], a2 + len - 1);
                            acc.cs[0] += cnt; acc.cs[1] += (unsigned long long)a0 * cnt; acc.cs[2] += (unsigned long long)a1 * cnt;
                            acc.cs[3] += (unsigned long long)(2 * a2 + len - 1) * cnt / 2ull;
                        }
                    }
                }
            }
        }
        const u64 busy = __ballot(wany != 0ull);
        u64 ovf = __ballot(__popcll(stall) > kSeg);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase B, four voxels per lane, four windows per step: registers and LDS only, 16-byte stores
        for (int g = 0; g < 16; ++g) {
            if (w0 + 4 * g >= nwords) break;
            if (SPARSE && !((busy >> (4 * g)) & 15ull)) continue;
            const int tt = 4 * g + (lane >> 4), p = 4 * (lane & 15);
            if ((ovf >> tt) & 1ull) continue;
            const u64 w = s_wany[wv][tt], stw = s_st[wv][tt], info = s_info[wv][tt];
            const u32 m4 = (u32)(w >> p) & 15u;
            const int nv = (int)(info >> 32) - p;
            if (nv <= 0 || (SPARSE && !m4)) continue;
            const int r = __popcll(stw & le_mask(p));          // segments that start at or before voxel p
            const u32 nb = (u32)(stw >> (p + 1)) & 7u;          // segment starts at voxels p + 1 .. p + 3 (bits past 63 are absent)
            const int* tab = s_tab[wv][tt];
            int L[4];
            const int L0 = r > 0 ? tab[r - 1] : 0;
            if (nb == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) L[j] = ((m4 >> j) & 1u) ? L0 : 0;
            } else {
                int rj = r;
                L[0] = (m4 & 1u) ? L0 : 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    rj += (int)((nb >> (j - 1)) & 1u);
                    L[j] = ((m4 >> j) & 1u) ? tab[rj - 1] : 0;
                }
            }
            int* dst = parent + (u32)info + (u32)p;
            if (nv >= 4) {
                i32x4a4 o; o.x = L[0]; o.y = L[1]; o.z = L[2]; o.w = L[3];
                if (SPARSE) *(i32x4a4*)dst = o;
                else __builtin_nontemporal_store(o, (i32x4a4*)dst);       // the full label volume: 4 B/voxel streamed out once
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < nv && (!SPARSE || ((m4 >> j) & 1u))) dst[j] = L[j];
            }
        }
        // ---- phase C, windows with more than kSeg segments (salt noise): one lane per voxel, labels read back from the segment starts
        if (ovf) {
            __threadfence();
            while (ovf) {
                const int tt = __ffsll((unsigned long long)ovf) - 1;
                ovf &= ovf - 1;
                const u64 w = s_wany[wv][tt], stw = s_st[wv][tt], info = s_info[wv][tt];
                const bool member = (w >> lane) & 1ull;
                int Lv = 0;
                if (member) Lv = ld(&parent[(u32)info + (u32)hi_bit(stw & le_mask(lane))]);
                if (SPARSE ? member : lane < (int)(info >> 32)) parent[(u32)info + (u32)lane] = Lv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (STATS) {
        acc_flush(acc, slab, slo, shi, scs, K, dcap, recs, copy);
        __syncthreads();
        if (threadIdx.x < kSlots && slab[threadIdx.x] > 0) {
            const int key = slab[threadIdx.x];
            const int k = (key - 1) / (dcap + 1), L = key - k * (dcap + 1);
            char* rec = recs + ((i64)(copy * K + k) * dcap + (L - 1)) * 64;
            int* bb = (int*)rec;
            unsigned long long* cs = (unsigned long long*)(rec + 32);
            for (int a = 0; a < 3; ++a) {
                if (slo[threadIdx.x][a] < ld(&bb[a])) atomicMin(&bb[a], slo[threadIdx.x][a]);
                if (shi[threadIdx.x][a] > ld(&bb[3 + a])) atomicMax(&bb[3 + a], shi[threadIdx.x][a]);
            }
            for (int a = 0; a < 4; ++a) atomicAdd(&cs[a], scs[threadIdx.x][a]);
        }
    }
}

// the shadow copies folded: final record (k, c) for c < min(total[k], dcap); the first kFirst of every colour also go into the head
// block that comes back to the host with the component counts in ONE copy
constexpr int kFirst = 64;

__global__ __launch_bounds__(256) void k_ccl_fold(int K, int dcap, const i64* __restrict__ total, const char* __restrict__ shadow, char* __restrict__ fin,
                                                  char* __restrict__ head, int ncopies) {
    const int k = (int)blockIdx.y, lane = threadIdx.x & 63;
    const i64 n = total[k] < dcap ? total[k] : dcap;
    // a wave per record, a lane per shadow copy (ncopies <= 64), reduced across the lanes
    for (i64 c = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += (i64)gridDim.x * 4) {
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
        unsigned long long cs[4] = {0, 0, 0, 0};
        if (lane < ncopies) {
            const char* rec = shadow + ((i64)(lane * K + k) * dcap + c) * 64;
            const int* bb = (const int*)rec;
            const unsigned long long* sp = (const unsigned long long*)(rec + 32);
            for (int a = 0; a < 3; ++a) { lo[a] = bb[a]; hi[a] = bb[3 + a]; }
            for (int a = 0; a < 4; ++a) cs[a] = sp[a];
        }
        for (int off = 32; off > 0; off >>= 1) {
            for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], __shfl_xor(lo[a], off)); hi[a] = max(hi[a], __shfl_xor(hi[a], off)); }
            for (int a = 0; a < 4; ++a) cs[a] += __shfl_xor(cs[a], off);
        }
        if (lane < 2 && (lane == 0 || c < kFirst)) {
            char* out = lane == 0 ? fin + ((i64)k * dcap + c) * 64 : head + ((i64)k * kFirst + c) * 64;
            int* ob = (int*)out;
            unsigned long long* os = (unsigned long long*)(out + 32);
            for (int a = 0; a < 3; ++a) { ob[a] = lo[a]; ob[3 + a] = hi[a]; }
            ob[6] = ob[7] = 0;
            for (int a = 0; a < 4; ++a) os[a] = cs[a];
        }
    }
}

}  // namespace

// Scratch slot 40: [header 64 B: i64 total[8]] [head: K x kFirst records] [final: K x dcap records] [shadow: kCopies x K x dcap records]
static int label_colors_impl(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const uint8_t* colors, int K, int C,
                             int32_t* d_labels, int64_t* ncomp, int64_t cap, bool members_only, int64_t* bbox_lo_hi, int64_t* count,
                             int64_t* coord_sum, int* stats_valid, pb3d_ccl_dev* dev = nullptr) {
    PB3D_REQUIRE(ctx && colors && ncomp && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_label_color: bad argument");
    PB3D_REQUIRE(K >= 1 && K <= kMaxColors, "pb3d_label_colors: between 1 and %d colours per call (got %d)", kMaxColors, K);
    const i64 n = A0 * A1 * A2;
    for (int k = 0; k < K; ++k) { ncomp[k] = 0; if (stats_valid) stats_valid[k] = 0; }
    if (n == 0) { if (stats_valid) for (int k = 0; k < K; ++k) stats_valid[k] = 1; return PB3D_OK; }
    PB3D_REQUIRE(n < (1ll << 31), "pb3d_label_color: grid too large for 32-bit labels");
    PB3D_REQUIRE(d_grid && d_labels, "pb3d_label_color: null buffer");
    const bool stats = (stats_valid != nullptr || dev != nullptr) && cap > 0;
    if (stats && !dev) PB3D_REQUIRE(bbox_lo_hi && count && coord_sum, "pb3d_label_color_stats: null output");
    CclColors cols;
    for (int k = 0; k < kMaxColors; ++k) cols.c[k] = 0xffffffffu;
    for (int k = 0; k < K; ++k) {
        cols.c[k] = C == 1 ? (u32)colors[k] : ((u32)colors[3 * k] | ((u32)colors[3 * k + 1] << 8) | ((u32)colors[3 * k + 2] << 16));
        for (int j = 0; j < k; ++j) PB3D_REQUIRE(cols.c[j] != cols.c[k], "pb3d_label_colors: colour %d repeats colour %d", k, j);
    }
    const i64 rows = A0 * A1, P = (A2 + 63) / 64, nwords = rows * P;
    const i64 nchunks = (nwords + kWinPerBlock - 1) / kWinPerBlock;
    PB3D_REQUIRE(nchunks <= 0x7fffffffll / kMaxColors, "pb3d_label_color: grid too large");
    void *bits, *rootbits, *chunks;
    ctx->ccl_last.valid = false;
    PB3D_TRY(pb3d_scratch(ctx, 42, (size_t)K * (size_t)nwords * 8, &bits));  // a slot of its own: the bits outlive the call (ctx->ccl_last)
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)K * (size_t)nwords * 8, &rootbits));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)K * (size_t)nchunks * (8 + 8 * (kWinPerBlock / 64)) + 16, &chunks));   // [root flags | chunk counts | chunk bases]
    // records kept on the device per colour; the shadow copies (same-address atomics serialise: every block flushes into copy
    // blockIdx % ncopies) are bounded to 16 MiB
    int dcap = 0, ncopies = 1;
    if (stats) {
        dcap = (int)(cap < 16384 / K ? cap : 16384 / K); if (dcap < 1) dcap = 1;
        ncopies = kMaxCopies;
        while (ncopies > 1 && (size_t)ncopies * (size_t)K * (size_t)dcap * 64 > ((size_t)16 << 20)) ncopies >>= 1;
    }
    const size_t head_bytes = 64 + (size_t)K * kFirst * 64, fin_bytes = (size_t)K * (size_t)dcap * 64;
    void* sblk = nullptr;
    PB3D_TRY(pb3d_scratch(ctx, 40, head_bytes + fin_bytes * (size_t)(1 + ncopies), &sblk));
    char* head = (char*)sblk + 64;
    char* fin = (char*)sblk + head_bytes;
    char* shadow = fin + fin_bytes;
    i64* total = (i64*)sblk;
    u64* rootflag = (u64*)chunks;
    u32* chunk_count = (u32*)(rootflag + (size_t)K * (size_t)nchunks * (kWinPerBlock / 64));
    u32* chunk_base = chunk_count + (size_t)K * (size_t)nchunks;
    const pb3d_magic mP = pb3d_make_magic((u32)P), m1 = pb3d_make_magic((u32)A1);
    int* parent = (int*)d_labels;
    const int KT = K == 1 ? 1 : (K == 2 ? 2 : (K <= 4 ? 4 : 8));

    {
        // rows per wave: the largest power of two R <= 16 with A2 <= 1024 / R
        int lgR = 0;
        while (lgR < 4 && A2 <= (kChunkVox >> (lgR + 1))) ++lgR;
        int zero_is_colour = 0;
        for (int k = 0; k < K; ++k) zero_is_colour |= cols.c[k] == 0u;
        const i64 nrec = stats ? (i64)ncopies * K * dcap : 0;
        const dim3 ig(pb3d_stream_blocks(ctx, (rows + (1 << lgR) - 1) >> lgR, 4, ctx->tune_ccl_init_blocks));
#define PB3D_CCL_INIT(CC, KK) hipLaunchKernelGGL((k_ccl_init<CC, KK>), ig, dim3(256), 0, ctx->stream, d_grid, rows, (int)A2, (int)P, K, cols, nwords, (u64*)bits, parent, lgR, zero_is_colour, nrec, shadow)
        if (C == 1) { if (KT == 1) PB3D_CCL_INIT(1, 1); else if (KT == 2) PB3D_CCL_INIT(1, 2); else if (KT == 4) PB3D_CCL_INIT(1, 4); else PB3D_CCL_INIT(1, 8); }
        else { if (KT == 1) PB3D_CCL_INIT(3, 1); else if (KT == 2) PB3D_CCL_INIT(3, 2); else if (KT == 4) PB3D_CCL_INIT(3, 4); else PB3D_CCL_INIT(3, 8); }
#undef PB3D_CCL_INIT
    }
    PB3D_CHECK_LAUNCH();
    if (2 * P <= kTileCol && A0 <= 65535 && A1 <= 65535 && ctx->tune_ccl_merge != 1) {
        // rows a plane-to-plane tile links (+ the row above them)
        const int maxcol = ctx->tune_ccl_tilecols > 0 ? ctx->tune_ccl_tilecols : 32;
        int RT = (int)((maxcol < kTileCol ? maxcol : kTileCol) / P - 1);
        if (RT < 1) RT = 1;
        hipLaunchKernelGGL(k_ccl_merge_tile<0>, dim3((unsigned)((A1 + 63) / 64), (unsigned)A0, (unsigned)K), dim3(256), (size_t)(kTileLev + 1) * (size_t)(P + 1) * 8,
                           ctx->stream, (const u64*)bits, nwords, (int)A0, (int)A1, (int)A2, (int)P, 1, parent);
        hipLaunchKernelGGL(k_ccl_merge_tile<1>, dim3((unsigned)((A0 + 63) / 64), (unsigned)((A1 + RT - 1) / RT), (unsigned)K), dim3(256),
                           (size_t)(kTileLev + 1) * (size_t)((RT + 1) * P + 1) * 8, ctx->stream, (const u64*)bits, nwords, (int)A0, (int)A1, (int)A2, (int)P, RT, parent);
    } else {
        hipLaunchKernelGGL(k_ccl_merge, dim3(pb3d_stream_blocks(ctx, nwords, 256, 16), (unsigned)K), dim3(256), 0, ctx->stream, (const u64*)bits, nwords, mP, m1,
                           (int)A0, (int)A1, (int)A2, parent);
    }
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_roots, dim3((unsigned)nchunks, (unsigned)K), dim3(256), 0, ctx->stream, (const u64*)bits, nwords, mP, (int)A2, (const int*)parent,
                       (u64*)rootbits, rootflag, chunk_count);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_scan, dim3((unsigned)K), dim3(1024), 0, ctx->stream, (const u32*)chunk_count, nchunks, chunk_base, total);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_number, dim3((unsigned)nchunks, (unsigned)K), dim3(256), 0, ctx->stream, nwords, mP, (int)A2, (const u64*)rootbits,
                       (const u64*)rootflag, (const u32*)chunk_base, parent);
    PB3D_CHECK_LAUNCH();
    {
        // a persistent grid: every block ends with one flush of its statistics slots -- as many blocks as are resident together (a grid
        // of eight per CU ran a second, partial round with the 5 per CU that 87 registers allow: the full-label pass took 0.99 ms at 1024^3)
#define PB3D_CCL_FIN(KK, SP, STT) do { \
            int per_cu = ctx->tune_ccl_blocks; \
            if (per_cu <= 0 && !(SP)) per_cu = 64;      /* the full label volume is a 4 B/voxel write stream: many short workgroups (1.78 -> 1.70 ms at 1024^3; the members-only form prefers the resident grid: 1.11 against 1.15 - 1.24) */ \
            if (per_cu <= 0) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_ccl_finish<KK, SP, STT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4; } \
            const dim3 fg(pb3d_stream_blocks(ctx, (nwords + 63) / 64, 4, per_cu)); \
            hipLaunchKernelGGL((k_ccl_finish<KK, SP, STT>), fg, dim3(256), 0, ctx->stream, (const u64*)bits, nwords, K, mP, m1, (int)A2, parent, dcap, shadow, ncopies); } while (0)
#define PB3D_CCL_FIN_K(SP, STT) do { if (KT == 1) PB3D_CCL_FIN(1, SP, STT); else if (KT == 2) PB3D_CCL_FIN(2, SP, STT); else if (KT == 4) PB3D_CCL_FIN(4, SP, STT); else PB3D_CCL_FIN(8, SP, STT); } while (0)
        if (members_only) { if (stats) PB3D_CCL_FIN_K(true, true); else PB3D_CCL_FIN_K(true, false); }
        else { if (stats) PB3D_CCL_FIN_K(false, true); else PB3D_CCL_FIN_K(false, false); }
#undef PB3D_CCL_FIN_K
#undef PB3D_CCL_FIN
    }
    PB3D_CHECK_LAUNCH();
    if (stats) {
        hipLaunchKernelGGL(k_ccl_fold, dim3((unsigned)((dcap + 3) / 4 < 64 ? (dcap + 3) / 4 : 64), (unsigned)K), dim3(256), 0, ctx->stream, K, dcap, (const i64*)total, (const char*)shadow, fin, head, ncopies);
        PB3D_CHECK_LAUNCH();
    }
    // the membership bits of THIS label volume stay where they are: consumers that only need the members' labels (the component loop,
    // the recolouring) walk the 1-bit-per-voxel arrays instead of the 4-byte-per-voxel one
    pb3d_ctx::CclLast& cl = ctx->ccl_last;
    cl.valid = true; cl.labels = d_labels; cl.bits = bits; cl.rows = rows; cl.A2 = A2; cl.P = P; cl.gen = ctx->scratch_slot_gen[42];
    cl.members_only = members_only; cl.K = K; cl.C = C;
    for (int k = 0; k < kMaxColors; ++k) cl.colors[k] = cols.c[k];
    if (dev) {          // a consumer on the device (pb3d_recolor_backward_dev): counts and records stay there, no host wait
        dev->total = total; dev->records = fin; dev->dcap = dcap;
        return PB3D_OK;
    }
    // the component counts and -- optimistically -- the statistics of the first kFirst components of every colour come back in ONE copy
    struct Rec { int bb[8]; unsigned long long cs[4]; };
    static_assert(sizeof(Rec) == 64, "the statistics block is 64-byte records");
    char* hb = (char*)ctx->pinned + 1024;
    static_assert(64 + (size_t)kMaxColors * kFirst * 64 + 1024 + 64 <= (1 << 16), "the pinned area holds the read-back block");
    PB3D_HIP(hipMemcpyAsync(hb, sblk, stats ? head_bytes : 64, hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    const i64* nroots = (const i64*)hb;
    for (int k = 0; k < K; ++k) ncomp[k] = nroots[k];
    if (stats) {
        for (int k = 0; k < K; ++k) {
            if (nroots[k] > dcap) continue;
            int64_t* bb = bbox_lo_hi + (size_t)k * (size_t)cap * 6;
            int64_t* ct = count + (size_t)k * (size_t)cap;
            int64_t* sm = coord_sum + (size_t)k * (size_t)cap * 3;
            auto put = [&](i64 c, const Rec& r) {
                for (int a = 0; a < 3; ++a) { bb[6 * c + a] = r.bb[a]; bb[6 * c + 3 + a] = (i64)r.bb[3 + a] + 1; }
                ct[c] = (i64)r.cs[0];
                for (int a = 0; a < 3; ++a) sm[3 * c + a] = (i64)r.cs[1 + a];
            };
            const i64 n0 = nroots[k] < kFirst ? nroots[k] : kFirst;
            const Rec* hr = (const Rec*)(hb + 64) + (size_t)k * kFirst;
            for (i64 c = 0; c < n0; ++c) put(c, hr[c]);
            if (nroots[k] > n0) {
                std::vector<Rec> more((size_t)(nroots[k] - n0));
                PB3D_HIP(hipMemcpyAsync(more.data(), fin + ((size_t)k * (size_t)dcap + (size_t)n0) * 64, more.size() * sizeof(Rec), hipMemcpyDeviceToHost, ctx->stream));
                PB3D_TRY(pb3d_stream_sync(ctx));
                for (i64 c = n0; c < nroots[k]; ++c) put(c, more[(size_t)(c - n0)]);
            }
            stats_valid[k] = 1;
        }
    }
    return PB3D_OK;
}

extern "C" int pb3d_label_color_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                                    int32_t* d_labels, int64_t* ncomp) {
    return label_colors_impl(ctx, d_grid_rgb, A0, A1, A2, color, 1, 3, d_labels, ncomp, 0, false, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int pb3d_label_color_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                                          int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                          int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_color_stats: null output");
    return label_colors_impl(ctx, d_grid_rgb, A0, A1, A2, color, 1, 3, d_labels, ncomp, cap, members_only != 0, bbox_lo_hi, count, coord_sum, stats_valid);
}

// the components of SEVERAL colours in one labelling sequence: the colour grid is read once
extern "C" int pb3d_label_colors_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t* colors,
                                           int ncolors, int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi,
                                           int64_t* count, int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_colors_stats: null output");
    return label_colors_impl(ctx, d_grid_rgb, A0, A1, A2, colors, ncolors, 3, d_labels, ncomp, cap, members_only != 0, bbox_lo_hi, count, coord_sum, stats_valid);
}

// the same on a 1-byte LABEL volume (row N3): components of the voxels whose label is `value`
extern "C" int pb3d_label_value_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, uint8_t value,
                                          int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                          int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_value_stats: null output");
    return label_colors_impl(ctx, d_grid_lab, A0, A1, A2, &value, 1, 1, d_labels, ncomp, cap, members_only != 0, bbox_lo_hi, count, coord_sum, stats_valid);
}

extern "C" int pb3d_label_values_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, const uint8_t* values,
                                           int nvalues, int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi,
                                           int64_t* count, int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_values_stats: null output");
    return label_colors_impl(ctx, d_grid_lab, A0, A1, A2, values, nvalues, 1, d_labels, ncomp, cap, members_only != 0, bbox_lo_hi, count, coord_sum, stats_valid);
}

// labelling + statistics of ONE colour with everything left on the device (csrc/components.hip: pb3d_recolor_backward_dev)
int pb3d_ccl_label_on_device(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3], int channels, int32_t* d_labels,
                             int64_t cap, pb3d_ccl_dev* dev) {
    int64_t ncomp = 0;
    return label_colors_impl(ctx, d_grid, A0, A1, A2, color, 1, channels, d_labels, &ncomp, cap, true, nullptr, nullptr, nullptr, nullptr, dev);
}
