// N1: 3-D connected components (6-connectivity) of one colour -- scipy.ndimage.label(mask) with the default structure, as called at
// reference utils/voxel_carving_utils.py:175 and :254: components are numbered in raster order of their first voxel.
//
// Round-2 form.  The first version kept a byte mask, a parent array, a root array and the caller's label array (14 B/voxel of state),
// walked one thread per voxel in every pass and ranked the roots through the float point extractor: 10.9 ms at 1024^3.  Here
//   * membership is ONE BIT per voxel, formed by the pass that also initialises the forest (k_ccl_init: the only pass that reads the
//     colour grid -- a wavefront per row, 16 voxels per lane);
//   * the caller's int32 label array IS the union-find forest: while the forest is being built a member holds ~parent (negative),
//     a non-member holds 0 -- already its final value -- and a finished voxel holds its label (positive), so the three states never
//     collide and a walker that meets a positive value has met its answer;
//   * all passes work in the ROW FRAME: a row (fastest axis, A2 voxels) is ceil(A2 / 64) 64-bit windows of a row-padded bit array
//     (word row * P + t; the bits past A2 in a row's last word are zero), so A2 needs no alignment, the neighbours of a window along
//     the two slow axes are the same window one row / one plane on (+P, +A1 * P words), and the links are pure bit arithmetic: a link
//     (v, v+s) is made only at the FIRST voxel of every overlap of two runs, i.e. one union per pair of touching runs instead of one
//     per touching face;
//   * union-find nodes are whole runs (k_ccl_init hands every voxel of a run the run's first voxel as parent, carrying the start
//     across windows), the smaller index always becomes the root, so root == first voxel in raster order and label == rank of
//     the root among all roots: a popcount scan over per-window root bit masks (k_ccl_roots / k_ccl_scan / k_ccl_number);
//   * the last pass resolves ONE walk per run and writes the labels of the run's voxels as coalesced rows.
// HBM traffic at 1024^3: 3 B/voxel read once, 4 B/voxel written once plus 4 B per member voxel twice; everything else is bit masks.
#include <vector>

#include "pb3d_internal.h"

namespace {

typedef u32 u32x4c __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- forest on complemented indices: parent[v] == ~v at a root -------------------------------------------------------------
// find with path halving (every store is an ancestor, so the forest stays valid under concurrent unions)
__device__ __forceinline__ int uf_find(int* parent, int v) {
    while (true) {
        const int p = ~ld(&parent[v]);
        if (p == v) return v;
        const int gp = ~ld(&parent[p]);
        if (gp != p) st(&parent[v], ~gp);
        v = p;
    }
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }        // the larger root goes under the smaller: ~b > ~a
        const int old = atomicMax(&parent[a], ~b);
        if (old == ~a) return;
        a = ~old;                                            // someone re-parented a meanwhile: retry from there
    }
}

__device__ __forceinline__ u64 low_mask(int nbits) { return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull); }
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)v, l);
}
typedef u32 u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));

// ---- membership bits + forest initialisation: one wavefront per row ---------------------------------------------------------
// A row is taken in chunks of 1024 voxels: lane l reads the 48 bytes of voxels 16 l .. 16 l + 15 of the chunk (16-byte loads at
// whatever alignment the row has), compares them with the colour, and lanes 0..15 collect the chunk's sixteen 64-bit windows with
// cross-lane reads (window l = the masks of lanes 4 l .. 4 l + 3).  The windows go to the padded bit array for the later passes and
// drive the forest initialisation at once: one lane per voxel of window tt -- member: ~(first voxel of the run), non-member: 0; the run
// start is carried across windows and chunks.
constexpr int kChunkVox = 1024;

// C = 1: the grid is a 1-byte LABEL volume (row N3) and color24 the label -- a lane's 16 voxels are one 16-byte load.
// SPARSE: only the members' entries of the forest / label array are written (the 4 B/voxel of zeros for everybody else are most of this
// pass's traffic: 292 MB of 511 at Taj 512).  Every later pass of the labelling touches members only; consumers must consult the bits.
template <int C, bool SPARSE>
__global__ __launch_bounds__(256) void k_ccl_init(const u8* __restrict__ grid, i64 rows, int A2, int P, u32 color24, u64* __restrict__ bits,
                                                  int* __restrict__ parent) {
    const int lane = threadIdx.x & 63;
    const u64 le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const i64 nbytes = C * rows * (i64)A2;
    for (i64 rw = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); rw < rows; rw += (i64)gridDim.x * 4) {
        const u32 row = (u32)__builtin_amdgcn_readfirstlane((int)rw);
        const u32 base = row * (u32)A2;
        u32 carry_start = 0, prevbit = 0;
        for (int c0 = 0; c0 < A2; c0 += kChunkVox) {
            // ---- 16 membership bits of this lane's voxels c0 + 16 lane .. + 15
            const int v = c0 + 16 * lane;
            u32 m16 = 0;
            if (C == 1 && v < A2) {
                const i64 boff = (i64)base + v;
                u32 w[4];
                if (boff + 16 <= nbytes) { const u32x4a1 t = *(const u32x4a1*)(grid + boff); w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; }
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u32 t = 0;
                        for (int b = 0; b < 4; ++b) { const i64 o = boff + 4 * k + b; if (o < nbytes) t |= (u32)grid[o] << (8 * b); }
                        w[k] = t;
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) m16 |= (u32)(((w[i >> 2] >> (8 * (i & 3))) & 0xffu) == color24) << i;
                if (v + 16 > A2) m16 &= (1u << (A2 - v)) - 1u;
            } else if (v < A2) {
                const i64 boff = 3 * ((i64)base + v);
                u32 w[13];
                w[12] = 0u;
                if (boff + 48 <= nbytes) {                               // (past the row's end these are the next row's voxels: masked off below)
                    const u32x4a1* p = (const u32x4a1*)(grid + boff);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const u32x4a1 t = p[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
                } else {                                                 // the grid's last 47 bytes: byte by byte
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        u32 t = 0;
                        for (int b = 0; b < 4; ++b) { const i64 o = boff + 4 * k + b; if (o < nbytes) t |= (u32)grid[o] << (8 * b); }
                        w[k] = t;
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int j = (3 * i) >> 2, sh = (3 * i) & 3;
                    m16 |= (u32)((__builtin_amdgcn_alignbyte(w[j + 1], w[j], (u32)sh) & 0x00ffffffu) == color24) << i;
                }
                if (v + 16 > A2) m16 &= (1u << (A2 - v)) - 1u;
            }
            // ---- the chunk's windows, one per lane 0..15
            const u64 wl = (u64)(u32)__shfl((int)m16, 4 * lane) | ((u64)(u32)__shfl((int)m16, 4 * lane + 1) << 16) |
                           ((u64)(u32)__shfl((int)m16, 4 * lane + 2) << 32) | ((u64)(u32)__shfl((int)m16, 4 * lane + 3) << 48);
            const int t0 = c0 >> 6;
            const int cw = (A2 - c0 + 63) >> 6 < 16 ? (A2 - c0 + 63) >> 6 : 16;
            if (lane < cw) bits[(i64)row * P + t0 + lane] = wl;
            // ---- forest initialisation, one lane per voxel of window tt
            for (int tt = 0; tt < cw; ++tt) {
                const int t = t0 + tt;
                const u64 w64 = readlane64(wl, tt);
                const u64 starts = w64 & ~((w64 << 1) | (u64)prevbit);
                if (lane < A2 - 64 * t) {
                    int val = 0;
                    if ((w64 >> lane) & 1ull) {
                        const u64 upto = starts & le;
                        const u32 s0 = upto ? (u32)(64 * t + 63 - __clzll((long long)upto)) : carry_start;
                        val = ~(int)(base + s0);
                    }
                    if (!SPARSE || val != 0) parent[base + 64u * (u32)t + (u32)lane] = val;
                }
                if ((w64 >> 63) && starts) carry_start = (u32)(64 * t + 63 - __clzll((long long)starts));
                prevbit = (u32)(w64 >> 63);
            }
        }
    }
}

// ---- links along the two slow axes: one lane per 64-voxel window of a row ----------------------------------------------------
__global__ __launch_bounds__(256) void k_ccl_merge(const u64* __restrict__ bits, i64 nwords, pb3d_magic mP, pb3d_magic m1, int A0, int A1, int A2,
                                                   int* parent) {
    const i64 P = mP.d;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < nwords; idx += (i64)gridDim.x * blockDim.x) {
        const u64 M = bits[idx];
        if (!M) continue;
        const u32 row = pb3d_div((u32)idx, mP), t = (u32)idx - row * mP.d;
        const u32 a0 = pb3d_div(row, m1), a1 = row - a0 * m1.d;
        const u32 base = row * (u32)A2 + 64u * t;
        const u64 pm = t ? bits[idx - 1] >> 63 : 0ull;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            if (dir == 0 ? a1 + 1 >= (u32)A1 : a0 + 1 >= (u32)A0) continue;
            const u32 s = dir == 0 ? (u32)A2 : (u32)A1 * (u32)A2;
            const i64 nidx = idx + (dir == 0 ? P : (i64)A1 * P);            // the same window one row / one plane on
            const u64 c = M & bits[nidx];
            if (!c) continue;
            const u64 pc = (pm && t) ? bits[nidx - 1] >> 63 : 0ull;
            u64 reps = c & ~((c << 1) | pc);                    // first voxel of every overlap of two runs
            while (reps) {
                const int i = __ffsll((unsigned long long)reps) - 1;
                reps &= reps - 1;
                uf_union(parent, (int)(base + (u32)i), (int)(base + (u32)i + s));
            }
        }
    }
}

// ---- roots: run starts that are still their own parent; 1024 windows per block, 4 consecutive windows per thread -------------
constexpr int kWinPerBlock = 1024;

__device__ __forceinline__ u64 run_starts(const u64* __restrict__ bits, u32 idx, const pb3d_magic mP, int A2, u32* base_out) {
    const u32 row = pb3d_div(idx, mP), t = idx - row * mP.d;
    *base_out = row * (u32)A2 + 64u * t;
    const u64 w = bits[idx];
    if (!w) return 0ull;
    const u64 pm = t ? bits[idx - 1] >> 63 : 0ull;
    return w & ~((w << 1) | pm);
}

__global__ __launch_bounds__(256) void k_ccl_roots(const u64* __restrict__ bits, i64 nwords, pb3d_magic mP, int A2, const int* parent,
                                                   u64* __restrict__ rootbits, u32* __restrict__ chunk_count) {
    __shared__ u32 wsum[4];
    u32 cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i64 idx = (i64)blockIdx.x * kWinPerBlock + 4 * threadIdx.x + q;
        if (idx >= nwords) break;
        u32 base;
        u64 cand = run_starts(bits, (u32)idx, mP, A2, &base), roots = 0;
        while (cand) {
            const int i = __ffsll((unsigned long long)cand) - 1;
            cand &= cand - 1;
            const int v = (int)(base + (u32)i);
            if (ld(&parent[v]) == ~v) roots |= 1ull << i;
        }
        rootbits[idx] = roots;
        cnt += (u32)__popcll(roots);
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the chunk counts, one block (16 384 chunks at 1024^3); *total = number of components
__global__ __launch_bounds__(1024) void k_ccl_scan(const u32* __restrict__ counts, i64 nchunks, u32* __restrict__ chunk_base, i64* __restrict__ total) {
    __shared__ u32 part[1024];
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
        if (threadIdx.x == 63) *total = (i64)run;
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (i64 i = b; i < e; ++i) { chunk_base[i] = run; run += counts[i]; }
}

// parent[root] = its 1-based rank in raster order (positive: from here on the entry is a finished label)
__global__ __launch_bounds__(256) void k_ccl_number(i64 nwords, pb3d_magic mP, int A2, const u64* __restrict__ rootbits,
                                                    const u32* __restrict__ chunk_base, int* parent) {
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 r[4];
    u32 mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i64 idx = (i64)blockIdx.x * kWinPerBlock + 4 * threadIdx.x + q;
        r[q] = idx < nwords ? rootbits[idx] : 0ull;
        mine += (u32)__popcll(r[q]);
    }
    u32 inc = mine;
    for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 label = chunk_base[blockIdx.x] + inc - mine;
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

// ---- labels: one wavefront per row; the first lane of every run walks to a finished entry, the run's lanes store its label -----
// A walker may pass through entries other wavefronts are finishing at the same moment: it then reads either the old ~parent (still a
// valid path) or the label itself (the answer).
constexpr int kChunkWin = 32;                                    // windows of a row handled together by the last pass (2048 voxels)

// STATS: the per-component statistics of pb3d_component_stats_dev (bounding box, voxel count, coordinate sums) are gathered HERE, where
// the labels are in registers: every segment of member voxels inside a window is one closed-form contribution of its first lane
// into a small per-block table in LDS (flushed with one set of global atomics per label per block).  Only windows that hold members
// cost anything -- the separate statistics pass re-read the whole 4 B/voxel label volume (152 us at Taj 512, as much as the labelling).
// Labels above `cap` are not recorded (the caller then runs the separate pass).
constexpr int kFinSlots = 16;

template <bool STATS>
__global__ __launch_bounds__(256) void k_ccl_finish(const u64* __restrict__ bits, i64 rows, int A2, int P, int* parent, pb3d_magic m1, int cap,
                                                    int* __restrict__ bbox, unsigned long long* __restrict__ cnt_sum, int abl) {
    __shared__ int table[4][kChunkWin][32];                      // labels of the runs that start in window t, in order
    __shared__ int slab[kFinSlots];
    __shared__ int slo[kFinSlots][3], shi[kFinSlots][3];
    __shared__ unsigned long long scs[kFinSlots][4];
    if (STATS) {
        if (threadIdx.x < kFinSlots) {
            slab[threadIdx.x] = 0;
            for (int a = 0; a < 3; ++a) { slo[threadIdx.x][a] = 0x7fffffff; shi[threadIdx.x][a] = -1; }
            for (int a = 0; a < 4; ++a) scs[threadIdx.x][a] = 0ull;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    for (i64 rw = (i64)blockIdx.x * 4 + wv; rw < rows; rw += (i64)gridDim.x * 4) {
        const u32 row = (u32)__builtin_amdgcn_readfirstlane((int)rw);
        const u32 base = row * (u32)A2;
        int carry_label = 0;
        u32 prevbit = 0;
        for (int t0 = 0; t0 < P; t0 += kChunkWin) {
            const int cw = P - t0 < kChunkWin ? P - t0 : kChunkWin;
            const u64 wl = lane < cw ? bits[(i64)row * P + t0 + lane] : 0ull;
            // phase A, one lane per WINDOW: all the walks of the chunk are in flight together
            const u64 before = __shfl_up((unsigned long long)wl, 1);
            const u64 pb = lane == 0 ? (u64)prevbit : before >> 63;
            const u64 starts_l = wl & ~((wl << 1) | pb);
            {
                u64 s = starts_l;
                int k = 0;
                while (s) {
                    const int i = __ffsll((unsigned long long)s) - 1;
                    s &= s - 1;
                    int p = ld(&parent[base + 64u * (u32)(t0 + lane) + (u32)i]);
                    while (p < 0) p = ld(&parent[~p]);
                    table[wv][lane][k++] = p;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // phase B, one lane per VOXEL of window tt: registers and LDS only
            for (int tt = 0; tt < cw; ++tt) {
                const u64 w = readlane64(wl, tt);
                if (!w) continue;
                const u64 upto = readlane64(starts_l, tt) & le;
                const int ridx = __popcll(upto);
                const int Lt = table[wv][tt][ridx > 0 ? ridx - 1 : 0];
                const int L = ridx > 0 ? Lt : carry_label;
                if ((w >> lane) & 1ull) st(&parent[base + 64u * (u32)(t0 + tt) + (u32)lane], L);
                if (STATS) {
                    const bool seg = ((w >> lane) & 1ull) && (lane == 0 || !((w >> (lane - 1)) & 1ull));
                    if (seg && L > 0 && L <= cap && !(abl & 2)) {
                        const u64 stop = ~w & ~le;                                   // the first non-member above this lane
                        const int len = (stop ? __ffsll((unsigned long long)stop) - 1 : 64) - lane;
                        const u32 a0 = pb3d_div(row, m1), a1 = row - a0 * m1.d;
                        const int a2 = 64 * (t0 + tt) + lane;
                        const int lo[3] = {(int)a0, (int)a1, a2}, hi[3] = {(int)a0, (int)a1, a2 + len - 1};
                        const unsigned long long cnt = (unsigned long long)len;
                        const unsigned long long sm[3] = {(unsigned long long)a0 * cnt, (unsigned long long)a1 * cnt,
                                                          (unsigned long long)(2 * a2 + len - 1) * cnt / 2ull};
                        int slot = L & (kFinSlots - 1), found = -1;
                        for (int k = 0; k < kFinSlots; ++k) {
                            const int old = atomicCAS(&slab[slot], 0, L);
                            if (old == 0 || old == L) { found = slot; break; }
                            slot = (slot + 1) & (kFinSlots - 1);
                        }
                        if (found >= 0) {
                            for (int a = 0; a < 3; ++a) { atomicMin(&slo[found][a], lo[a]); atomicMax(&shi[found][a], hi[a]); }
                            atomicAdd(&scs[found][0], cnt);
                            for (int a = 0; a < 3; ++a) atomicAdd(&scs[found][1 + a], sm[a]);
                        } else {
                            int* bb = bbox + 16 * (L - 1);                          // one 64-byte record per component: 6 ints of box, 4 u64 of count / sums
                            for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], lo[a]); atomicMax(&bb[3 + a], hi[a]); }
                            unsigned long long* cs = cnt_sum + 8 * (L - 1);
                            atomicAdd(&cs[0], cnt);
                            for (int a = 0; a < 3; ++a) atomicAdd(&cs[1 + a], sm[a]);
                        }
                    }
                }
                carry_label = __builtin_amdgcn_readlane(L, 63);      // meaningful only when the window ends inside a run
            }
            prevbit = (u32)(readlane64(wl, cw - 1) >> 63);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (STATS) {
        __syncthreads();
        if (threadIdx.x < kFinSlots && slab[threadIdx.x] > 0 && !(abl & 1)) {
            const int Lc = slab[threadIdx.x];
            int* bb = bbox + 16 * (Lc - 1);
            // thousands of blocks flush into the same few records and same-address atomics serialise: a box that is already wide
            // enough (the usual case after the first blocks) costs a load, not an atomic
            for (int a = 0; a < 3; ++a) {
                if (slo[threadIdx.x][a] < ld(&bb[a])) atomicMin(&bb[a], slo[threadIdx.x][a]);
                if (shi[threadIdx.x][a] > ld(&bb[3 + a])) atomicMax(&bb[3 + a], shi[threadIdx.x][a]);
            }
            unsigned long long* cs = cnt_sum + 8 * (Lc - 1);
            for (int a = 0; a < 4; ++a) atomicAdd(&cs[a], scs[threadIdx.x][a]);
        }
    }
}

// the statistics block: a 64-byte header (the component count lands here) + one 64-byte record per component
// {int lo[3], hi[3], pad[2]; u64 count, sum[3]} -- the count and the first records come back in ONE copy
__global__ __launch_bounds__(256) void k_fin_stats_init(int cap, int* __restrict__ bbox, unsigned long long* __restrict__ cnt_sum) {
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < cap; k += (int)(gridDim.x * blockDim.x)) {
        for (int a = 0; a < 3; ++a) { bbox[16 * k + a] = 0x7fffffff; bbox[16 * k + 3 + a] = -1; }
        for (int a = 0; a < 4; ++a) cnt_sum[8 * k + a] = 0ull;
    }
}

}  // namespace

namespace {
// The per-component statistics (bounding box, voxel count, coordinate sums) from the membership bits + the finished labels, queued
// right behind the labelling (one host round trip for both).  Only windows that hold members are looked at -- the separate pass of
// pb3d_component_stats_dev re-reads the whole 4 B/voxel label volume -- and the launch is a FEW persistent blocks: every block ends
// with a flush of its LDS table into the component records, and same-address global atomics serialise across the chip (gathering
// the statistics inside k_ccl_finish, one flush per four rows, cost 0.26 ms on the dome and 0.57 ms on the plinth of Taj 512).
// A wave takes 64 windows at a time (one lane each), then all its lanes work on each non-empty one: a segment of member voxels is one
// closed-form contribution of its first lane.
__global__ __launch_bounds__(256) void k_ccl_stats(const u64* __restrict__ bits, const int* __restrict__ labels, i64 nwords, pb3d_magic mP, pb3d_magic m1,
                                                   int A2, int cap, int* __restrict__ bbox, unsigned long long* __restrict__ cnt_sum) {
    __shared__ int slab[kFinSlots];
    __shared__ int slo[kFinSlots][3], shi[kFinSlots][3];
    __shared__ unsigned long long scs[kFinSlots][4];
    if (threadIdx.x < kFinSlots) {
        slab[threadIdx.x] = 0;
        for (int a = 0; a < 3; ++a) { slo[threadIdx.x][a] = 0x7fffffff; shi[threadIdx.x][a] = -1; }
        for (int a = 0; a < 4; ++a) scs[threadIdx.x][a] = 0ull;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const i64 nwaves = (i64)gridDim.x * 4, wid = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (i64 c0 = wid * 64; c0 < nwords; c0 += nwaves * 64) {
        const i64 idx = c0 + lane;
        const u64 mine = idx < nwords ? bits[idx] : 0ull;
        u64 todo = __ballot(mine != 0ull);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const u64 w = readlane64(mine, src);
            const u32 widx = (u32)(c0 + src);
            const u32 row = pb3d_div(widx, mP), t = widx - row * mP.d;
            const bool member = (w >> lane) & 1ull;
            const int L = member ? labels[(i64)row * A2 + 64 * (i64)t + lane] : 0;
            const bool seg = member && (lane == 0 || !((w >> (lane - 1)) & 1ull));
            if (!seg || L <= 0 || L > cap) continue;
            const u64 stop = ~w & ~le;
            const int len = (stop ? __ffsll((unsigned long long)stop) - 1 : 64) - lane;
            const u32 a0 = pb3d_div(row, m1), a1 = row - a0 * m1.d;
            const int a2 = 64 * (int)t + lane;
            const int lo[3] = {(int)a0, (int)a1, a2}, hi[3] = {(int)a0, (int)a1, a2 + len - 1};
            const unsigned long long cnt = (unsigned long long)len;
            const unsigned long long sm[3] = {(unsigned long long)a0 * cnt, (unsigned long long)a1 * cnt, (unsigned long long)(2 * a2 + len - 1) * cnt / 2ull};
            int slot = L & (kFinSlots - 1), found = -1;
            for (int k = 0; k < kFinSlots; ++k) {
                const int old = atomicCAS(&slab[slot], 0, L);
                if (old == 0 || old == L) { found = slot; break; }
                slot = (slot + 1) & (kFinSlots - 1);
            }
            if (found >= 0) {
                for (int a = 0; a < 3; ++a) { atomicMin(&slo[found][a], lo[a]); atomicMax(&shi[found][a], hi[a]); }
                atomicAdd(&scs[found][0], cnt);
                for (int a = 0; a < 3; ++a) atomicAdd(&scs[found][1 + a], sm[a]);
            } else {
                int* bb = bbox + 16 * (L - 1);
                for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], lo[a]); atomicMax(&bb[3 + a], hi[a]); }
                unsigned long long* cs = cnt_sum + 8 * (L - 1);
                atomicAdd(&cs[0], cnt);
                for (int a = 0; a < 3; ++a) atomicAdd(&cs[1 + a], sm[a]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kFinSlots && slab[threadIdx.x] > 0) {
        const int Lc = slab[threadIdx.x];
        int* bb = bbox + 16 * (Lc - 1);
        for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], slo[threadIdx.x][a]); atomicMax(&bb[3 + a], shi[threadIdx.x][a]); }
        unsigned long long* cs = cnt_sum + 8 * (Lc - 1);
        for (int a = 0; a < 4; ++a) atomicAdd(&cs[a], scs[threadIdx.x][a]);
    }
}

}  // namespace

static int label_color_impl(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                            int32_t* d_labels, int64_t* ncomp, int64_t cap, int64_t* bbox_lo_hi, int64_t* count, int64_t* coord_sum,
                            int* stats_valid, int C = 3, bool members_only = false) {
    PB3D_REQUIRE(ctx && color && ncomp && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_label_color: bad argument");
    const i64 n = A0 * A1 * A2;
    *ncomp = 0;
    if (stats_valid) *stats_valid = 0;
    if (n == 0) { if (stats_valid) *stats_valid = 1; return PB3D_OK; }
    PB3D_REQUIRE(n < (1ll << 31), "pb3d_label_color: grid too large for 32-bit labels");
    PB3D_REQUIRE(d_grid_rgb && d_labels, "pb3d_label_color: null buffer");
    const bool stats = stats_valid != nullptr && cap > 0;
    if (stats) PB3D_REQUIRE(bbox_lo_hi && count && coord_sum, "pb3d_label_color_stats: null output");
    const i64 rows = A0 * A1, P = (A2 + 63) / 64, nwords = rows * P;
    const i64 nchunks = (nwords + kWinPerBlock - 1) / kWinPerBlock;
    void *bits, *rootbits, *chunks;
    PB3D_TRY(pb3d_scratch(ctx, 42, (size_t)nwords * 8, &bits));          // a slot of its own: the bits outlive the call (ctx->ccl_last)
    ctx->ccl_last.valid = false;
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)nwords * 8, &rootbits));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)nchunks * 8 + 16, &chunks));
    const int dcap = stats ? (int)(cap < 16384 ? cap : 16384) : 0;
    void *sblk = nullptr, *sbb = nullptr, *scs = nullptr;
    PB3D_TRY(pb3d_scratch(ctx, 40, 64 + (size_t)16384 * 64, &sblk));
    sbb = (char*)sblk + 64;                 // int view of record k: sbb + 16 k ints
    scs = (char*)sblk + 64 + 32;            // u64 view of record k: scs + 8 k u64 (the second half of the record)
    u32* chunk_count = (u32*)chunks;
    u32* chunk_base = chunk_count + nchunks;
    i64* total = (i64*)sblk;                // (the header of the statistics block)
    const u32 color24 = C == 1 ? (u32)color[0] : ((u32)color[0] | ((u32)color[1] << 8) | ((u32)color[2] << 16));
    const pb3d_magic mP = pb3d_make_magic((u32)P), m1 = pb3d_make_magic((u32)A1);
    int* parent = (int*)d_labels;

    {
        const dim3 ig(pb3d_stream_blocks(ctx, rows, 4, 0));
        auto kern = C == 1 ? (members_only ? k_ccl_init<1, true> : k_ccl_init<1, false>) : (members_only ? k_ccl_init<3, true> : k_ccl_init<3, false>);
        hipLaunchKernelGGL(kern, ig, dim3(256), 0, ctx->stream, d_grid_rgb, rows, (int)A2, (int)P, color24, (u64*)bits, parent);
    }
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_merge, dim3(pb3d_stream_blocks(ctx, nwords, 256, 16)), dim3(256), 0, ctx->stream, (const u64*)bits, nwords, mP, m1, (int)A0,
                       (int)A1, (int)A2, parent);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_roots, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, (const u64*)bits, nwords, mP, (int)A2, (const int*)parent,
                       (u64*)rootbits, chunk_count);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)chunk_count, nchunks, chunk_base, total);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_number, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, nwords, mP, (int)A2, (const u64*)rootbits,
                       (const u32*)chunk_base, parent);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_finish<false>, dim3(pb3d_stream_blocks(ctx, rows, 4, 0)), dim3(256), 0, ctx->stream, (const u64*)bits, rows, (int)A2, (int)P,
                       parent, m1, 0, (int*)nullptr, (unsigned long long*)nullptr, 0);
    if (stats) {
        hipLaunchKernelGGL(k_fin_stats_init, dim3((unsigned)((dcap + 255) / 256)), dim3(256), 0, ctx->stream, dcap, (int*)sbb, (unsigned long long*)scs);
        const int sblocks = ctx->tune_misc[0] > 0 ? ctx->tune_misc[0] : 2 * ctx->cus;
        hipLaunchKernelGGL(k_ccl_stats, dim3((unsigned)sblocks), dim3(256), 0, ctx->stream, (const u64*)bits, (const int*)parent, nwords, mP, m1, (int)A2, dcap,
                           (int*)sbb, (unsigned long long*)scs);
    }
    PB3D_CHECK_LAUNCH();
    // the component count and -- optimistically -- the statistics of the first kFirst components come back in ONE copy, one round trip
    constexpr int kFirst = 64;
    struct Rec { int bb[8]; unsigned long long cs[4]; };
    struct Back { i64 nroots; i64 pad[7]; Rec rec[kFirst]; };
    static_assert(sizeof(Rec) == 64 && sizeof(Back) == 64 + 64 * kFirst, "the statistics block is 64-byte records");
    Back* hb = (Back*)((char*)ctx->pinned + 1024);
    static_assert(sizeof(Back) + 1024 + 64 <= (1 << 16), "the pinned area holds the read-back block");
    const int nf = dcap < kFirst ? dcap : kFirst;
    PB3D_HIP(hipMemcpyAsync(hb, sblk, 64 + (size_t)(stats ? nf : 0) * 64, hipMemcpyDeviceToHost, ctx->stream));
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
    const i64 nroots = hb->nroots;
    *ncomp = nroots;
    // the membership bits of THIS label volume stay where they are: a consumer that only needs the members' labels (recolouring) walks
    // the 1-bit-per-voxel array instead of the 4-byte-per-voxel one
    ctx->ccl_last.valid = true; ctx->ccl_last.labels = d_labels; ctx->ccl_last.bits = bits; ctx->ccl_last.rows = rows; ctx->ccl_last.A2 = A2;
    ctx->ccl_last.P = P; ctx->ccl_last.gen = ctx->scratch_gen; ctx->ccl_last.members_only = members_only;
    if (stats && nroots <= dcap) {
        auto put = [&](i64 k, const int* b6, const unsigned long long* c4) {
            for (int a = 0; a < 3; ++a) { bbox_lo_hi[6 * k + a] = b6[a]; bbox_lo_hi[6 * k + 3 + a] = (i64)b6[3 + a] + 1; }
            count[k] = (i64)c4[0];
            for (int a = 0; a < 3; ++a) coord_sum[3 * k + a] = (i64)c4[1 + a];
        };
        const i64 n0 = nroots < nf ? nroots : nf;
        for (i64 k = 0; k < n0; ++k) put(k, hb->rec[k].bb, hb->rec[k].cs);
        if (nroots > n0) {
            std::vector<Rec> more((size_t)(nroots - n0));
            PB3D_HIP(hipMemcpyAsync(more.data(), (const char*)sblk + 64 + 64 * n0, more.size() * sizeof(Rec), hipMemcpyDeviceToHost, ctx->stream));
            PB3D_HIP(hipStreamSynchronize(ctx->stream));
            for (i64 k = n0; k < nroots; ++k) put(k, more[(size_t)(k - n0)].bb, more[(size_t)(k - n0)].cs);
        }
        *stats_valid = 1;
    }
    return PB3D_OK;
}

extern "C" int pb3d_label_color_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                                    int32_t* d_labels, int64_t* ncomp) {
    return label_color_impl(ctx, d_grid_rgb, A0, A1, A2, color, d_labels, ncomp, 0, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int pb3d_label_color_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                                          int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                          int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_color_stats: null output");
    return label_color_impl(ctx, d_grid_rgb, A0, A1, A2, color, d_labels, ncomp, cap, bbox_lo_hi, count, coord_sum, stats_valid, 3, members_only != 0);
}

// the same on a 1-byte LABEL volume (row N3): components of the voxels whose label is `value`
extern "C" int pb3d_label_value_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, uint8_t value,
                                          int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                          int64_t* coord_sum, int* stats_valid) {
    PB3D_REQUIRE(stats_valid != nullptr, "pb3d_label_value_stats: null output");
    const uint8_t c3[3] = {value, 0, 0};
    return label_color_impl(ctx, d_grid_lab, A0, A1, A2, c3, d_labels, ncomp, cap, bbox_lo_hi, count, coord_sum, stats_valid, 1, members_only != 0);
}
