// K6: grid -> points.  Occupancy / colour-select with ORDERED stream compaction.
//
// reference utils/voxel_utils.py:7-21 (get_voxel_points_by_parts) and :35-51 (voxel_grid_to_points):
// both end in numpy.where(mask), i.e. the selected voxels in C order of (a0,a1,a2).  An unordered
// atomic append would break that, so compaction is a three-launch ordered scan: per-block popcounts
// (wave64 ballots) -> exclusive scan of the block counts -> fill at base + in-block rank.
#include "pb3d_internal.h"

namespace {

constexpr int kItems = 16;                 // 256-voxel sweeps per block
constexpr int kBlockVox = 256 * kItems;    // lattice voxels per block

struct SelParams {
    i64 A1, A2;        // grid dims (axis 1, 2)
    i64 L1, L2, nlat;  // lattice dims ([::s]) and total lattice voxels
    int C, ncolors, stride;
    u8 colors[3 * 32];
    u32 colors32[32];   // the same colours packed r | g << 8 | b << 16
    // colour-set membership as ONE perfect-hash probe per voxel (stride-1 kernels): entry = bits 31..24 of
    // mul24(colour, hashK); htab[entry] = colour << 8 for the set's colours, 1 (never equal) elsewhere
    u32 hashK;
    u32 htab[256];
    // exact u32 division by A2 and by A1 (Granlund-Montgomery round-up form): q = (t + ((n - t) >> s1)) >> s2, t = mulhi(m, n)
    u32 m2, m1;
    int s2a, s2b, s1a, s1b;
    int packed;        // 1: (a2, a1, a0) of a voxel fit 21 bits each (k_points_fill16 keeps per-group coordinates as one u64)
    int labelsel;      // C == 1 with a label set (row N3: 1-byte label volumes): selected = label in the set (htab[label] == 2)
};

struct Magic { u32 m; int sa, sb; };
inline Magic make_magic(u32 d) {     // 1 <= d < 2^31
    int L = 0;
    while ((1ull << L) < d) ++L;
    Magic g;
    g.m = (u32)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    g.sa = L < 1 ? L : 1;
    g.sb = L > 1 ? L - 1 : 0;
    return g;
}
__device__ __forceinline__ u32 magic_div(u32 n, u32 m, int sa, int sb) {
    const u32 t = __umulhi(m, n);
    return (t + ((n - t) >> sa)) >> sb;
}

__device__ __forceinline__ i64 lattice_to_voxel(const SelParams& p, i64 li, i64* i0, i64* i1, i64* i2) {
    if (p.stride == 1) {
        *i2 = -1;  // coordinates derived lazily by the caller
        return li;
    }
    const i64 a2 = li % p.L2;
    const i64 r = li / p.L2;
    const i64 a1 = r % p.L1, a0 = r / p.L1;
    *i0 = a0; *i1 = a1; *i2 = a2;
    return ((a0 * p.stride) * p.A1 + a1 * p.stride) * p.A2 + a2 * p.stride;
}

__device__ __forceinline__ bool selected(const SelParams& p, const u8* __restrict__ grid, i64 vox) {
    const u8* g = grid + vox * p.C;
    if (p.labelsel) {
        bool s = false;
        for (int k = 0; k < p.ncolors; ++k) s |= g[0] == p.colors[k];
        return s;
    }
    if (p.ncolors > 0) {
        const u8 r = g[0], gg = g[1], b = g[2];
        bool s = false;
        for (int k = 0; k < p.ncolors; ++k) s |= (r == p.colors[3 * k]) & (gg == p.colors[3 * k + 1]) & (b == p.colors[3 * k + 2]);
        return s;
    }
    bool s = false;
    for (int c = 0; c < p.C; ++c) s |= g[c] != 0;
    return s;
}

__global__ __launch_bounds__(256) void k_points_count(const u8* __restrict__ grid, SelParams p, u32* __restrict__ block_counts) {
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i64 base = (i64)blockIdx.x * kBlockVox;
    u32 cnt = 0;
    for (int it = 0; it < kItems; ++it) {
        const i64 li = base + it * 256 + threadIdx.x;
        bool sel = false;
        if (li < p.nlat) {
            i64 i0, i1, i2;
            sel = selected(p, grid, lattice_to_voxel(p, li, &i0, &i1, &i2));
        }
        cnt += (u32)__popcll(__ballot(sel));
    }
    if (lane == 0) wsum[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Exclusive scan of the nb block counts into 64-bit offsets, three small launches:
//   k_scan_local : 256 threads scan 1024 counts (one 16-byte load per thread) -> in-segment exclusive offsets + segment totals
//   k_scan_totals: one block scans the segment totals into 64-bit segment bases (+ grand total)
//   k_scan_add   : offsets[b] = base[b / 1024] + local[b]
constexpr int kSeg = 1024;

__global__ __launch_bounds__(256) void k_scan_local(const u32* __restrict__ counts, i64 nb, u32* __restrict__ local, u32* __restrict__ seg_total) {
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const i64 i0 = (i64)blockIdx.x * kSeg + 4 * threadIdx.x;
    u32 c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = i0 + k < nb ? counts[i0 + k] : 0u;
    const u32 mine = c[0] + c[1] + c[2] + c[3];
    u32 inc = mine;
    for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 before = inc - mine;
    for (int k = 0; k < wv; ++k) before += wsum[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < nb) local[i0 + k] = before; before += c[k]; }
    if (threadIdx.x == 255) seg_total[blockIdx.x] = before;
}

__global__ __launch_bounds__(1024) void k_scan_totals(const u32* __restrict__ seg_total, i64 nseg, i64* __restrict__ seg_base, i64* __restrict__ total) {
    __shared__ i64 part[1024];
    const i64 per = (nseg + 1023) / 1024;
    const i64 b = (i64)threadIdx.x * per, e = b + per < nseg ? b + per : nseg;
    i64 s = 0;
    for (i64 i = b; i < e; ++i) s += seg_total[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        i64 run = 0;
        for (int i = 0; i < 1024; ++i) { const i64 v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    i64 run = part[threadIdx.x];
    for (i64 i = b; i < e; ++i) { seg_base[i] = run; run += seg_total[i]; }
}

__global__ __launch_bounds__(256) void k_scan_add(const u32* __restrict__ local, const i64* __restrict__ seg_base, i64 nb, i64* __restrict__ offsets) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (i64)gridDim.x * blockDim.x) offsets[i] = seg_base[i / kSeg] + local[i];
}

__global__ __launch_bounds__(256) void k_points_fill(const u8* __restrict__ grid, SelParams p, const i64* __restrict__ block_off,
                                                     float* __restrict__ pts, u8* __restrict__ cols) {
    __shared__ u32 segoff[kItems * 4 + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i64 base = (i64)blockIdx.x * kBlockVox;
    u64 bal[kItems];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const i64 li = base + it * 256 + threadIdx.x;
        bool sel = false;
        if (li < p.nlat) {
            i64 i0, i1, i2;
            sel = selected(p, grid, lattice_to_voxel(p, li, &i0, &i1, &i2));
        }
        bal[it] = __ballot(sel);
        if (lane == 0) segoff[it * 4 + w] = (u32)__popcll(bal[it]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int i = 0; i < kItems * 4; ++i) { const u32 v = segoff[i]; segoff[i] = run; run += v; }
    }
    __syncthreads();
    const i64 out0 = block_off[blockIdx.x];
    const u64 lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        if (!((bal[it] >> lane) & 1)) continue;
        const i64 li = base + it * 256 + threadIdx.x;
        const i64 a2 = li % p.L2;
        const i64 r = li / p.L2;
        const i64 a1 = r % p.L1, a0 = r / p.L1;
        const i64 vox = ((a0 * p.stride) * p.A1 + a1 * p.stride) * p.A2 + a2 * p.stride;
        const i64 pos = out0 + segoff[it * 4 + w] + __popcll(bal[it] & lt);
        const float s = (float)p.stride;
        pts[3 * pos + 0] = __fmul_rn((float)a2, s);
        pts[3 * pos + 1] = __fmul_rn((float)a1, s);
        pts[3 * pos + 2] = __fmul_rn((float)a0, s);
        for (int c = 0; c < p.C; ++c) cols[p.C * pos + c] = grid[vox * p.C + c];
    }
}

// ------------------------------------------------------------------------------------------------
// stride == 1, RGB grids: 16 consecutive voxels per thread (48 bytes as three 16-byte loads), so a
// block still covers kBlockVox = 4096 voxels in lattice order and shares the block-count scan with the
// generic kernels.  In-block order = thread order (thread t owns voxels 16t .. 16t+15).
// ------------------------------------------------------------------------------------------------
typedef u32 u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 select16(const SelParams& p, const u32* htab, const u8* __restrict__ grid, i64 v0, i64 nvox, u32 w[12]) {
    // loads the 48 bytes of voxels v0..v0+15 (bounds-checked at the grid's end) and returns their 16 select bits
    if (v0 + 16 <= nvox) {
        const u32x4v* g = (const u32x4v*)(grid + 3 * v0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const u32x4v t = g[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            u32 t = 0;
            for (int b = 0; b < 4; ++b) { const i64 o = 3 * v0 + 4 * k + b; if (o < 3 * nvox) t |= (u32)grid[o] << (8 * b); }
            w[k] = t;
        }
    }
    u32 bits = 0;
    if (p.ncolors > 0 && p.hashK) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = (3 * i) >> 2, sh = (3 * i) & 3;
            const u32 v = __builtin_amdgcn_alignbyte(j + 1 < 12 ? w[j + 1] : 0u, w[j], (u32)sh);   // top byte: the next voxel's red
            const u32 e = (__umul24(v, p.hashK) >> 22) & 0x3fcu;                                  // mul24 ignores the top byte
            bits |= (u32)((v << 8) == *(const u32*)((const u8*)htab + e)) << i;
        }
    } else if (p.ncolors > 0) {                  // no collision-free multiplier found (practically never): compare loop
        u32 vox[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = (3 * i) >> 2, sh = (3 * i) & 3;
            vox[i] = __builtin_amdgcn_alignbyte(j + 1 < 12 ? w[j + 1] : 0u, w[j], (u32)sh) & 0x00ffffffu;
        }
        for (int k = 0; k < p.ncolors; ++k) {
            const u32 ck = p.colors32[k];
#pragma unroll
            for (int i = 0; i < 16; ++i) bits |= (u32)(vox[i] == ck) << i;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = (3 * i) >> 2, sh = (3 * i) & 3;
            bits |= (u32)((__builtin_amdgcn_alignbyte(j + 1 < 12 ? w[j + 1] : 0u, w[j], (u32)sh) << 8) != 0u) << i;
        }
    }
    const i64 left = nvox - v0;
    if (left < 16) bits &= (1u << left) - 1u;
    return bits;
}

// occupancy grids (C == 1): the 16 voxels are one 16-byte load; selected = byte != 0
__device__ __forceinline__ u32 select16_occ(const u8* __restrict__ grid, i64 v0, i64 nvox, u32 w[4], const u32* htab = nullptr) {
    if (v0 + 16 <= nvox) {
        const u32x4v t = *(const u32x4v*)(grid + v0);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 t = 0;
            for (int b = 0; b < 4; ++b) { const i64 o = v0 + 4 * k + b; if (o < nvox) t |= (u32)grid[o] << (8 * b); }
            w[k] = t;
        }
    }
    u32 bits = 0;
    if (htab) {          // label volumes: one table read per voxel (htab[label] == 2 for the labels of the set)
#pragma unroll
        for (int i = 0; i < 16; ++i) bits |= (u32)(htab[(w[i >> 2] >> (8 * (i & 3))) & 0xffu] == 2u) << i;
        const i64 left = nvox - v0;
        if (left < 16) bits &= (1u << left) - 1u;
        return bits;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) bits |= (u32)(((w[i >> 2] >> (8 * (i & 3))) & 0xffu) != 0u) << i;
    return bits;
}

// masks (one u16 per thread = per 16 voxels, n / 8 bytes in all): the fill pass takes the selection from them instead of recomputing it,
// and loads the grid only where something is selected -- an empty 128-byte line is then read once (here), not twice.
template <int C>
__global__ __launch_bounds__(256) void k_points_count16(const u8* __restrict__ grid, SelParams p, u32* __restrict__ block_counts,
                                                        unsigned short* __restrict__ masks) {
    __shared__ u32 wsum[4];
    __shared__ u32 htab[256];
    htab[threadIdx.x] = p.htab[threadIdx.x];
    __syncthreads();
    const i64 v0 = (i64)blockIdx.x * kBlockVox + 16 * threadIdx.x;
    u32 w[12];
    const u32 sel = v0 < p.nlat ? (C == 3 ? select16(p, htab, grid, v0, p.nlat, w) : select16_occ(grid, v0, p.nlat, w, p.labelsel ? htab : nullptr)) : 0u;
    masks[(i64)blockIdx.x * 256 + threadIdx.x] = (unsigned short)sel;
    u32 c = (u32)__popc(sel);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Phase 1: every thread ranks its selected voxels inside the block and drops (local index, colour) records into LDS
// in output order.  Phase 2: the block writes the compacted points and colours; coordinates come from the block's
// base (a0,a1,a2) plus the local index with exact multiply-high divisions (no integer divide in the loop), colour
// bytes leave as whole dwords cut out of two neighbouring 24-bit records (C == 1: four one-byte records per dword).
// SINGLE = true: the one-pass form for resident callers (pb3d_points_extract_dev).  There is no count pass and no scanned
// offset table: a workgroup counts its own selection, publishes it, and obtains the sum of its predecessors (blockIdx order) by a
// decoupled look-back over 64-bit status words (flag << 62 | value; flag 1 = this block's count, 2 = inclusive prefix).  A status
// word is complete in itself, so relaxed agent-scope atomics are the whole protocol (no data is handed over behind it).
// Ordering: a block only ever waits for blocks with SMALLER ids.  The dispatcher deals blocks to the XCDs round-robin and each XCD
// starts its share in increasing order, so the smallest unfinished id is always resident and waits for nobody: the chain drains.
// (A ticket counter would make that a theorem instead of an observation, but 262 144 atomics on one address cost 2.6 ms at
// 1024^3 -- more than the whole pass.)  HIP promises no dispatch order, so every spin is BOUNDED: a block that gives up publishes a
// poison word (flag 3), everyone behind it aborts at once, and the host falls back to the two-pass protocol.
struct ScanState {
    unsigned long long* status;     // one word per block, zeroed before the launch
    i64* total;                     // out: number of selected voxels, or -1 when the look-back was abandoned
    i64 capacity;                   // rows the output buffers can take; blocks that would write past it write nothing
};
constexpr unsigned long long kFlagA = 1ull << 62, kFlagP = 2ull << 62, kFlagX = 3ull << 62, kValMask = (1ull << 62) - 1;
constexpr int kSpinLimit = 1 << 22;   // ~ seconds: far beyond any legitimate wait (a predecessor's count takes microseconds)

template <int C, bool SINGLE>
__global__ __launch_bounds__(256) void k_points_fill16(const u8* __restrict__ grid, SelParams p, const i64* __restrict__ block_off,
                                                       float* __restrict__ pts, u8* __restrict__ cols, ScanState st,
                                                       const unsigned short* __restrict__ masks) {
    __shared__ u32 wsum[4];
    __shared__ u32 htab[SINGLE ? 256 : 1];              // the two-pass form takes the selection from the count pass's masks
    __shared__ unsigned short lidx[kBlockVox];
    __shared__ __attribute__((aligned(16))) u32 lrec[kBlockVox + 8];
    __shared__ unsigned long long gcoord[64];           // (a2 | a1 << 21 | a0 << 42) of the first voxel of every 64-voxel group of the block (512 bytes: a 2 KB table cost a workgroup per CU)
    __shared__ i64 sh_excl;
    if (SINGLE) { htab[threadIdx.x] = p.htab[threadIdx.x]; __syncthreads(); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 bid = blockIdx.x;
    const i64 base = (i64)bid * kBlockVox;
    const i64 v0 = base + 16 * threadIdx.x;
    // coordinates of a point = its group's first voxel + the position in the group (one pair of divisions per GROUP, not per point)
    const bool packed_coords = p.packed != 0;             // decided on the host (make_params): every axis < 2^21, A2 >= 64, fewer than 2^32 voxels
    if (packed_coords && (threadIdx.x & 3) == 0) {
        const u32 r = magic_div((u32)v0, p.m2, p.s2a, p.s2b), g2 = (u32)v0 - r * (u32)p.A2;
        const u32 q = magic_div(r, p.m1, p.s1a, p.s1b), g1 = r - q * (u32)p.A1;
        gcoord[threadIdx.x >> 2] = (unsigned long long)g2 | ((unsigned long long)g1 << 21) | ((unsigned long long)q << 42);
    }
    u32 w[12];
    u32 bits;
    if (SINGLE) bits = v0 < p.nlat ? (C == 3 ? select16(p, htab, grid, v0, p.nlat, w) : select16_occ(grid, v0, p.nlat, w, p.labelsel ? htab : nullptr)) : 0u;
    else {
        // the count pass left the selection: only the voxels' bytes (their colours) are needed, and only where something is selected
        bits = masks[(i64)bid * 256 + threadIdx.x];
        if (bits) {
            if (v0 + 16 <= p.nlat) {
                const u32x4v* g = (const u32x4v*)(grid + (C == 3 ? 3 : 1) * v0);
#pragma unroll
                for (int q = 0; q < (C == 3 ? 3 : 1); ++q) { const u32x4v t = g[q]; w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w; }
            } else {                                    // the grid's ragged end: bounds-checked byte loads
#pragma unroll
                for (int k2 = 0; k2 < (C == 3 ? 12 : 4); ++k2) {
                    u32 t = 0;
                    for (int b = 0; b < 4; ++b) { const i64 o = (C == 3 ? 3 : 1) * v0 + 4 * k2 + b; if (o < (C == 3 ? 3 : 1) * p.nlat) t |= (u32)grid[o] << (8 * b); }
                    w[k2] = t;
                }
            }
        }
    }
    const u32 c = (u32)__popc(bits);
    u32 inc = c;
    for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 k = inc - c;
    for (int q = 0; q < wv; ++q) k += wsum[q];
    const u32 total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (SINGLE) {
        // Publish this block's count at once, then look back with the WHOLE workgroup: 1024 predecessors per round (four status
        // words per lane, nearest first), because with ~1500 blocks in flight the nearest inclusive prefix is that far back and a
        // round trip to another XCD's flag costs microseconds -- a one-wave walk of 64 per round waited longer than the block
        // takes to do its work.  Wave w owns the predecessors 256 w .. 256 w + 255 behind the block; each wave reduces its own
        // slice (sum up to and including its first inclusive prefix), lane 0 of wave 0 combines the four in order.
        __shared__ i64 part_sum[4];
        __shared__ int part_stop[4];          // 1: this wave's slice held an inclusive prefix; 2: poison
        if (threadIdx.x == 0) __hip_atomic_store(&st.status[bid], (bid == 0 ? kFlagP : kFlagA) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i64 excl = 0;
        bool poisoned = false, found = bid == 0;
        for (i64 j0 = (i64)bid - 1; !found && !poisoned; j0 -= 1024) {
            i64 wsum_ = 0;
            int wstop = 0;
            // the four status words of this lane are requested together (one round trip, not four); only words that are still
            // empty are polled again
            unsigned long long sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i64 idx = j0 - 256 * wv - 64 * q - lane;
                sv[q] = idx >= 0 ? __hip_atomic_load(&st.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagP;   // past block 0: an empty prefix
            }
#pragma unroll
            for (int q = 0; q < 4 && !wstop; ++q) {
                const i64 idx = j0 - 256 * wv - 64 * q - lane;
                unsigned long long s_ = sv[q];
                int spins = 0;
                while (!(s_ >> 62)) {
                    __builtin_amdgcn_s_sleep(2);
                    s_ = ++spins > kSpinLimit ? kFlagX : __hip_atomic_load(&st.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__ballot((s_ >> 62) == 3ull)) { wstop = 2; break; }
                const u64 pm = __ballot((s_ >> 62) == 2ull);
                const int stop = pm ? __builtin_ctzll(pm) : 64;
                i64 v = lane <= stop ? (i64)(s_ & kValMask) : 0;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                wsum_ += v;
                if (pm) wstop = 1;
            }
            if (lane == 0) { part_sum[wv] = wsum_; part_stop[wv] = wstop; }
            __syncthreads();
            for (int w2 = 0; w2 < 4; ++w2) {                       // every thread forms the same combination
                if (part_stop[w2] == 2) { poisoned = true; break; }
                excl += part_sum[w2];
                if (part_stop[w2] == 1) { found = true; break; }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (poisoned) {
                __hip_atomic_store(&st.status[bid], kFlagX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *st.total = -1;
            } else {
                if (bid != 0) __hip_atomic_store(&st.status[bid], kFlagP | (unsigned long long)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (base + kBlockVox >= p.nlat) *st.total = excl + total;      // the last block knows the grand total
            }
            sh_excl = poisoned ? -1 : excl;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if ((bits >> i) & 1u) {
            const int j = (3 * i) >> 2, sh = (3 * i) & 3;
            lidx[k] = (unsigned short)(16 * threadIdx.x + i);
            lrec[k] = C == 3 ? __builtin_amdgcn_alignbyte(j + 1 < 12 ? w[j + 1] : 0u, w[j], (u32)sh) & 0x00ffffffu
                             : (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
            ++k;
        }
    }
    __syncthreads();
    const i64 out0 = SINGLE ? sh_excl : block_off[blockIdx.x];
    if (SINGLE && (out0 < 0 || out0 + total > st.capacity)) return;   // abandoned, or the caller's buffers are too small: write nothing
    // block base -> (b0, b1, b2); uniform, once per thread
    u32 b2, b1; i64 b0;
    if (p.nlat <= 0xffffffffll) {
        const u32 r = magic_div((u32)base, p.m2, p.s2a, p.s2b);
        b2 = (u32)base - r * (u32)p.A2;
        const u32 q = magic_div(r, p.m1, p.s1a, p.s1b);
        b1 = r - q * (u32)p.A1; b0 = q;
    } else {
        const i64 r = base / p.A2;
        b2 = (u32)(base - r * p.A2); b0 = r / p.A1; b1 = (u32)(r - b0 * p.A1);
    }
    // Points leave as whole 16-byte vectors: a dword store per float at a 12-byte stride touches every 128-byte line with three
    // instructions of 16-of-48 bytes each (what bounded this kernel); the coordinates of 512 points at a time are staged in LDS at
    // the phase of the output address, so that an LDS piece IS an aligned 16-byte piece of the output.  Only the first and last
    // piece of a chunk are partial (scalar stores of the floats that belong to this block).
    __shared__ __attribute__((aligned(16))) float lp[4 + 3 * 512 + 4];
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    for (u32 c0 = 0; c0 < total; c0 += 512) {
        const u32 nc = total - c0 < 512u ? total - c0 : 512u;
        float* gfirst = pts + 3 * (out0 + c0);                          // first float of this chunk
        const u32 shift = (u32)(((uintptr_t)gfirst >> 2) & 3u);          // floats past a 16-byte boundary
        for (u32 q = threadIdx.x; q < nc; q += 256) {
            const u32 li = lidx[c0 + q];
            u32 a2, a1; i64 a0;
            if (packed_coords) {
                const unsigned long long gc = gcoord[li >> 6];
                a2 = ((u32)gc & 0x1fffffu) + (li & 63u); a1 = (u32)(gc >> 21) & 0x1fffffu; a0 = (i64)(gc >> 42);
                if (a2 >= (u32)p.A2) { a2 -= (u32)p.A2; if (++a1 >= (u32)p.A1) { a1 = 0; ++a0; } }      // the group runs over a row end (A2 >= 64: once)
            } else {
                const u32 x = b2 + li;                                  // < A2 + 4096
                const u32 q2 = magic_div(x, p.m2, p.s2a, p.s2b);
                a2 = x - q2 * (u32)p.A2;
                const u32 y = b1 + q2;                                  // < A1 + 4096
                const u32 q1 = magic_div(y, p.m1, p.s1a, p.s1b);
                a1 = y - q1 * (u32)p.A1;
                a0 = b0 + q1;
            }
            float* l = lp + shift + 3 * q;
            l[0] = (float)a2; l[1] = (float)a1; l[2] = (float)a0;
        }
        __syncthreads();
        float* gbase = gfirst - shift;                                  // 16-byte aligned; lp[i] <-> gbase[i]
        const u32 nfl = shift + 3 * nc;
        for (u32 pz = threadIdx.x; 4 * pz < nfl; pz += 256) {
            const u32 f0 = 4 * pz < shift ? shift : 4 * pz, f1 = 4 * pz + 4 < nfl ? 4 * pz + 4 : nfl;
            if (f1 - f0 == 4) *(f32x4v*)(gbase + 4 * pz) = *(const f32x4v*)(lp + 4 * pz);
            else for (u32 f = f0; f < f1; ++f) gbase[f] = lp[f];
        }
        __syncthreads();
    }
    // colours: head bytes up to the first dword boundary of the output, then whole dwords, then the tail
    u8* co = cols + C * out0;
    const u32 nbytes = C * total;
    const u32 head = (u32)((4 - ((uintptr_t)co & 3u)) & 3u);
    const u32 hb = head < nbytes ? head : nbytes;
    auto stream_byte = [&](u32 q) -> u8 {
        if (C == 1) return (u8)lrec[q];
        const u32 r = (q * 43691u) >> 17; return (u8)(lrec[r] >> (8 * (q - 3 * r)));
    };
    if (threadIdx.x < hb) co[threadIdx.x] = stream_byte(threadIdx.x);
    const u32 ndw = (nbytes - hb) / 4;
    u32* cw = (u32*)(co + hb);
    if (C == 3) {
        // three dwords = four records at a time: stream bytes hb + 12 g ... start in record r0 = g * 4 + hb / 3 at phase ph = hb % 3 (the
        // same for every group of the block); the 24-bit records are strung into dwords W0..W3 and cut at the phase with v_alignbyte
        const u32 r00 = hb / 3, ph = hb - 3 * r00;
        const u32 ng = ndw / 3;
        for (u32 g = threadIdx.x; g < ng; g += 256) {
            // records r00 + 4 g .. + 4 out of two aligned 16-byte LDS reads (five dword reads at a stride of 4 dwords were 4-way bank conflicts)
            const u32x4v va = *(const u32x4v*)(lrec + 4 * g), vb = *(const u32x4v*)(lrec + 4 * g + 4);      // lrec has 8 words of slack
            const u32 e0 = r00 ? va.y : va.x, e1 = r00 ? va.z : va.y, e2 = r00 ? va.w : va.z, e3 = r00 ? vb.x : va.w, e4 = r00 ? vb.y : vb.x;
            const u32 W0 = e0 | (e1 << 24), W1 = (e1 >> 8) | (e2 << 16), W2 = (e2 >> 16) | (e3 << 8), W3 = e4;
            cw[3 * g] = __builtin_amdgcn_alignbyte(W1, W0, ph); cw[3 * g + 1] = __builtin_amdgcn_alignbyte(W2, W1, ph);
            cw[3 * g + 2] = __builtin_amdgcn_alignbyte(W3, W2, ph);
        }
        for (u32 d = 3 * ng + threadIdx.x; d < ndw; d += 256) {     // the last one or two dwords
            const u32 q = hb + 4 * d;
            const u32 r = (q * 43691u) >> 17, ph2 = q - 3 * r;      // q / 3 exactly (q < 2^16)
            const unsigned long long two = (unsigned long long)lrec[r] | ((unsigned long long)lrec[r + 1] << 24);
            cw[d] = (u32)(two >> (8 * ph2));
        }
    } else {
        for (u32 d = threadIdx.x; d < ndw; d += 256) {
            const u32 q = hb + 4 * d;                               // first stream byte of this dword
            cw[d] = lrec[q] | (lrec[q + 1] << 8) | (lrec[q + 2] << 16) | (lrec[q + 3] << 24);
        }
    }
    const u32 tail0 = hb + 4 * ndw;
    if (threadIdx.x < nbytes - tail0) co[tail0 + threadIdx.x] = stream_byte(tail0 + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The fill pass of the two-pass protocol, WAVE-PRIVATE form (round 4).  k_points_fill16 above ranks, stages and writes per BLOCK:
// rank -> LDS -> coordinates of 512 points at a time -> colours, five block barriers and 31 KB of LDS (five blocks per CU), and the
// SQ counters of round 2 / the 4-bit colour-id experiment of round 3 showed the pass bound by that chain, not by its bytes.
// Here every wavefront is on its own: it owns 1024 consecutive voxels (a quarter of a count-pass block), finds its output offset
// from the block's scanned offset + the popcounts of the earlier waves' masks, and walks its voxels in four rounds of 256 -- lane l
// takes voxels 4 l .. 4 l + 3 of the round, so a round's points are contiguous in the output and every lane takes part in every
// step.  Per round: rank inside the wave (one shuffle scan), coordinates from the lane's first voxel (one pair of exact
// multiply-high divisions per LANE and round, then increments), the points' floats scattered into a wave-private LDS window at the
// phase of the output address, whole 16-byte pieces stored (1 KB contiguous per instruction), the last partial piece carried into
// the next round; the colours likewise as 24-bit records strung into aligned dwords, four records = three dwords at a time.
// No block barrier, 4.2 KB of LDS per wave.  All four rounds' loads (12 bytes per lane where something is selected) are in flight
// before the first is used.
// ------------------------------------------------------------------------------------------------
typedef u32 u32x3v_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int C>
__global__ __launch_bounds__(256) void k_points_fillw(const u8* __restrict__ grid, SelParams p, const i64* __restrict__ block_off,
                                                      float* __restrict__ pts, u8* __restrict__ cols, const unsigned short* __restrict__ masks) {
    constexpr int FW = 4 + 3 * 256 + 4;                  // floats: <= 3 carried + 768 of a round
    constexpr int RW = 8 + 256 + 8;                      // records: <= 8 waiting + 256 of a round + the over-read of the last group
    __shared__ __attribute__((aligned(16))) float fwin[4][FW];
    __shared__ __attribute__((aligned(16))) u32 rwin[4][RW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 bid = blockIdx.x;
    const i64 wbase = (i64)bid * kBlockVox + 1024 * wv;
    if (wbase >= p.nlat) return;                          // (whole wave; the kernel has no block barrier)
    const unsigned short* bm = masks + (i64)bid * 256;
    const u32 mymask = bm[64 * wv + lane];                // selection of voxels wbase + 16 lane .. + 15 (bits past the grid's end are zero)
    u32 pre = 0;
    for (int q = 0; q < wv; ++q) pre += (u32)__popc((u32)bm[64 * q + lane]);
    u32 tot = (u32)__popc(mymask);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); tot += __shfl_xor(tot, o); }
    if (tot == 0) return;
    const i64 out0 = block_off[bid] + pre;
    float* fw = fwin[wv];
    u32* rw = rwin[wv];
    // ---- all loads of the wave
    u32 sel[4];
    u32 w[4][C == 3 ? 3 : 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u32 m = (u32)__shfl((int)mymask, 16 * r + (lane >> 2));
        sel[r] = (m >> (4 * (lane & 3))) & 0xfu;
#pragma unroll
        for (int k = 0; k < (C == 3 ? 3 : 1); ++k) w[r][k] = 0u;
        if (sel[r]) {
            const i64 v = wbase + 256 * r + 4 * lane;
            if (v + 4 <= p.nlat) {
                if (C == 3) { const u32x3v_a4 t = *(const u32x3v_a4*)(grid + 3 * v); w[r][0] = t.x; w[r][1] = t.y; w[r][2] = t.z; }
                else w[r][0] = *(const u32*)(grid + v);
            } else {                                      // the grid's ragged end
                for (int b = 0; b < 4 * (C == 3 ? 3 : 1); ++b)
                    if (C * v + b < C * p.nlat) w[r][b >> 2] |= (u32)grid[C * v + b] << (8 * (b & 3));
            }
        }
    }
    // ---- the wave's first voxel -> (b0, b1, b2)   (nlat < 2^32: the launcher's condition)
    u32 b0, b1, b2;
    {
        const u32 rr = magic_div((u32)wbase, p.m2, p.s2a, p.s2b);
        b2 = (u32)wbase - rr * (u32)p.A2;
        const u32 q = magic_div(rr, p.m1, p.s1a, p.s1b);
        b1 = rr - q * (u32)p.A1; b0 = q;
    }
    // ---- output streams
    float* gout = pts + 3 * out0;
    const u32 shift = (u32)(((uintptr_t)gout >> 2) & 3u);           // floats past a 16-byte boundary
    float* gal = gout - shift;                                      // aligned; fw[i] <-> gal[i]
    u32 pend = shift, first_lo = shift;                             // floats [0, first_lo) of the first piece belong to the wave before
    u8* co = cols + (i64)C * out0;
    const u32 nbytes = (u32)C * tot;
    const u32 head = (u32)((4 - ((uintptr_t)co & 3u)) & 3u);
    const u32 hb = head < nbytes ? head : nbytes;                   // bytes up to the first dword boundary of the colour stream
    u32* cw = (u32*)(co + hb);
    // records: record k of the wave sits at window index k + roff until the first compaction; groups start at multiples of 4
    const u32 r00 = C == 3 ? hb / 3 : hb, ph = C == 3 ? hb - 3 * r00 : 0u;      // first record / byte phase of the aligned dwords
    const u32 roff = (4u - r00) & 3u;
    u32 rfill = roff, gidx = r00 ? 4u : 0u, gdone = 0;
    bool head_done = hb == 0;
    auto emit_head = [&]() {
        if ((u32)lane < hb) co[lane] = C == 3 ? (u8)(rw[roff] >> (8 * lane)) : (u8)rw[roff + lane];
        head_done = true;
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u32 c = (u32)__popc(sel[r]);
        u32 inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const u32 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        const u32 n = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        if (n == 0) continue;
        // ---- scatter: coordinates and colour records of this lane's selected voxels
        {
            const u32 x = b2 + 256u * r + 4u * lane;                 // < A2 + 1024
            const u32 q2 = magic_div(x, p.m2, p.s2a, p.s2b);
            u32 a2 = x - q2 * (u32)p.A2;
            const u32 y = b1 + q2;
            const u32 q1 = magic_div(y, p.m1, p.s1a, p.s1b);
            u32 a1 = y - q1 * (u32)p.A1;
            u32 a0 = b0 + q1;
            u32 k = inc - c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((sel[r] >> i) & 1u) {
                    float* l = fw + pend + 3 * k;
                    l[0] = (float)a2; l[1] = (float)a1; l[2] = (float)a0;
                    u32 rec;
                    if (C == 3) {
                        const int j = (3 * i) >> 2, sh = (3 * i) & 3;
                        rec = __builtin_amdgcn_alignbyte(j + 1 < 3 ? w[r][j + 1 < 3 ? j + 1 : 0] : 0u, w[r][j], (u32)sh) & 0x00ffffffu;
                    } else rec = (w[r][0] >> (8 * i)) & 0xffu;
                    rw[rfill + k] = rec;
                    ++k;
                }
                if (++a2 == (u32)p.A2) { a2 = 0; if (++a1 == (u32)p.A1) { a1 = 0; ++a0; } }
            }
        }
        wave_sync();
        // ---- points: whole 16-byte pieces leave, the rest is carried
        {
            const u32 nfl = pend + 3 * n, npc = nfl >> 2, left = nfl & 3u;
            for (u32 pz = lane; pz < npc; pz += 64) {
                if (pz == 0 && first_lo) { for (u32 f = first_lo; f < 4; ++f) gal[f] = fw[f]; }
                else *(f32x4w*)(gal + 4 * pz) = *(const f32x4w*)(fw + 4 * pz);      // (nontemporal stores measured slower here: fill 1.47 -> 1.62 ms at 1024^3)
            }
            if (npc) {
                first_lo = 0;
                float t = 0.f;
                if ((u32)lane < left) t = fw[4 * npc + lane];
                wave_sync();
                if ((u32)lane < left) fw[lane] = t;
                gal += 4 * npc; pend = left;
            } else pend = nfl;
        }
        // ---- colours: groups of four records = C dwords whose records (C == 3: and the one behind them) have arrived
        {
            rfill += n;
            const u32 need = C == 3 ? 1u : 0u;
            const u32 ng = rfill >= gidx + need ? (rfill - gidx - need) >> 2 : 0u;
            if (ng) {
                if (!head_done) emit_head();
                for (u32 g = lane; g < ng; g += 64) {
                    const u32x4v va = *(const u32x4v*)(rw + gidx + 4 * g);
                    if (C == 3) {
                        const u32 e4 = rw[gidx + 4 * g + 4];
                        const u32 W0 = va.x | (va.y << 24), W1 = (va.y >> 8) | (va.z << 16), W2 = (va.z >> 16) | (va.w << 8), W3 = e4;
                        u32* d = cw + 3 * (gdone + g);
                        d[0] = __builtin_amdgcn_alignbyte(W1, W0, ph); d[1] = __builtin_amdgcn_alignbyte(W2, W1, ph);
                        d[2] = __builtin_amdgcn_alignbyte(W3, W2, ph);
                    } else cw[gdone + g] = va.x | (va.y << 8) | (va.z << 16) | (va.w << 24);
                }
                const u32 src = gidx + 4 * ng, rem = rfill - src;      // <= 4 records wait for the next round
                u32 t = 0;
                if ((u32)lane < rem) t = rw[src + lane];
                wave_sync();
                if ((u32)lane < rem) rw[lane] = t;
                gdone += ng; gidx = 0; rfill = rem;
            }
        }
        wave_sync();
    }
    // ---- the ends of the two streams
    if ((u32)lane >= first_lo && (u32)lane < pend) gal[lane] = fw[lane];
    if (!head_done) emit_head();
    {
        const u32 q0 = hb + 4 * (u32)C * gdone;               // first stream byte that has not left
        if (q0 < nbytes && (u32)lane < nbytes - q0) {
            if (C == 3) { const u32 bo = ph + lane, rr = (bo * 43691u) >> 17; co[q0 + lane] = (u8)(rw[gidx + rr] >> (8 * (bo - 3 * rr))); }
            else co[q0 + lane] = (u8)rw[gidx + lane];
        }
    }
}

int make_params(i64 A0, i64 A1, i64 A2, int C, const u8* colors, int ncolors, int stride, SelParams* p) {
    PB3D_REQUIRE(A0 >= 0 && A1 >= 0 && A2 >= 0 && (C == 1 || C == 3), "pb3d_points: bad shape (%lld,%lld,%lld,%d)",
                 (long long)A0, (long long)A1, (long long)A2, C);
    PB3D_REQUIRE(stride >= 1, "pb3d_points: stride must be >= 1");
    PB3D_REQUIRE(ncolors >= 0 && ncolors <= 32, "pb3d_points: at most 32 colours");
    PB3D_REQUIRE(ncolors == 0 || colors, "pb3d_points: null colour set");      // C == 1: `colors` holds ncolors 1-byte LABELS
    p->A1 = A1; p->A2 = A2;
    p->packed = (stride == 1 && A0 < (1 << 21) && A1 < (1 << 21) && A2 < (1 << 21) && A2 >= 64 && A0 * A1 * A2 <= 0xffffffffll) ? 1 : 0;
    const i64 L0 = (A0 + stride - 1) / stride;
    p->L1 = (A1 + stride - 1) / stride; p->L2 = (A2 + stride - 1) / stride;
    p->nlat = L0 * p->L1 * p->L2;
    p->C = C; p->ncolors = ncolors; p->stride = stride;
    memset(p->colors, 0, sizeof(p->colors));
    p->labelsel = (C == 1 && ncolors > 0) ? 1 : 0;
    if (p->labelsel) {
        memcpy(p->colors, colors, (size_t)ncolors);
        for (int k = 0; k < 32; ++k) p->colors32[k] = 0xffffffffu;
        p->hashK = 0;
        for (int e = 0; e < 256; ++e) p->htab[e] = 1u;
        for (int k = 0; k < ncolors; ++k) p->htab[colors[k]] = 2u;
        const Magic h2 = make_magic((u32)(A2 > 0 ? A2 : 1)), h1 = make_magic((u32)(A1 > 0 ? A1 : 1));
        p->m2 = h2.m; p->s2a = h2.sa; p->s2b = h2.sb;
        p->m1 = h1.m; p->s1a = h1.sa; p->s1b = h1.sb;
        return PB3D_OK;
    }
    if (ncolors) memcpy(p->colors, colors, (size_t)3 * ncolors);
    for (int k = 0; k < 32; ++k)
        p->colors32[k] = k < ncolors ? ((u32)colors[3 * k] | ((u32)colors[3 * k + 1] << 8) | ((u32)colors[3 * k + 2] << 16)) : 0xffffffffu;
    // perfect hash of the colour set: first odd 24-bit multiplier of a fixed sequence under which all (distinct)
    // colours fall into different entries of the 256-entry table
    p->hashK = 0;
    for (int e = 0; e < 256; ++e) p->htab[e] = 1u;
    if (ncolors > 0) {
        u32 K = 0x9e3779u;
        for (int attempt = 0; attempt < (1 << 16) && !p->hashK; ++attempt) {
            K = (K * 1664525u + 1013904223u) & 0xffffffu;
            const u32 Ko = K | 1u;
            u32 tab[256];
            for (int e = 0; e < 256; ++e) tab[e] = 1u;
            bool ok = true;
            for (int k = 0; k < ncolors && ok; ++k) {
                const u32 c = p->colors32[k];
                const u32 e = (u32)(((unsigned long long)c * Ko) & 0xffffffffull) >> 24;
                if (tab[e] == 1u) tab[e] = c << 8;
                else if (tab[e] != (c << 8)) ok = false;        // a repeated colour is not a collision
            }
            if (ok) { p->hashK = Ko; memcpy(p->htab, tab, sizeof(tab)); }
        }
    }
    const Magic g2 = make_magic((u32)(A2 > 0 ? A2 : 1)), g1 = make_magic((u32)(A1 > 0 ? A1 : 1));
    p->m2 = g2.m; p->s2a = g2.sa; p->s2b = g2.sb;
    p->m1 = g1.m; p->s1a = g1.sa; p->s1b = g1.sb;
    return PB3D_OK;
}

}  // namespace

extern "C" {

int pb3d_points_count_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C,
                          const uint8_t* colors, int ncolors, int stride, int64_t* n) {
    PB3D_REQUIRE(ctx != nullptr && n != nullptr, "pb3d_points_count: null argument");
    SelParams p;
    PB3D_TRY(make_params(A0, A1, A2, C, colors, ncolors, stride, &p));
    *n = 0;
    if (p.nlat == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid != nullptr, "pb3d_points_count: null grid");
    const i64 nb = (p.nlat + kBlockVox - 1) / kBlockVox;
    PB3D_REQUIRE(nb < (1ll << 31), "pb3d_points_count: grid too large");
    void *counts, *offsets;
    PB3D_TRY(pb3d_scratch(ctx, 8, (size_t)nb * sizeof(u32), &counts));
    PB3D_TRY(pb3d_scratch(ctx, 9, (size_t)(nb + 1) * sizeof(i64), &offsets));
    const bool fast16 = stride == 1 && (((uintptr_t)d_grid) & 15u) == 0;
    void* masks = nullptr;
    if (fast16) PB3D_TRY(pb3d_scratch(ctx, 25, (size_t)nb * 256 * sizeof(unsigned short), &masks));
    if (fast16 && C == 3) hipLaunchKernelGGL(k_points_count16<3>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (u32*)counts, (unsigned short*)masks);
    else if (fast16) hipLaunchKernelGGL(k_points_count16<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (u32*)counts, (unsigned short*)masks);
    else hipLaunchKernelGGL(k_points_count, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (u32*)counts);
    PB3D_CHECK_LAUNCH();
    i64* total = (i64*)offsets + nb;
    {
        const i64 nseg = (nb + kSeg - 1) / kSeg;
        void *local, *segs;
        PB3D_TRY(pb3d_scratch(ctx, 11, (size_t)nb * sizeof(u32), &local));
        PB3D_TRY(pb3d_scratch(ctx, 15, 64 + (size_t)nseg * (sizeof(u32) + sizeof(i64)), &segs));
        i64* seg_base = (i64*)((u8*)segs + 64);                    // [0,64): the rotate kernels' flag word lives in this slot
        u32* seg_total = (u32*)(seg_base + nseg);
        hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, (const u32*)counts, nb, (u32*)local, seg_total);
        PB3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)seg_total, nseg, seg_base, total);
        PB3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_scan_add, dim3(pb3d_stream_blocks(ctx, nb, 256, 8)), dim3(256), 0, ctx->stream, (const u32*)local,
                           (const i64*)seg_base, nb, (i64*)offsets);
        PB3D_CHECK_LAUNCH();
    }
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, total, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    *n = *(i64*)ctx->pinned;
    return PB3D_OK;
}

// Must follow pb3d_points_count_dev with identical arguments on the same context (the scanned block
// offsets stay in the context's scratch).  d_pts: n*3 floats, d_cols: n*C bytes.
int pb3d_points_fill_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C,
                         const uint8_t* colors, int ncolors, int stride, int64_t n, float* d_pts, uint8_t* d_cols) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_points_fill: null context");
    SelParams p;
    PB3D_TRY(make_params(A0, A1, A2, C, colors, ncolors, stride, &p));
    if (p.nlat == 0 || n == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid && d_pts && d_cols, "pb3d_points_fill: null buffer");
    const i64 nb = (p.nlat + kBlockVox - 1) / kBlockVox;
    PB3D_REQUIRE(ctx->scratch[9] && ctx->scratch_bytes[9] >= (size_t)(nb + 1) * sizeof(i64),
                 "pb3d_points_fill: call pb3d_points_count first");
    const bool fast16 = stride == 1 && (((uintptr_t)d_grid) & 15u) == 0;
    PB3D_REQUIRE(!fast16 || (ctx->scratch[25] && ctx->scratch_bytes[25] >= (size_t)nb * 256 * sizeof(unsigned short)),
                 "pb3d_points_fill: call pb3d_points_count first");
    // the wave-private form (k_points_fillw) from round 4 on; knob points_fill = 1: the block form of rounds 2 / 3 (development A/B)
    const bool wavefill = fast16 && p.nlat <= 0xffffffffll && ctx->tune_points_fill != 1;
    if (wavefill && C == 1)
        hipLaunchKernelGGL((k_points_fillw<1>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)ctx->scratch[9], d_pts, d_cols,
                           (const unsigned short*)ctx->scratch[25]);
    else if (wavefill)
        hipLaunchKernelGGL((k_points_fillw<3>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)ctx->scratch[9], d_pts, d_cols,
                           (const unsigned short*)ctx->scratch[25]);
    else if (fast16 && C == 1)
        hipLaunchKernelGGL((k_points_fill16<1, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)ctx->scratch[9], d_pts,
                           d_cols, ScanState{}, (const unsigned short*)ctx->scratch[25]);
    else if (fast16)
        hipLaunchKernelGGL((k_points_fill16<3, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)ctx->scratch[9], d_pts,
                           d_cols, ScanState{}, (const unsigned short*)ctx->scratch[25]);
    else
        hipLaunchKernelGGL(k_points_fill, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)ctx->scratch[9], d_pts,
                           d_cols);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}


int pb3d_points_extract_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C, const uint8_t* colors, int ncolors,
                            int64_t capacity, float* d_pts, uint8_t* d_cols, int64_t* n) {
    PB3D_REQUIRE(ctx != nullptr && n != nullptr && capacity >= 0, "pb3d_points_extract: bad argument");
    *n = 0;
    SelParams p;
    PB3D_TRY(make_params(A0, A1, A2, C, colors, ncolors, 1, &p));
    if (p.nlat == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid && (capacity == 0 || (d_pts && d_cols)), "pb3d_points_extract: null buffer");
    PB3D_REQUIRE((((uintptr_t)d_grid) & 15u) == 0, "pb3d_points_extract: needs a 16-byte aligned grid");
    const i64 nb = (p.nlat + kBlockVox - 1) / kBlockVox;
    PB3D_REQUIRE(nb < (1ll << 31), "pb3d_points_extract: grid too large");
    void* stv;
    PB3D_TRY(pb3d_scratch(ctx, 24, (size_t)nb * sizeof(unsigned long long) + 64, &stv));   // not slot 9: a pending count -> fill pair keeps its offsets there
    PB3D_HIP(hipMemsetAsync(stv, 0, (size_t)nb * sizeof(unsigned long long) + 64, ctx->stream));
    ScanState st;
    st.status = (unsigned long long*)((u8*)stv + 64);
    st.total = (i64*)((u8*)stv + 8);
    st.capacity = capacity;
    if (C == 1) hipLaunchKernelGGL((k_points_fill16<1, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)nullptr, d_pts, d_cols, st, (const unsigned short*)nullptr);
    else hipLaunchKernelGGL((k_points_fill16<3, true>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_grid, p, (const i64*)nullptr, d_pts, d_cols, st, (const unsigned short*)nullptr);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, st.total, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    *n = *(i64*)ctx->pinned;
    if (*n < 0) {          // the look-back was abandoned (a dispatch order this code does not expect): nothing hung, nothing was written
        *n = 0;
        pb3d_set_error("pb3d_points_extract: look-back abandoned; use pb3d_points_count / pb3d_points_fill");
        return PB3D_EUNSUPPORTED;
    }
    return PB3D_OK;
}

}  // extern "C"
