// N1/N2: per-component statistics, the crop / paste steps of left_right_guided_carve, surface extrusion and component
// recolouring.  The connected-component labelling itself (pb3d_label_color_dev) is csrc/ccl.hip.
#include "pb3d_internal.h"

namespace {

// per component: bbox (lo inclusive, hi inclusive), voxel count, coordinate sums.
// Three levels of aggregation keep the global atomics off the critical path even when one component owns
// millions of voxels: (1) a run of equal labels along the fastest axis is ONE closed-form contribution, (2) the runs'
// first lanes accumulate into a small per-block table in LDS (16 labels, open addressing), (3) the table is flushed with
// one set of global atomics per label per block.  A label that does not fit the table goes straight to global memory.
constexpr int kStatSlots = 16;

// One wavefront per row, one lane per voxel of a 64-voxel window: a row's voxels share a0 and a1, and a RUN of equal labels is an
// interval of a2, so its contribution is closed-form (count = length, coordinate sums from the end points) and only the FIRST
// lane of each run works -- the first version reduced every voxel through six rounds of nine shuffles (1.6 ms at 1024^3, VALU bound).
__global__ __launch_bounds__(256) void k_comp_stats(const int* __restrict__ labels, i64 rows, pb3d_magic m1, int A2, int* __restrict__ bbox,
                                                    unsigned long long* __restrict__ cnt_sum) {
    __shared__ int slab[kStatSlots];
    __shared__ int slo[kStatSlots][3], shi[kStatSlots][3];
    __shared__ unsigned long long scs[kStatSlots][4];
    if (threadIdx.x < kStatSlots) {
        slab[threadIdx.x] = 0;
        for (int a = 0; a < 3; ++a) { slo[threadIdx.x][a] = 0x7fffffff; shi[threadIdx.x][a] = -1; }
        for (int a = 0; a < 4; ++a) scs[threadIdx.x][a] = 0ull;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int nwin = (A2 + 63) >> 6;
    for (i64 rw = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); rw < rows; rw += (i64)gridDim.x * 4) {
        const u32 row = (u32)__builtin_amdgcn_readfirstlane((int)rw);
        const u32 a0 = pb3d_div(row, m1), a1 = row - a0 * m1.d;
        const int* lrow = labels + (i64)row * A2;
        for (int t0 = 0; t0 < nwin; t0 += 16) {
            int Lq[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {                                      // sixteen windows' loads in flight before any is used
                const int a2 = 64 * (t0 + q) + lane;
                Lq[q] = a2 < A2 ? lrow[a2] : 0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int a2 = 64 * (t0 + q) + lane;
                const int L = Lq[q];
                const int before = __shfl_up(L, 1);
                const bool member = L > 0, start = member && (lane == 0 || before != L);
                const u64 mb = __ballot(member);
                if (!mb) continue;
                const u64 sb = __ballot(start);
                if (!start) continue;
                const u64 stop = (sb | ~mb) & ~le;                                  // the next run start or non-member above this lane
                const int len = (stop ? __ffsll((unsigned long long)stop) - 1 : 64) - lane;
                const int lo[3] = {(int)a0, (int)a1, a2}, hi[3] = {(int)a0, (int)a1, a2 + len - 1};
                const unsigned long long cnt = (unsigned long long)len;
                const unsigned long long sm[3] = {(unsigned long long)a0 * cnt, (unsigned long long)a1 * cnt,
                                                  (unsigned long long)(2 * a2 + len - 1) * cnt / 2ull};
                int slot = L & (kStatSlots - 1), found = -1;
                for (int k = 0; k < kStatSlots; ++k) {
                    const int old = atomicCAS(&slab[slot], 0, L);
                    if (old == 0 || old == L) { found = slot; break; }
                    slot = (slot + 1) & (kStatSlots - 1);
                }
                if (found >= 0) {
                    for (int a = 0; a < 3; ++a) { atomicMin(&slo[found][a], lo[a]); atomicMax(&shi[found][a], hi[a]); }
                    atomicAdd(&scs[found][0], cnt);
                    for (int a = 0; a < 3; ++a) atomicAdd(&scs[found][1 + a], sm[a]);
                } else {
                    int* bb = bbox + 6 * (L - 1);
                    for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], lo[a]); atomicMax(&bb[3 + a], hi[a]); }
                    unsigned long long* cs = cnt_sum + 4 * (L - 1);
                    atomicAdd(&cs[0], cnt);
                    for (int a = 0; a < 3; ++a) atomicAdd(&cs[1 + a], sm[a]);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kStatSlots && slab[threadIdx.x] > 0) {
        const int Lc = slab[threadIdx.x];
        int* bb = bbox + 6 * (Lc - 1);
        for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], slo[threadIdx.x][a]); atomicMax(&bb[3 + a], shi[threadIdx.x][a]); }
        unsigned long long* cs = cnt_sum + 4 * (Lc - 1);
        for (int a = 0; a < 4; ++a) atomicAdd(&cs[a], scs[threadIdx.x][a]);
    }
}

// occupancy of the bbox crop of a colour grid: occ[xs,ys,zs] = any(grid[x0+xs, y0+ys, z0+zs, :] > 0)
// (C = 1: the grid is a 1-byte label volume)
__global__ __launch_bounds__(256) void k_crop_occ(const u8* __restrict__ grid, i64 A1, i64 A2, i64 x0, i64 y0, i64 z0, i64 Wc, i64 Hc,
                                                  i64 Dc, u8* __restrict__ occ, int C) {
    const i64 n = Wc * Hc * Dc;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 zs = i % Dc, r = i / Dc, ys = r % Hc, xs = r / Hc;
        const u8* p = grid + (((x0 + xs) * A1 + (y0 + ys)) * A2 + (z0 + zs)) * C;
        occ[i] = (C == 1 ? p[0] : (p[0] | p[1] | p[2])) ? 1 : 0;
    }
}

// left_right_guided_carve paste (reference :197-201) inside the bbox of component `id`:
//   carved[v] = 0 where labels[v] == id ; then carved[v] = colored[v] where carved_occ && colored[v] != 0
__global__ __launch_bounds__(256) void k_comp_paste(const u8* __restrict__ colored, const int* __restrict__ labels, int id,
                                                    const u8* __restrict__ carved_occ, i64 A1, i64 A2, i64 x0, i64 y0, i64 z0, i64 Wc,
                                                    i64 Hc, i64 Dc, u8* __restrict__ carved, int C) {
    const i64 n = Wc * Hc * Dc;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 zs = i % Dc, r = i / Dc, ys = r % Hc, xs = r / Hc;
        const i64 v = ((x0 + xs) * A1 + (y0 + ys)) * A2 + (z0 + zs);
        if (C == 1) {       // label volume: one byte per voxel
            const u8 c = colored[v];
            if (carved_occ[i] && c) carved[v] = (u8)(c * carved_occ[i]);
            else if (labels[v] == id) carved[v] = 0;
            continue;
        }
        const u8 c0 = colored[3 * v], c1 = colored[3 * v + 1], c2 = colored[3 * v + 2];
        // subgrid * carved_occ in uint8 (carved_occ is 0/1 here): non-zero iff both are
        if (carved_occ[i] && (c0 | c1 | c2)) {
            carved[3 * v] = (u8)(c0 * carved_occ[i]); carved[3 * v + 1] = (u8)(c1 * carved_occ[i]); carved[3 * v + 2] = (u8)(c2 * carved_occ[i]);
        } else if (labels[v] == id) {
            carved[3 * v] = 0; carved[3 * v + 1] = 0; carved[3 * v + 2] = 0;
        }
    }
}

// recolor_backward_components (reference :263-265): voxels whose component is flagged get new_color
// (C = 1: the grid is a 1-byte label volume, r the new label)
__global__ __launch_bounds__(256) void k_recolor_flagged(const int* __restrict__ labels, const u8* __restrict__ comp_flag, i64 n, u8 r,
                                                         u8 g, u8 b, u8* __restrict__ grid, int C) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (i64)gridDim.x * blockDim.x) {
        const int L = labels[v];
        if (L > 0 && comp_flag[L - 1]) {
            if (C == 1) grid[v] = r;
            else { grid[3 * v] = r; grid[3 * v + 1] = g; grid[3 * v + 2] = b; }
        }
    }
}

// the same from the membership bits of the labelling (ctx->ccl_last): a wavefront takes 64 windows at a time (one lane each), then all
// its lanes work on each non-empty one, a lane per voxel (coalesced label reads) -- only windows that hold members read any label:
// 9 MB of bits instead of 292 MB of labels at Taj 512
__device__ __forceinline__ u64 rl64(u64 v, int l) {
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)v, l);
}
__global__ __launch_bounds__(256) void k_recolor_bits(const u64* __restrict__ bits, const int* __restrict__ labels, const u8* __restrict__ comp_flag,
                                                      i64 nwords, pb3d_magic mP, int A2, u8 r, u8 g, u8 b, u8* __restrict__ grid, int C, int nflag) {
    const int lane = threadIdx.x & 63;
    const i64 nwaves = (i64)gridDim.x * 4, wid = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (i64 c0 = wid * 64; c0 < nwords; c0 += nwaves * 64) {
        const i64 idx = c0 + lane;
        const u64 mine = idx < nwords ? bits[idx] : 0ull;
        u64 todo = __ballot(mine != 0ull);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const u64 w = rl64(mine, src);
            if (!((w >> lane) & 1ull)) continue;
            const u32 widx = (u32)(c0 + src);
            const u32 row = pb3d_div(widx, mP), t = widx - row * mP.d;
            const i64 v = (i64)row * A2 + 64 * (i64)t + lane;
            const int L = labels[v];
            if (L > 0 && L <= nflag && comp_flag[L - 1]) {        // (labels beyond the flag array: not decided here, not painted)
                if (C == 1) grid[v] = r;
                else { grid[3 * v] = r; grid[3 * v + 1] = g; grid[3 * v + 2] = b; }
            }
        }
    }
}

// recolor_backward_components without the host (reference :256-265): the k components with the smallest mean coordinate on sort_axis
// are kept -- sorted() is stable, so equal means keep their numbering order -- the others flagged.  One workgroup: component i's rank is
// the number of components that sort before it.  means = float64 sum / float64 count, as np.mean of the int64 column gives them.
// status[0] = number of components, status[1] = 1 when there are more than the records hold (nothing is flagged then: the caller's
// host path decides).
constexpr int kRecolorDeviceMax = 2048;
__global__ __launch_bounds__(1024) void k_recolor_select(const i64* __restrict__ total, const char* __restrict__ records, int dcap, int keep_k, int axis,
                                                         u8* __restrict__ flags, i64* __restrict__ status) {
    __shared__ double mean[kRecolorDeviceMax];
    const i64 n = total[0];
    const bool over = n > dcap || n > kRecolorDeviceMax;
    if (threadIdx.x == 0 && status) { status[0] = n; status[1] = over ? 1 : 0; }
    const int m = over ? 0 : (int)n;
    for (int i = (int)threadIdx.x; i < dcap; i += (int)blockDim.x) flags[i] = 0;
    for (int i = (int)threadIdx.x; i < m; i += (int)blockDim.x) {
        const unsigned long long* cs = (const unsigned long long*)(records + (i64)i * 64 + 32);
        mean[i] = (double)(long long)cs[1 + axis] / (double)(long long)cs[0];
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < m; i += (int)blockDim.x) {
        const double mi = mean[i];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (mean[j] < mi) || (mean[j] == mi && j < i);
        flags[i] = rank >= keep_k ? 1 : 0;
    }
}

// extrude_from_surface, axis 2 (reference :218-228): one wavefront per (x,y) column.  start = index of the
// first occupied voxel from the chosen side (0 / D-1 for an empty column, like np.argmax), then `depth`
// cells from there, inside the grid, are painted where valid[x,y].
// (src and dst may be the same volume: a column is scanned and painted by one wavefront / one thread)
__global__ __launch_bounds__(256) void k_extrude_z(const u8* src, u8* dst, const u8* __restrict__ valid_wh,
                                                   i64 W, i64 H, i64 D, int plus, int depth, int has_color, u8 cr, u8 cg, u8 cb, int C) {
    const int lane = threadIdx.x & 63;
    const i64 col = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= W * H) return;
    if (!valid_wh[col]) return;     // wave-uniform: nothing is painted in this column, so its surface is not looked for either
    const u8* s = src + col * D * C;
    i64 start = plus ? 0 : D - 1;   // argmax of an all-zero column is index 0 (of the possibly reversed view)
    for (i64 base = 0; base < D; base += 64) {
        const i64 j = base + lane;                     // position along the scan direction
        const i64 z = plus ? j : D - 1 - j;
        const bool on = j < D && (C == 1 ? s[z] : (s[3 * z] | s[3 * z + 1] | s[3 * z + 2]));
        const u64 bal = __ballot(on);
        if (bal) {
            const i64 jj = base + (__ffsll((unsigned long long)bal) - 1);
            start = plus ? jj : D - 1 - jj;
            break;
        }
    }
    for (int d = lane; d < depth; d += 64) {
        const i64 z = plus ? start + d : start - d;
        if (z < 0 || z >= D) continue;
        u8* o = dst + (col * D + z) * C;
        o[0] = has_color ? cr : (u8)0;
        if (C == 3) { o[1] = has_color ? cg : (u8)0; o[2] = has_color ? cb : (u8)0; }
    }
}

// extrude_from_surface, axis 0 (reference :230-240): columns run along x for every (y,z); valid is indexed
// [y, z] exactly as upstream indexes its (H,W) mask with the z coordinate (which needs D == W).
// Round 4: one workgroup (8 waves) per 64 consecutive (y,z) columns, LANES ALONG z -- the voxels a wave reads at one x are neighbours in
// memory (192 bytes for 64 lanes) -- and the x axis scanned in rounds of doubling length, every round split over the waves; a wave lowers
// the column's first-hit index in LDS.  (Rounds 2-3
// scanned a column with the 64 lanes of a wave along x, a plane apart each: every 3-byte load its own 64-byte sector, 20x the traffic,
// 59 us per call at Taj 512; a lane per column walking x alone is a chain of W / 8 dependent round trips.)
// (src and dst may be the same volume: a column is scanned by its workgroup before that workgroup paints it, columns are independent)
typedef u32 u32_una __attribute__((aligned(1)));
constexpr int kExtrudeWaves = 8;
__global__ __launch_bounds__(64 * kExtrudeWaves) void k_extrude_x(const u8* src, u8* dst, const u8* __restrict__ valid_hw,
                                                   i64 W, i64 H, i64 D, i64 Wmask, int plus, int depth, int has_color, u8 cr, u8 cg,
                                                   u8 cb, int C) {
    __shared__ int best[64];                                   // per column: the smallest scan index with a voxel found so far
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const i64 n = H * D;
    const i64 mine = (i64)blockIdx.x * 64 + lane;
    bool v = false;
    i64 y = 0, z = 0;
    if (mine < n) { y = mine / D; z = mine - y * D; v = valid_hw[y * Wmask + z] != 0; }
    if (!__ballot(v)) return;                                  // (every wave of the workgroup sees the same 64 columns: uniform)
    if (wv == 0) best[lane] = 0x7fffffff;
    __syncthreads();
    const i64 last_vox = W * H * D - 1;
    // rounds of 16, 32, 64, ... planes of the scan order, each split evenly over the waves: a surface right behind the boundary (a dense
    // mask over a full grid) costs one round of two planes per wave, a column that holds nothing log2(W / 16) rounds -- at most twice the
    // planes a lone walker would read, with up to 16 of them in flight per lane
    for (i64 R = 0, len = 16; R < W; R += len, len *= 2) {
        const i64 per = (len + kExtrudeWaves - 1) / kExtrudeWaves;
        const i64 rend = R + len < W ? R + len : W;
        const i64 a = R + wv * per, b = a + per < rend ? a + per : rend;
        if (v && __hip_atomic_load(&best[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (int)a) {
            for (i64 jb = a; jb < b; jb += 16) {
                u32 on[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    on[k] = 0u;
                    if (jb + k >= b) continue;                                                // (wave-uniform: a round's share may be 2, 4, 8 planes)
                    const i64 j = jb + k;
                    const i64 x = plus ? j : W - 1 - j;
                    const i64 vox = (x * H + y) * D + z;
                    const u8* p = src + vox * C;
                    if (C == 1) on[k] = p[0];
                    else if (vox != last_vox) on[k] = *(const u32_una*)p & 0x00ffffffu;       // the voxel's three bytes in one load (the 4th byte exists)
                    else on[k] = (u32)p[0] | (u32)p[1] | (u32)p[2];
                }
                int h = -1;
#pragma unroll
                for (int k = 15; k >= 0; --k)
                    if (on[k]) h = k;
                if (h >= 0) { atomicMin(&best[lane], (int)(jb + h)); break; }
            }
        }
        if (!__syncthreads_or(v && best[lane] == 0x7fffffff)) break;       // every column under the mask has its surface
    }
    if (!v) return;
    const int f = best[lane];
    const i64 start = f == 0x7fffffff ? (plus ? 0 : W - 1) : (plus ? (i64)f : W - 1 - (i64)f);   // argmax of an all-zero column is index 0 (of the possibly reversed view)
    for (int d = wv; d < depth; d += kExtrudeWaves) {
        const i64 x = plus ? start + d : start - d;
        if (x < 0 || x >= W) continue;
        u8* o = dst + ((x * H + y) * D + z) * C;
        o[0] = has_color ? cr : (u8)0;
        if (C == 3) { o[1] = has_color ? cg : (u8)0; o[2] = has_color ? cb : (u8)0; }
    }
}

// notebook-1 output orientation (reference :384-385): oriented = flip(grid.transpose(2,1,0,3), axis=1), i.e.
// out[z, b, x, :] = grid[x, H-1-b, z, :] as a C-contiguous (D,H,W,3) array.  32x32 (x,z) tiles through LDS.
__global__ __launch_bounds__(256) void k_orient(const u8* __restrict__ grid, u8* __restrict__ out, i64 W, i64 H, i64 D) {
    __shared__ u8 t[32][32 * 3 + 4];
    const i64 x0 = (i64)blockIdx.x * 32, z0 = (i64)blockIdx.y * 32, y = blockIdx.z;
    for (int i = threadIdx.x; i < 32 * 96; i += 256) {         // rows of the source tile: fixed x, 32 z * 3 bytes contiguous
        const int xl = i / 96, b = i - xl * 96;
        const i64 x = x0 + xl, zb = z0 * 3 + b;
        t[xl][b] = (x < W && zb < D * 3) ? grid[((x * H + y) * D) * 3 + zb] : (u8)0;
    }
    __syncthreads();
    const i64 yb = H - 1 - y;
    for (int i = threadIdx.x; i < 32 * 96; i += 256) {         // rows of the destination tile: fixed z, 32 x * 3 bytes contiguous
        const int zl = i / 96, b = i - zl * 96;
        const int xl = b / 3, c = b - 3 * xl;
        const i64 z = z0 + zl, x = x0 + xl;
        if (z < D && x < W) out[((z * H + yb) * W + x) * 3 + c] = t[xl][zl * 3 + c];
    }
}

// the same permutation with dword accesses on both sides (W % 4 == 0 and D % 4 == 0, 4-byte aligned buffers): a 32 x 32 tile's
// source and destination rows are 96 bytes = 24 aligned dwords; a destination dword gathers its four bytes from LDS.  The byte
// version spends 12 byte loads and 12 byte stores per thread and tile (0.39 ms at 512 x 278 x 512).
__global__ __launch_bounds__(256) void k_orient4(const u8* __restrict__ grid, u8* __restrict__ out, i64 W, i64 H, i64 D) {
    __shared__ __attribute__((aligned(4))) u8 t[32][32 * 3 + 4];
    const i64 x0 = (i64)blockIdx.x * 32, z0 = (i64)blockIdx.y * 32, y = blockIdx.z;
    for (int i = threadIdx.x; i < 32 * 24; i += 256) {          // source rows: fixed x, 32 z * 3 bytes = 24 dwords
        const int xl = i / 24, k = i - xl * 24;
        const i64 x = x0 + xl, zb = z0 * 3 + 4 * k;
        u32 v = 0;
        if (x < W && zb < D * 3) v = *(const u32*)(grid + ((x * H + y) * D) * 3 + zb);       // D % 4 == 0: a dword is inside the row or past it
        *(u32*)&t[xl][4 * k] = v;
    }
    __syncthreads();
    const i64 yb = H - 1 - y;
    for (int i = threadIdx.x; i < 32 * 24; i += 256) {          // destination rows: fixed z, 32 x * 3 bytes = 24 dwords
        const int zl = i / 24, k = i - zl * 24;
        const i64 z = z0 + zl, xb = x0 * 3 + 4 * k;
        if (z >= D || xb >= W * 3) continue;
        u32 v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = 4 * k + q, xl = b / 3, c = b - 3 * xl;
            v |= (u32)t[xl][zl * 3 + c] << (8 * q);
        }
        *(u32*)(out + ((z * H + yb) * W) * 3 + xb) = v;
    }
}

__global__ __launch_bounds__(256) void k_stats_init(i64 ncomp, int* __restrict__ bbox, unsigned long long* __restrict__ cnt_sum) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < ncomp; k += (i64)gridDim.x * blockDim.x) {
        for (int a = 0; a < 3; ++a) { bbox[6 * k + a] = 0x7fffffff; bbox[6 * k + 3 + a] = -1; }
        for (int a = 0; a < 4; ++a) cnt_sum[4 * k + a] = 0ull;
    }
}

// ... of a 1-byte label volume (row N3): out[z, H-1-y, x] = grid[x, y, z]; 64 x 64 byte tiles, dword rows on both sides where aligned
__global__ __launch_bounds__(256) void k_orient_label(const u8* __restrict__ grid, u8* __restrict__ out, i64 W, i64 H, i64 D) {
    __shared__ u8 t[64][64 + 4];
    const i64 x0 = (i64)blockIdx.x * 64, z0 = (i64)blockIdx.y * 64, y = blockIdx.z;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int xl = i >> 6, zl = i & 63;
        const i64 x = x0 + xl, z = z0 + zl;
        t[xl][zl] = (x < W && z < D) ? grid[(x * H + y) * D + z] : (u8)0;
    }
    __syncthreads();
    const i64 yb = H - 1 - y;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int zl = i >> 6, xl = i & 63;
        const i64 x = x0 + xl, z = z0 + zl;
        if (x < W && z < D) out[(z * H + yb) * W + x] = t[xl][zl];
    }
}

// The same permutation for grids whose x and z extents are multiples of 128 (Taj 512 x 278 x 512, 1024^3): 128 x 128-pixel tiles, so a
// source row (fixed x, 128 z) and a destination row (fixed z, 128 x) are both 384 bytes = THREE WHOLE 128-byte lines, moved as 16-byte
// vectors; the 3-byte pixels are transposed by byte gathers from LDS (row pitch 400 bytes).  k_orient4's 32-pixel tiles cut every row
// into 96-byte pieces that start mid-line: 0.30 of the HBM peak at Taj 512, 0.39 at 1024^3.
constexpr int kOT = 128, kOPitch = 404;       // 101 dwords: ODD, so the byte gathers of one destination row (pixels a row pitch apart) fall into different banks
__global__ __launch_bounds__(256) void k_orient128(const u8* __restrict__ grid, u8* __restrict__ out, i64 W, i64 H, i64 D) {
    typedef u32 u32x4o __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) u8 ot[];             // kOT rows of kOPitch bytes
    const i64 x0 = (i64)blockIdx.x * kOT, z0 = (i64)blockIdx.y * kOT, y = blockIdx.z;
    u32x4o v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {                                      // 128 source rows x 24 units of 16 bytes
        const int i = threadIdx.x + 256 * j, xl = i / 24, k = i - xl * 24;
        v[j] = __builtin_nontemporal_load((const u32x4o*)(grid + (((x0 + xl) * H + y) * D + z0) * 3 + 16 * k));
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int i = threadIdx.x + 256 * j, xl = i / 24, k = i - xl * 24;
        u32* d4 = (u32*)(ot + xl * kOPitch + 16 * k);                     // (dwords: the odd pitch leaves rows 4-byte aligned only)
        d4[0] = v[j].x; d4[1] = v[j].y; d4[2] = v[j].z; d4[3] = v[j].w;
    }
    __syncthreads();
    const i64 yb = H - 1 - y;
#pragma unroll 2
    for (int j = 0; j < 12; ++j) {                                      // 128 destination rows x 24 units
        const int i = threadIdx.x + 256 * j, zl = i / 24, k = i - zl * 24;
        u32 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32 d = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int bb = 16 * k + 4 * q + b, xl = (bb * 21846) >> 16, c = bb - 3 * xl;      // bb / 3 for bb < 384
                d |= (u32)ot[xl * kOPitch + 3 * zl + c] << (8 * b);
            }
            w[q] = d;
        }
        // (round 4 tried six unaligned LDS dword reads -- one per source pixel -- strung together with v_perm instead of the sixteen byte reads:
        // correct and 45 % SLOWER, 2.04 against 1.41 ms at 1024^3: a misaligned ds_read_b32 a row pitch apart costs more than four byte reads)
        u32x4o r; r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
        __builtin_nontemporal_store(r, (u32x4o*)(out + (((z0 + zl) * H + yb) * W + x0) * 3 + 16 * k));
    }
}

// *count += non-zero bytes of b[0..n)
__global__ __launch_bounds__(256) void k_count_nonzero(const u8* __restrict__ b, i64 n, unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) c += b[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

}  // namespace

extern "C" {

int pb3d_count_nonzero_dev(pb3d_ctx* ctx, const uint8_t* d_bytes, int64_t n, int64_t* d_count) {
    PB3D_REQUIRE(ctx != nullptr && n >= 0 && d_count != nullptr, "pb3d_count_nonzero: bad argument");
    if (n == 0) return PB3D_OK;
    PB3D_REQUIRE(d_bytes != nullptr, "pb3d_count_nonzero: null buffer");
    hipLaunchKernelGGL(k_count_nonzero, dim3(pb3d_stream_blocks(ctx, n, 256 * 16, 8)), dim3(256), 0, ctx->stream, d_bytes, n, (unsigned long long*)d_count);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_component_stats_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t A0, int64_t A1, int64_t A2, int64_t ncomp,
                             int64_t* bbox_lo_hi, int64_t* count, int64_t* coord_sum) {
    PB3D_REQUIRE(ctx && ncomp >= 0, "pb3d_component_stats: bad argument");
    if (ncomp == 0) return PB3D_OK;
    PB3D_REQUIRE(d_labels && bbox_lo_hi && count && coord_sum, "pb3d_component_stats: null buffer");
    const i64 n = A0 * A1 * A2;
    PB3D_REQUIRE(n < (1ll << 31), "pb3d_component_stats: grid too large for 32-bit labels");
    void *bb, *cs;
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)ncomp * 6 * sizeof(int), &bb));
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)ncomp * 4 * sizeof(unsigned long long), &cs));
    // lo = +inf, hi = -1, sums = 0 (set on the device: an upload of the initial boxes cost a transfer and a synchronisation per call)
    hipLaunchKernelGGL(k_stats_init, dim3(pb3d_stream_blocks(ctx, ncomp, 256, 8)), dim3(256), 0, ctx->stream, ncomp, (int*)bb, (unsigned long long*)cs);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_comp_stats, dim3(pb3d_stream_blocks(ctx, A0 * A1, 4, 8)), dim3(256), 0, ctx->stream, d_labels, A0 * A1, pb3d_make_magic((u32)A1),
                       (int)A2, (int*)bb, (unsigned long long*)cs);
    PB3D_CHECK_LAUNCH();
    int* hbb = (int*)malloc((size_t)ncomp * 6 * sizeof(int));
    unsigned long long* hcs = (unsigned long long*)malloc((size_t)ncomp * 4 * sizeof(unsigned long long));
    if (!hbb || !hcs) { free(hbb); free(hcs); pb3d_set_error("pb3d_component_stats: out of host memory"); return PB3D_ENOMEM; }
    hipError_t e = hipMemcpyAsync(hbb, bb, (size_t)ncomp * 6 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hcs, cs, (size_t)ncomp * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) { e = hipStreamSynchronize(ctx->stream); ++ctx->sync_count; }
    if (e == hipSuccess)
        for (i64 k = 0; k < ncomp; ++k) {
            for (int a = 0; a < 3; ++a) { bbox_lo_hi[6 * k + a] = hbb[6 * k + a]; bbox_lo_hi[6 * k + 3 + a] = (i64)hbb[6 * k + 3 + a] + 1; }
            count[k] = (i64)hcs[4 * k];
            for (int a = 0; a < 3; ++a) coord_sum[3 * k + a] = (i64)hcs[4 * k + 1 + a];
        }
    free(hbb); free(hcs);
    PB3D_HIP(e);
    return PB3D_OK;
}

static int crop_occupancy_impl(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3],
                               uint8_t* d_occ, int C) {
    PB3D_REQUIRE(ctx && lo && hi, "pb3d_crop_occupancy: bad argument");
    const i64 Wc = hi[0] - lo[0], Hc = hi[1] - lo[1], Dc = hi[2] - lo[2];
    PB3D_REQUIRE(lo[0] >= 0 && lo[1] >= 0 && lo[2] >= 0 && hi[0] <= A0 && hi[1] <= A1 && hi[2] <= A2 && Wc >= 0 && Hc >= 0 && Dc >= 0,
                 "pb3d_crop_occupancy: box outside the grid");
    if (Wc * Hc * Dc == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid && d_occ, "pb3d_crop_occupancy: null buffer");
    hipLaunchKernelGGL(k_crop_occ, dim3(pb3d_stream_blocks(ctx, Wc * Hc * Dc, 256, 8)), dim3(256), 0, ctx->stream, d_grid, A1, A2, lo[0],
                       lo[1], lo[2], Wc, Hc, Dc, d_occ, C);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_crop_occupancy_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3],
                            const int64_t hi[3], uint8_t* d_occ) {
    return crop_occupancy_impl(ctx, d_grid_rgb, A0, A1, A2, lo, hi, d_occ, 3);
}

int pb3d_crop_occupancy_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3],
                                  const int64_t hi[3], uint8_t* d_occ) {
    return crop_occupancy_impl(ctx, d_grid_lab, A0, A1, A2, lo, hi, d_occ, 1);
}

static int component_paste_impl(pb3d_ctx* ctx, const uint8_t* d_colored, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                                int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved, int C) {
    PB3D_REQUIRE(ctx && lo && hi, "pb3d_component_paste: bad argument");
    const i64 Wc = hi[0] - lo[0], Hc = hi[1] - lo[1], Dc = hi[2] - lo[2];
    PB3D_REQUIRE(lo[0] >= 0 && lo[1] >= 0 && lo[2] >= 0 && hi[0] <= A0 && hi[1] <= A1 && hi[2] <= A2 && Wc >= 0 && Hc >= 0 && Dc >= 0,
                 "pb3d_component_paste: box outside the grid");
    if (Wc * Hc * Dc == 0) return PB3D_OK;
    PB3D_REQUIRE(d_colored && d_labels && d_carved_occ && d_carved, "pb3d_component_paste: null buffer");
    hipLaunchKernelGGL(k_comp_paste, dim3(pb3d_stream_blocks(ctx, Wc * Hc * Dc, 256, 8)), dim3(256), 0, ctx->stream, d_colored, d_labels, id,
                       d_carved_occ, A1, A2, lo[0], lo[1], lo[2], Wc, Hc, Dc, d_carved, C);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_component_paste_dev(pb3d_ctx* ctx, const uint8_t* d_colored, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                             int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved) {
    return component_paste_impl(ctx, d_colored, d_labels, id, d_carved_occ, A0, A1, A2, lo, hi, d_carved, 3);
}

int pb3d_component_paste_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                                   int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved) {
    return component_paste_impl(ctx, d_grid_lab, d_labels, id, d_carved_occ, A0, A1, A2, lo, hi, d_carved, 1);
}

static int recolor_impl(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                        const uint8_t new_color[3], uint8_t* d_grid_rgb, int C, bool last_labelled = false) {
    PB3D_REQUIRE(ctx && new_color && nvox >= 0 && ncomp >= 0, "pb3d_recolor_components: bad argument");
    if (nvox == 0 || ncomp == 0) return PB3D_OK;
    PB3D_REQUIRE(d_labels && comp_flag && d_grid_rgb, "pb3d_recolor_components: null buffer");
    void* f;
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)ncomp, &f));
    PB3D_TRY(pb3d_h2d_async(ctx, f, comp_flag, (size_t)ncomp));      // (comp_flag is a caller-owned host buffer: staged, no wait)
    const pb3d_ctx::CclLast& cl = ctx->ccl_last;
    const bool bits_ok = cl.valid && cl.labels == (const void*)d_labels && cl.rows * cl.A2 == nvox && cl.gen == ctx->scratch_slot_gen[42] &&
                         cl.rows * cl.P < (1ll << 32) && cl.K == 1;
    PB3D_REQUIRE(!last_labelled || bits_ok, "pb3d_recolor_last_labelled: d_labels is not the volume the last pb3d_label_* call on this context wrote");
    if (last_labelled) {
        const i64 nwords = cl.rows * cl.P;
        hipLaunchKernelGGL(k_recolor_bits, dim3(pb3d_stream_blocks(ctx, (nwords + 63) / 64, 4, 8)), dim3(256), 0, ctx->stream, (const u64*)cl.bits, d_labels,
                           (const u8*)f, nwords, pb3d_make_magic((u32)cl.P), (int)cl.A2, new_color[0], new_color[1], new_color[2], d_grid_rgb, C,
                           (int)(ncomp < 0x7fffffff ? ncomp : 0x7fffffff));
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    hipLaunchKernelGGL(k_recolor_flagged, dim3(pb3d_stream_blocks(ctx, nvox, 256, 8)), dim3(256), 0, ctx->stream, d_labels, (const u8*)f, nvox,
                       new_color[0], new_color[1], new_color[2], d_grid_rgb, C);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_recolor_components_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                const uint8_t new_color[3], uint8_t* d_grid_rgb) {
    return recolor_impl(ctx, d_labels, nvox, comp_flag, ncomp, new_color, d_grid_rgb, 3);
}

int pb3d_recolor_last_labelled_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                   const uint8_t new_color[3], uint8_t* d_grid, int channels) {
    PB3D_REQUIRE(channels == 1 || channels == 3, "pb3d_recolor_last_labelled: channels is 1 (labels) or 3 (colours)");
    return recolor_impl(ctx, d_labels, nvox, comp_flag, ncomp, new_color, d_grid, channels, true);
}

// recolor_backward_components (reference utils/voxel_carving_utils.py:252-266) on a resident grid WITHOUT a host round trip: labelling
// for the members only with the statistics left on the device, the keep decision by one workgroup, the recolouring over the membership
// bits -- everything queued on the context's stream.
int pb3d_recolor_backward_dev(pb3d_ctx* ctx, uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3], const uint8_t new_color[3],
                              int keep_k, int sort_axis, int channels, int32_t* d_labels, int64_t* d_status) {
    PB3D_REQUIRE(ctx && color && new_color && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_recolor_backward: bad argument");
    PB3D_REQUIRE(channels == 1 || channels == 3, "pb3d_recolor_backward: channels is 1 (labels) or 3 (colours)");
    PB3D_REQUIRE(sort_axis >= 0 && sort_axis <= 2, "pb3d_recolor_backward: sort_axis is 0, 1 or 2");
    const i64 nvox = A0 * A1 * A2;
    if (nvox == 0) {
        if (d_status) PB3D_HIP(hipMemsetAsync(d_status, 0, 16, ctx->stream));
        return PB3D_OK;
    }
    PB3D_REQUIRE(d_grid && d_labels, "pb3d_recolor_backward: null buffer");
    pb3d_ccl_dev dev;
    PB3D_TRY(pb3d_ccl_label_on_device(ctx, d_grid, A0, A1, A2, color, channels, d_labels, kRecolorDeviceMax, &dev));
    void* f;
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)dev.dcap, &f));
    hipLaunchKernelGGL(k_recolor_select, dim3(1), dim3(1024), 0, ctx->stream, dev.total, dev.records, dev.dcap, keep_k < 0 ? 0 : keep_k, sort_axis, (u8*)f,
                       (i64*)d_status);
    PB3D_CHECK_LAUNCH();
    const pb3d_ctx::CclLast& cl = ctx->ccl_last;
    PB3D_REQUIRE(cl.valid && cl.rows * cl.P < (1ll << 32), "pb3d_recolor_backward: grid too large");
    const i64 nwords = cl.rows * cl.P;
    hipLaunchKernelGGL(k_recolor_bits, dim3(pb3d_stream_blocks(ctx, (nwords + 63) / 64, 4, 8)), dim3(256), 0, ctx->stream, (const u64*)cl.bits, d_labels,
                       (const u8*)f, nwords, pb3d_make_magic((u32)cl.P), (int)cl.A2, new_color[0], new_color[1], new_color[2], d_grid, channels, dev.dcap);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_recolor_components_label_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                      uint8_t new_label, uint8_t* d_grid_lab) {
    const uint8_t c3[3] = {new_label, 0, 0};
    return recolor_impl(ctx, d_labels, nvox, comp_flag, ncomp, c3, d_grid_lab, 1);
}

int pb3d_orient_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t W, int64_t H, int64_t D, uint8_t* d_out) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0, "pb3d_orient: bad shape");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_lab && d_out && d_grid_lab != d_out, "pb3d_orient: null or aliased buffer");
    PB3D_REQUIRE(H <= 65535 && (D + 63) / 64 <= 65535, "pb3d_orient: grid too large");
    hipLaunchKernelGGL(k_orient_label, dim3((unsigned)((W + 63) / 64), (unsigned)((D + 63) / 64), (unsigned)H), dim3(256), 0, ctx->stream, d_grid_lab, d_out, W, H, D);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_orient_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, uint8_t* d_out) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0, "pb3d_orient: bad shape");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_out && d_grid_rgb != d_out, "pb3d_orient: null or aliased buffer");
    PB3D_REQUIRE(H <= 65535 && (D + 31) / 32 <= 65535, "pb3d_orient: grid too large");
    if (W % kOT == 0 && D % kOT == 0 && ((((uintptr_t)d_grid_rgb) | ((uintptr_t)d_out)) & 15u) == 0 && ctx->tune_orient_tile != 1) {
        if (!ctx->orient_lds_set) {
            PB3D_HIP(hipFuncSetAttribute((const void*)k_orient128, hipFuncAttributeMaxDynamicSharedMemorySize, kOT * kOPitch));
            ctx->orient_lds_set = true;
        }
        hipLaunchKernelGGL(k_orient128, dim3((unsigned)(W / kOT), (unsigned)(D / kOT), (unsigned)H), dim3(256), kOT * kOPitch, ctx->stream, d_grid_rgb, d_out, W, H, D);
        PB3D_CHECK_LAUNCH();
        return PB3D_OK;
    }
    dim3 grid((unsigned)((W + 31) / 32), (unsigned)((D + 31) / 32), (unsigned)H);
    if (W % 4 == 0 && D % 4 == 0 && ((((uintptr_t)d_grid_rgb) | ((uintptr_t)d_out)) & 3u) == 0)
        hipLaunchKernelGGL(k_orient4, grid, dim3(256), 0, ctx->stream, d_grid_rgb, d_out, W, H, D);
    else
        hipLaunchKernelGGL(k_orient, grid, dim3(256), 0, ctx->stream, d_grid_rgb, d_out, W, H, D);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

static int extrude_impl(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                        int axis, int plus, int depth, const uint8_t* fill_color, uint8_t* d_out, int C) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0, "pb3d_extrude: bad shape");
    PB3D_REQUIRE(axis == 0 || axis == 2, "pb3d_extrude: axis must be 0 or 2");
    const i64 n = W * H * D;
    if (n == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_valid && d_out, "pb3d_extrude: null buffer");
    // d_out == d_grid_rgb: in place (a chain of extrusions on a resident grid then moves no volume at all)
    if (d_out != d_grid_rgb) PB3D_HIP(hipMemcpyAsync(d_out, d_grid_rgb, (size_t)n * (size_t)C, hipMemcpyDeviceToDevice, ctx->stream));
    if (depth <= 0) return PB3D_OK;
    const u8 cr = fill_color ? fill_color[0] : 0, cg = fill_color ? fill_color[1] : 0, cb = fill_color ? fill_color[2] : 0;
    if (axis == 2) {
        hipLaunchKernelGGL(k_extrude_z, dim3((unsigned)((W * H + 3) / 4)), dim3(256), 0, ctx->stream, d_grid_rgb, d_out, d_valid, W, H, D,
                           plus ? 1 : 0, depth, fill_color ? 1 : 0, cr, cg, cb, C);
    } else {
        PB3D_REQUIRE(valid_w >= D, "pb3d_extrude: axis-0 extrusion indexes the (H,W) mask with z and needs W_mask >= D");
        PB3D_REQUIRE((H * D + 63) / 64 < (1ll << 31), "pb3d_extrude: grid too large");
        hipLaunchKernelGGL(k_extrude_x, dim3((unsigned)((H * D + 63) / 64)), dim3(64 * kExtrudeWaves), 0, ctx->stream, d_grid_rgb, d_out, d_valid, W,
                           H, D, valid_w, plus ? 1 : 0, depth, fill_color ? 1 : 0, cr, cg, cb, C);
    }
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_extrude_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                     int axis, int plus, int depth, const uint8_t* fill_color, uint8_t* d_out) {
    return extrude_impl(ctx, d_grid_rgb, W, H, D, d_valid, valid_w, axis, plus, depth, fill_color, d_out, 3);
}

int pb3d_extrude_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                           int axis, int plus, int depth, int fill_label, uint8_t* d_out) {
    const uint8_t c3[3] = {(uint8_t)(fill_label < 0 ? 0 : fill_label), 0, 0};
    return extrude_impl(ctx, d_grid_lab, W, H, D, d_valid, valid_w, axis, plus, depth, fill_label < 0 ? nullptr : c3, d_out, 1);
}

}  // extern "C"
