// N1/N2: 3-D connected components (6-connectivity) of one colour, per-component statistics, the
// crop / paste steps of left_right_guided_carve, surface extrusion and component recolouring.
//
// scipy.ndimage.label(mask) with the default structure (call sites reference
// utils/voxel_carving_utils.py:175 and :254) numbers components in raster order of their first voxel.
// On the device: lock-free union-find over the voxel lattice where a root is always the SMALLEST
// linear index of its set, so "first voxel in raster order" == root, and the label of a component is
// the rank of its root among all roots (ordered compaction of the root flags, csrc/points.hip).
#include "pb3d_internal.h"

namespace {

// parent[] is read and written concurrently by every thread: all accesses are relaxed atomics so the
// compiler can neither cache nor reorder them away
__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int* parent, int v) {
    while (true) {
        const int p = ld(&parent[v]);
        if (p == v) return v;
        v = p;
    }
}

// find with path halving, used while merging only (every store is an ancestor, so the forest stays valid);
// the flatten pass uses the read-only find above so that final roots are never overwritten by a helper store
__device__ __forceinline__ int uf_find_halving(int* parent, int v) {
    while (true) {
        const int p = ld(&parent[v]);
        if (p == v) return v;
        const int gp = ld(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[v], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = p;
    }
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find_halving(parent, a);
        b = uf_find_halving(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // attach the larger root under the smaller
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                         // someone re-parented a meanwhile: retry from there
    }
}

__device__ __forceinline__ bool is_color(const u8* __restrict__ g, i64 v, u8 r, u8 gg, u8 b) {
    return g[3 * v] == r && g[3 * v + 1] == gg && g[3 * v + 2] == b;
}

// parent[v] = first voxel of v's run of members along the fastest axis, cut at row starts and at the 64-voxel segments a
// wavefront covers (ballot arithmetic, no atomics): the a2-links inside a segment are never made one by one.
__global__ __launch_bounds__(256) void k_ccl_init(const u8* __restrict__ grid, i64 n, pb3d_magic m2, u8 r, u8 g, u8 b, int* __restrict__ parent,
                                                  u8* __restrict__ member) {
    const int lane = threadIdx.x & 63;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 nloop = (n + stride - 1) / stride;               // same trip count for every lane: ballots stay convergent
    for (i64 it = 0; it < nloop; ++it) {
        const i64 v = it * stride + (i64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool m = v < n && is_color(grid, v, r, g, b);
        const u64 bal = __ballot(m);
        const bool prev = lane > 0 && ((bal >> (lane - 1)) & 1ull);
        const bool start = m && (!prev || (u32)v - pb3d_div((u32)v, m2) * m2.d == 0u);       // v % A2 == 0 (n < 2^31)
        const u64 starts = __ballot(start);
        if (v < n) {
            int par = (int)v;
            if (m) {
                const u64 upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                par = (int)(v - (lane - (63 - __clzll((long long)upto))));
            }
            parent[v] = par;
            member[v] = m ? 1 : 0;
        }
    }
}

// Links across the three axes.  A link (v, v+s) is skipped when the pair one step back along the fastest axis, (v-1, v-1+s),
// exists in the same row: it connects the same two runs (runs are already linked inside by k_ccl_init / the a2-links), so for
// blob-like components the atomics drop from one per face to one per run pair.
__global__ __launch_bounds__(256) void k_ccl_merge(const u8* __restrict__ member, i64 A0, i64 A1, i64 A2, pb3d_magic m2, pb3d_magic m1, int* parent) {
    const i64 n = A0 * A1 * A2;
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (i64)gridDim.x * blockDim.x) {
        if (!member[v]) continue;
        const u32 r = pb3d_div((u32)v, m2), a2 = (u32)v - r * m2.d, a0 = pb3d_div(r, m1), a1 = r - a0 * m1.d;     // n < 2^31
        const bool back = a2 > 0 && member[v - 1];
        if (a2 + 1 < A2 && (v & 63) == 63 && member[v + 1]) uf_union(parent, (int)v, (int)(v + 1));   // only across init's segments
        if (a1 + 1 < A1 && member[v + A2] && !(back && member[v + A2 - 1])) uf_union(parent, (int)v, (int)(v + A2));
        if (a0 + 1 < A0 && member[v + A1 * A2] && !(back && member[v + A1 * A2 - 1])) uf_union(parent, (int)v, (int)(v + A1 * A2));
    }
}

// parent[v] <- root(v) for members; rootflag[v] = 1 at roots
__global__ __launch_bounds__(256) void k_ccl_flatten(u8* __restrict__ member_to_rootflag, i64 n, const int* parent, int* __restrict__ root_out,
                                                     u8* __restrict__ rootimg) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (i64)gridDim.x * blockDim.x) {
        if (!member_to_rootflag[v]) { root_out[v] = -1; rootimg[v] = 0; continue; }
        const int root = uf_find(const_cast<int*>(parent), (int)v);
        root_out[v] = root;                                  // written to a separate array: parent[] stays intact for the other finds
        member_to_rootflag[v] = (root == (int)v) ? 1 : 2;   // 1 = root, 2 = member
        rootimg[v] = (root == (int)v) ? 1 : 0;              // the image the ordered compaction of the roots reads
    }
}

// roots come out of the ordered compaction as (a2,a1,a0) float triples; labels[root] = rank + 1
__global__ __launch_bounds__(256) void k_ccl_rank(const float* __restrict__ pts, i64 nroots, i64 A1, i64 A2, int* __restrict__ labels) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < nroots; k += (i64)gridDim.x * blockDim.x) {
        const i64 a2 = (i64)pts[3 * k], a1 = (i64)pts[3 * k + 1], a0 = (i64)pts[3 * k + 2];
        labels[(a0 * A1 + a1) * A2 + a2] = (int)(k + 1);
    }
}

__global__ __launch_bounds__(256) void k_ccl_relabel(const int* __restrict__ root, const u8* __restrict__ flag, i64 n,
                                                     int* __restrict__ labels) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (i64)gridDim.x * blockDim.x) {
        if (flag[v] == 2) labels[v] = labels[root[v]];   // the root's entry was written by k_ccl_rank and is stable
        else if (flag[v] == 0) labels[v] = 0;
    }
}

// per component: bbox (lo inclusive, hi inclusive), voxel count, coordinate sums.
// Three levels of aggregation keep the global atomics off the critical path even when one component owns
// millions of voxels: (1) lanes of a wavefront that share a label are reduced with shuffles, (2) the wave leaders
// accumulate into a small per-block table in LDS (16 labels, open addressing), (3) the table is flushed with one
// set of global atomics per label per block.  A label that does not fit the table goes straight to global memory.
constexpr int kStatSlots = 16;

__global__ __launch_bounds__(256) void k_comp_stats(const int* __restrict__ labels, i64 A0, i64 A1, i64 A2, pb3d_magic m2, pb3d_magic m1,
                                                    int* __restrict__ bbox, unsigned long long* __restrict__ cnt_sum) {
    __shared__ int slab[kStatSlots];
    __shared__ int slo[kStatSlots][3], shi[kStatSlots][3];
    __shared__ unsigned long long scs[kStatSlots][4];
    if (threadIdx.x < kStatSlots) {
        slab[threadIdx.x] = 0;
        for (int a = 0; a < 3; ++a) { slo[threadIdx.x][a] = 0x7fffffff; shi[threadIdx.x][a] = -1; }
        for (int a = 0; a < 4; ++a) scs[threadIdx.x][a] = 0ull;
    }
    __syncthreads();
    const i64 n = A0 * A1 * A2;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 nloop = (n + stride - 1) / stride;
    for (i64 it = 0; it < nloop; ++it) {
        const i64 v = it * stride + (i64)blockIdx.x * blockDim.x + threadIdx.x;
        int L = 0;
        int c[3] = {0, 0, 0};
        if (v < n) {
            L = labels[v];
            if (L > 0) { const u32 r = pb3d_div((u32)v, m2), q = pb3d_div(r, m1); c[2] = (int)((u32)v - r * m2.d); c[1] = (int)(r - q * m1.d); c[0] = (int)q; }
        }
        u64 todo = __ballot(L > 0);
        while (todo) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const int Lc = __shfl(L, leader);
            const bool mine = L == Lc;
            const u64 grp = __ballot(mine);
            int lo[3], hi[3]; long long sm[3];
            for (int a = 0; a < 3; ++a) { lo[a] = mine ? c[a] : 0x7fffffff; hi[a] = mine ? c[a] : -1; sm[a] = mine ? c[a] : 0; }
            for (int off = 32; off > 0; off >>= 1)
                for (int a = 0; a < 3; ++a) {
                    const int l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
                    const long long s2 = __shfl_xor(sm[a], off);
                    lo[a] = l2 < lo[a] ? l2 : lo[a]; hi[a] = h2 > hi[a] ? h2 : hi[a]; sm[a] += s2;
                }
            if ((threadIdx.x & 63) == leader) {
                int slot = Lc & (kStatSlots - 1), found = -1;
                for (int t = 0; t < kStatSlots; ++t) {
                    const int old = atomicCAS(&slab[slot], 0, Lc);
                    if (old == 0 || old == Lc) { found = slot; break; }
                    slot = (slot + 1) & (kStatSlots - 1);
                }
                const unsigned long long cnt = (unsigned long long)__popcll(grp);
                if (found >= 0) {
                    for (int a = 0; a < 3; ++a) { atomicMin(&slo[found][a], lo[a]); atomicMax(&shi[found][a], hi[a]); }
                    atomicAdd(&scs[found][0], cnt);
                    for (int a = 0; a < 3; ++a) atomicAdd(&scs[found][1 + a], (unsigned long long)sm[a]);
                } else {
                    int* bb = bbox + 6 * (Lc - 1);
                    for (int a = 0; a < 3; ++a) { atomicMin(&bb[a], lo[a]); atomicMax(&bb[3 + a], hi[a]); }
                    unsigned long long* cs = cnt_sum + 4 * (Lc - 1);
                    atomicAdd(&cs[0], cnt);
                    for (int a = 0; a < 3; ++a) atomicAdd(&cs[1 + a], (unsigned long long)sm[a]);
                }
            }
            todo &= ~grp;
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
__global__ __launch_bounds__(256) void k_crop_occ(const u8* __restrict__ grid, i64 A1, i64 A2, i64 x0, i64 y0, i64 z0, i64 Wc, i64 Hc,
                                                  i64 Dc, u8* __restrict__ occ) {
    const i64 n = Wc * Hc * Dc;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 zs = i % Dc, r = i / Dc, ys = r % Hc, xs = r / Hc;
        const u8* p = grid + (((x0 + xs) * A1 + (y0 + ys)) * A2 + (z0 + zs)) * 3;
        occ[i] = (p[0] | p[1] | p[2]) ? 1 : 0;
    }
}

// left_right_guided_carve paste (reference :197-201) inside the bbox of component `id`:
//   carved[v] = 0 where labels[v] == id ; then carved[v] = colored[v] where carved_occ && colored[v] != 0
__global__ __launch_bounds__(256) void k_comp_paste(const u8* __restrict__ colored, const int* __restrict__ labels, int id,
                                                    const u8* __restrict__ carved_occ, i64 A1, i64 A2, i64 x0, i64 y0, i64 z0, i64 Wc,
                                                    i64 Hc, i64 Dc, u8* __restrict__ carved) {
    const i64 n = Wc * Hc * Dc;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 zs = i % Dc, r = i / Dc, ys = r % Hc, xs = r / Hc;
        const i64 v = ((x0 + xs) * A1 + (y0 + ys)) * A2 + (z0 + zs);
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
__global__ __launch_bounds__(256) void k_recolor_flagged(const int* __restrict__ labels, const u8* __restrict__ comp_flag, i64 n, u8 r,
                                                         u8 g, u8 b, u8* __restrict__ grid) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (i64)gridDim.x * blockDim.x) {
        const int L = labels[v];
        if (L > 0 && comp_flag[L - 1]) { grid[3 * v] = r; grid[3 * v + 1] = g; grid[3 * v + 2] = b; }
    }
}

// extrude_from_surface, axis 2 (reference :218-228): one wavefront per (x,y) column.  start = index of the
// first occupied voxel from the chosen side (0 / D-1 for an empty column, like np.argmax), then `depth`
// cells from there, inside the grid, are painted where valid[x,y].
__global__ __launch_bounds__(256) void k_extrude_z(const u8* __restrict__ src, u8* __restrict__ dst, const u8* __restrict__ valid_wh,
                                                   i64 W, i64 H, i64 D, int plus, int depth, int has_color, u8 cr, u8 cg, u8 cb) {
    const int lane = threadIdx.x & 63;
    const i64 col = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= W * H) return;
    const u8* s = src + col * D * 3;
    i64 start = plus ? 0 : D - 1;   // argmax of an all-zero column is index 0 (of the possibly reversed view)
    for (i64 base = 0; base < D; base += 64) {
        const i64 j = base + lane;                     // position along the scan direction
        const i64 z = plus ? j : D - 1 - j;
        const bool on = j < D && (s[3 * z] | s[3 * z + 1] | s[3 * z + 2]);
        const u64 bal = __ballot(on);
        if (bal) {
            const i64 jj = base + (__ffsll((unsigned long long)bal) - 1);
            start = plus ? jj : D - 1 - jj;
            break;
        }
    }
    if (!valid_wh[col]) return;
    for (int d = lane; d < depth; d += 64) {
        const i64 z = plus ? start + d : start - d;
        if (z < 0 || z >= D) continue;
        u8* o = dst + (col * D + z) * 3;
        o[0] = has_color ? cr : (u8)0; o[1] = has_color ? cg : (u8)0; o[2] = has_color ? cb : (u8)0;
    }
}

// extrude_from_surface, axis 0 (reference :230-240): columns run along x for every (y,z); valid is indexed
// [y, z] exactly as upstream indexes its (H,W) mask with the z coordinate (which needs D == W).
__global__ __launch_bounds__(256) void k_extrude_x(const u8* __restrict__ src, u8* __restrict__ dst, const u8* __restrict__ valid_hw,
                                                   i64 W, i64 H, i64 D, i64 Wmask, int plus, int depth, int has_color, u8 cr, u8 cg,
                                                   u8 cb) {
    const i64 n = H * D;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 y = i / D, z = i - y * D;
        i64 start = plus ? 0 : W - 1;
        for (i64 j = 0; j < W; ++j) {
            const i64 x = plus ? j : W - 1 - j;
            const u8* p = src + ((x * H + y) * D + z) * 3;
            if (p[0] | p[1] | p[2]) { start = x; break; }
        }
        if (!valid_hw[y * Wmask + z]) continue;
        for (int d = 0; d < depth; ++d) {
            const i64 x = plus ? start + d : start - d;
            if (x < 0 || x >= W) continue;
            u8* o = dst + ((x * H + y) * D + z) * 3;
            o[0] = has_color ? cr : (u8)0; o[1] = has_color ? cg : (u8)0; o[2] = has_color ? cb : (u8)0;
        }
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

}  // namespace

extern "C" {

int pb3d_label_color_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                         int32_t* d_labels, int64_t* ncomp) {
    PB3D_REQUIRE(ctx && color && ncomp && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_label_color: bad argument");
    const i64 n = A0 * A1 * A2;
    *ncomp = 0;
    if (n == 0) return PB3D_OK;
    PB3D_REQUIRE(n < (1ll << 31), "pb3d_label_color: grid too large for 32-bit labels");
    PB3D_REQUIRE(d_grid_rgb && d_labels, "pb3d_label_color: null buffer");
    void *parent, *flag;
    PB3D_TRY(pb3d_scratch(ctx, 4, (size_t)n * sizeof(int), &parent));
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)n, &flag));
    const unsigned blocks = pb3d_stream_blocks(ctx, n, 256, 8);
    hipLaunchKernelGGL(k_ccl_init, dim3(blocks), dim3(256), 0, ctx->stream, d_grid_rgb, n, pb3d_make_magic((u32)A2), color[0], color[1], color[2], (int*)parent,
                       (u8*)flag);
    PB3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ccl_merge, dim3(blocks), dim3(256), 0, ctx->stream, (const u8*)flag, A0, A1, A2, pb3d_make_magic((u32)A2), pb3d_make_magic((u32)A1),
                       (int*)parent);
    PB3D_CHECK_LAUNCH();
    // flag values after the flatten: 0 none, 1 root, 2 member; rootimg holds the roots only (what the ordered compaction selects)
    void *roots, *rootimg;
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)n * sizeof(int), &roots));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)n, &rootimg));
    hipLaunchKernelGGL(k_ccl_flatten, dim3(blocks), dim3(256), 0, ctx->stream, (u8*)flag, n, (const int*)parent, (int*)roots, (u8*)rootimg);
    PB3D_CHECK_LAUNCH();
    i64 nroots = 0;
    PB3D_TRY(pb3d_points_count_dev(ctx, (const u8*)rootimg, A0, A1, A2, 1, nullptr, 0, 1, &nroots));
    if (nroots == 0) PB3D_HIP(hipMemsetAsync(d_labels, 0, (size_t)n * sizeof(int), ctx->stream));
    if (nroots > 0) {      // every entry of d_labels is written: roots by k_ccl_rank, members and non-members by k_ccl_relabel
        void *pts, *cols;
        PB3D_TRY(pb3d_scratch(ctx, 13, (size_t)nroots * 3 * sizeof(float), &pts));
        PB3D_TRY(pb3d_scratch(ctx, 14, (size_t)nroots, &cols));
        PB3D_TRY(pb3d_points_fill_dev(ctx, (const u8*)rootimg, A0, A1, A2, 1, nullptr, 0, 1, nroots, (float*)pts, (u8*)cols));
        hipLaunchKernelGGL(k_ccl_rank, dim3(pb3d_stream_blocks(ctx, nroots, 256, 8)), dim3(256), 0, ctx->stream, (const float*)pts, nroots,
                           A1, A2, (int*)d_labels);
        PB3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_ccl_relabel, dim3(blocks), dim3(256), 0, ctx->stream, (const int*)roots, (const u8*)flag, n, (int*)d_labels);
        PB3D_CHECK_LAUNCH();
    }
    *ncomp = nroots;
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
    // lo = +inf, hi = -1
    int* hb = (int*)malloc((size_t)ncomp * 6 * sizeof(int));
    PB3D_REQUIRE(hb != nullptr, "pb3d_component_stats: out of host memory");
    for (i64 k = 0; k < ncomp; ++k) { for (int a = 0; a < 3; ++a) { hb[6 * k + a] = 0x7fffffff; hb[6 * k + 3 + a] = -1; } }
    hipError_t e = hipMemcpyAsync(bb, hb, (size_t)ncomp * 6 * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free(hb);
    PB3D_HIP(e);
    PB3D_HIP(hipMemsetAsync(cs, 0, (size_t)ncomp * 4 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_comp_stats, dim3(pb3d_stream_blocks(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, d_labels, A0, A1, A2, pb3d_make_magic((u32)A2),
                       pb3d_make_magic((u32)A1), (int*)bb,
                       (unsigned long long*)cs);
    PB3D_CHECK_LAUNCH();
    int* hbb = (int*)malloc((size_t)ncomp * 6 * sizeof(int));
    unsigned long long* hcs = (unsigned long long*)malloc((size_t)ncomp * 4 * sizeof(unsigned long long));
    if (!hbb || !hcs) { free(hbb); free(hcs); pb3d_set_error("pb3d_component_stats: out of host memory"); return PB3D_ENOMEM; }
    e = hipMemcpyAsync(hbb, bb, (size_t)ncomp * 6 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hcs, cs, (size_t)ncomp * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
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

int pb3d_crop_occupancy_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3],
                            const int64_t hi[3], uint8_t* d_occ) {
    PB3D_REQUIRE(ctx && lo && hi, "pb3d_crop_occupancy: bad argument");
    const i64 Wc = hi[0] - lo[0], Hc = hi[1] - lo[1], Dc = hi[2] - lo[2];
    PB3D_REQUIRE(lo[0] >= 0 && lo[1] >= 0 && lo[2] >= 0 && hi[0] <= A0 && hi[1] <= A1 && hi[2] <= A2 && Wc >= 0 && Hc >= 0 && Dc >= 0,
                 "pb3d_crop_occupancy: box outside the grid");
    if (Wc * Hc * Dc == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_occ, "pb3d_crop_occupancy: null buffer");
    hipLaunchKernelGGL(k_crop_occ, dim3(pb3d_stream_blocks(ctx, Wc * Hc * Dc, 256, 8)), dim3(256), 0, ctx->stream, d_grid_rgb, A1, A2, lo[0],
                       lo[1], lo[2], Wc, Hc, Dc, d_occ);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_component_paste_dev(pb3d_ctx* ctx, const uint8_t* d_colored, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                             int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved) {
    PB3D_REQUIRE(ctx && lo && hi, "pb3d_component_paste: bad argument");
    const i64 Wc = hi[0] - lo[0], Hc = hi[1] - lo[1], Dc = hi[2] - lo[2];
    PB3D_REQUIRE(lo[0] >= 0 && lo[1] >= 0 && lo[2] >= 0 && hi[0] <= A0 && hi[1] <= A1 && hi[2] <= A2 && Wc >= 0 && Hc >= 0 && Dc >= 0,
                 "pb3d_component_paste: box outside the grid");
    if (Wc * Hc * Dc == 0) return PB3D_OK;
    PB3D_REQUIRE(d_colored && d_labels && d_carved_occ && d_carved, "pb3d_component_paste: null buffer");
    hipLaunchKernelGGL(k_comp_paste, dim3(pb3d_stream_blocks(ctx, Wc * Hc * Dc, 256, 8)), dim3(256), 0, ctx->stream, d_colored, d_labels, id,
                       d_carved_occ, A1, A2, lo[0], lo[1], lo[2], Wc, Hc, Dc, d_carved);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_recolor_components_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                const uint8_t new_color[3], uint8_t* d_grid_rgb) {
    PB3D_REQUIRE(ctx && new_color && nvox >= 0 && ncomp >= 0, "pb3d_recolor_components: bad argument");
    if (nvox == 0 || ncomp == 0) return PB3D_OK;
    PB3D_REQUIRE(d_labels && comp_flag && d_grid_rgb, "pb3d_recolor_components: null buffer");
    void* f;
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)ncomp, &f));
    PB3D_HIP(hipMemcpyAsync(f, comp_flag, (size_t)ncomp, hipMemcpyHostToDevice, ctx->stream));
    PB3D_HIP(hipStreamSynchronize(ctx->stream));   // comp_flag is a caller-owned host buffer
    hipLaunchKernelGGL(k_recolor_flagged, dim3(pb3d_stream_blocks(ctx, nvox, 256, 8)), dim3(256), 0, ctx->stream, d_labels, (const u8*)f, nvox,
                       new_color[0], new_color[1], new_color[2], d_grid_rgb);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_orient_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, uint8_t* d_out) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0, "pb3d_orient: bad shape");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_out && d_grid_rgb != d_out, "pb3d_orient: null or aliased buffer");
    PB3D_REQUIRE(H <= 65535 && (D + 31) / 32 <= 65535, "pb3d_orient: grid too large");
    dim3 grid((unsigned)((W + 31) / 32), (unsigned)((D + 31) / 32), (unsigned)H);
    hipLaunchKernelGGL(k_orient, grid, dim3(256), 0, ctx->stream, d_grid_rgb, d_out, W, H, D);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_extrude_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                     int axis, int plus, int depth, const uint8_t* fill_color, uint8_t* d_out) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0, "pb3d_extrude: bad shape");
    PB3D_REQUIRE(axis == 0 || axis == 2, "pb3d_extrude: axis must be 0 or 2");
    const i64 n = W * H * D;
    if (n == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_rgb && d_valid && d_out && d_grid_rgb != d_out, "pb3d_extrude: null or aliased buffer");
    PB3D_HIP(hipMemcpyAsync(d_out, d_grid_rgb, (size_t)n * 3, hipMemcpyDeviceToDevice, ctx->stream));
    if (depth <= 0) return PB3D_OK;
    const u8 cr = fill_color ? fill_color[0] : 0, cg = fill_color ? fill_color[1] : 0, cb = fill_color ? fill_color[2] : 0;
    if (axis == 2) {
        hipLaunchKernelGGL(k_extrude_z, dim3((unsigned)((W * H + 3) / 4)), dim3(256), 0, ctx->stream, d_grid_rgb, d_out, d_valid, W, H, D,
                           plus ? 1 : 0, depth, fill_color ? 1 : 0, cr, cg, cb);
    } else {
        PB3D_REQUIRE(valid_w >= D, "pb3d_extrude: axis-0 extrusion indexes the (H,W) mask with z and needs W_mask >= D");
        hipLaunchKernelGGL(k_extrude_x, dim3(pb3d_stream_blocks(ctx, H * D, 256, 8)), dim3(256), 0, ctx->stream, d_grid_rgb, d_out, d_valid, W,
                           H, D, valid_w, plus ? 1 : 0, depth, fill_color ? 1 : 0, cr, cg, cb);
    }
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

}  // extern "C"
