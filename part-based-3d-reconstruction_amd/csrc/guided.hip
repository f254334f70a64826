// N1: the component loop of left_right_guided_carve (reference utils/voxel_carving_utils.py:178-201) as ONE launch sequence.
//
// Per 3-D component the reference crops the bounding box, takes the crop's occupancy, runs process_voxel_grid on it with the part's
// own angle step (19 steps at the notebook's angle 5; crops are 31 x 189 x 31 ... 99 x 116 x 99 at Taj 512), clears the component and
// pastes what survives.  Launched step by step that is ~20 tiny kernels per component.  But a Y-plane never mixes with another in
// the whole chain (every rotation is about Y, the mask is per (x,y)), and 32 bit-sliced planes of a crop are one dword per (x,z)
// cell -- 40 KB for the dome: a workgroup keeps its 32 planes in LDS for ALL steps.  So, per batch of components:
//     k_crop_cells   the result table + source offset of every (component, step, cell) from SciPy's f64 arithmetic (rot_common.h)
//     k_crop_chain   one workgroup per (component, 32 planes): occupancy of the crop from the colour grid -> bit-sliced in LDS ->
//                    0-degree carve (mask bits) -> every rotation step LDS -> LDS -> count the survivors (the log's "carved
//                    voxels") and clear the component's voxels that did not survive, IN PLACE.
// In place is exact: a paste only ever writes colored[v] back where it already is, or 0 on the component's own voxels (:197-201).
// Components whose boxes overlap interact through the ORIGINAL grid (a later crop sees an earlier component's voxels): those are
// put in consecutive batches that read a copy taken before the first write.
#include <vector>

#include "rot_common.h"

namespace {

struct CropDesc {
    int x0, y0, z0, Wc, Hc, Dc, id, g0, ng, pitch;
    u32 mask_off, cell_off;
    u32 slice_off;          // dwords: where this component's ng x ncell bit-sliced occupancy words start in the batch's slice buffer
    int pad[3];
};
static_assert(sizeof(CropDesc) == 64, "CropDesc is 64 bytes");

constexpr int GTHREADS = 512;
constexpr int kMaxLds = 150 * 1024;

// celltab[cell_off + s * ncell + cell] = table << 16 | LDS offset of tap (0,0); 0 = void cell
__global__ __launch_bounds__(256) void k_crop_cells(const CropDesc* __restrict__ descs, const RotParams* __restrict__ params, int nrot,
                                                    u32* __restrict__ celltab) {
    const int comp = (int)blockIdx.y / nrot, s = (int)blockIdx.y - comp * nrot;
    const CropDesc d = descs[comp];
    const RotParams p = params[blockIdx.y];
    const int ncell = d.Wc * d.Dc;
    for (int cell = (int)blockIdx.x * 256 + (int)threadIdx.x; cell < ncell; cell += (int)gridDim.x * 256) {
        const int xs = cell / d.Dc, zs = cell - xs * d.Dc;
        const Cell c = make_cell(p, xs, zs, d.Wc, d.Dc);
        u32 w = 0;
        if (c.s0 >= 0) w = (lut_of(c) << 16) | (u32)(c.s0 * d.pitch + c.s2);
        celltab[(i64)d.cell_off + (i64)s * ncell + cell] = w;
    }
}

// Occupancy of every crop, bit-sliced (any channel > 0, reference :190): slices[slice_off + g * ncell + cell] bit q = plane 32 g + q.
// One thread per (component, plane group, cell), 32 loads in flight each, spread over the WHOLE chip: inside k_crop_chain the same
// loads were issued by the four workgroups of a dome crop -- four CUs' worth of outstanding misses, 0.29 ms of the crop's 0.5.
__global__ __launch_bounds__(256) void k_crop_slice(const u8* __restrict__ src_rgb, i64 H, i64 D, const CropDesc* __restrict__ descs, int ncomp,
                                                    u32* __restrict__ slices, int C, i64 nvol) {
    typedef u32 u32_a1 __attribute__((aligned(1)));
    int c = 0;
    while (c + 1 < ncomp && descs[c + 1].g0 <= (int)blockIdx.y) ++c;        // uniform
    const CropDesc d = descs[c];
    const int g = (int)blockIdx.y - d.g0;
    const int ncell = d.Wc * d.Dc;
    const int cell = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (cell >= ncell) return;
    const int xs = cell / d.Dc, zs = cell - xs * d.Dc;
    const int np = d.Hc - 32 * g < 32 ? d.Hc - 32 * g : 32;
    const i64 rowb = D * C;
    const u8* p = src_rgb + (((i64)(d.x0 + xs) * H + d.y0 + 32 * g) * D + d.z0 + zs) * C;
    const u8* vol_end = src_rgb + nvol * C;
    u32 any[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int q = k < np ? k : np - 1;                // (a plane past the crop re-reads the last one; its bit is dropped)
        const u8* v = p + (i64)q * rowb;
        // a voxel's three bytes as ONE unaligned dword load (the fourth byte is the next voxel's; the volume's very last voxel byte-wise)
        if (C == 1) any[k] = (u32)v[0];
        else if (v + 4 <= vol_end) any[k] = *(const u32_a1*)v & 0x00ffffffu;
        else any[k] = (u32)v[0] | (u32)v[1] | (u32)v[2];
    }
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k)
        if (k < np) bits |= (u32)(any[k] != 0) << k;
    slices[(i64)d.slice_off + (i64)g * ncell + cell] = bits;
}

// src and dst may be the same volume (no __restrict__): a workgroup reads its crop's planes before it clears anything in them, and no
// other workgroup of the launch touches those voxels (disjoint boxes within a batch, disjoint planes within a component).
// NB: batches of 8 cells a thread evaluates per step (ceil(largest crop's cells / 4096): 1, 2, 3 or 5).  The table words of step s + 1
// -- they depend on nothing the chain computes -- are loaded, ALL of them, while step s is evaluated: round 3 loaded eight words, used
// them, loaded the next eight (three dependent round trips per step on the Taj dome's 9801 cells, 18 steps: ~150 us for the crop).
template <int NB>
__global__ __launch_bounds__(GTHREADS) void k_crop_chain(const u8* src_rgb, u8* dst_rgb, const int* __restrict__ labels, i64 H, i64 D,
                                                         const CropDesc* __restrict__ descs, int ncomp, const u8* __restrict__ masks,
                                                         const u32* __restrict__ celltab, int nrot, unsigned long long* __restrict__ counts, int restore, int C, i64 nvol, int abl, const u32* __restrict__ slices,
                                                         const u64* __restrict__ mbits64, int P) {
    extern __shared__ u32 lds[];
    const int tid = threadIdx.x;
    int c = 0;
    while (c + 1 < ncomp && descs[c + 1].g0 <= (int)blockIdx.x) ++c;        // uniform
    const CropDesc d = descs[c];
    const int g = (int)blockIdx.x - d.g0;
    const int Wc = d.Wc, Dc = d.Dc, pitch = d.pitch;
    const int plane = (Wc + 1) * pitch + 1;
    u32* A = lds;
    u32* B = lds + plane;
    u32* mb = lds + 2 * plane;
    const int np = d.Hc - 32 * g < 32 ? d.Hc - 32 * g : 32;
    const int ncell = Wc * Dc;
    for (int xs = tid; xs < Wc; xs += GTHREADS) {
        const u8* mp = masks + d.mask_off + (i64)xs * d.Hc + 32 * g;
        u32 b = 0;
        for (int q = 0; q < np; ++q) b |= (u32)(mp[q] != 0) << q;
        mb[xs] = b;
    }
    __syncthreads();
    // this thread's cells: tid, tid + 512, ... -- (xs, zs) advanced without a division per cell
    const int dx = GTHREADS / Dc, dz = GTHREADS - dx * Dc;
    const int xs0 = tid / Dc, zs0 = tid - xs0 * Dc;
    // Every loop below is latency-bound if written cell by cell (a dependent global load per iteration: the first version spent
    // 220 us on a dome crop): loads are issued eight at a time before any of them is used.
    {   // the crop's bit-sliced occupancy (k_crop_slice) with the 0-degree carve (:124, first iteration)
        const u32* sl = slices + (i64)d.slice_off + (i64)g * ncell;
        int xs = xs0, zs = zs0;
        for (int base = tid; base < ncell; base += 8 * GTHREADS) {
            u32 w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int cell = base + k * GTHREADS; w[k] = (cell < ncell && !(abl & 1)) ? sl[cell] : 0u; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (base + k * GTHREADS >= ncell) break;
                A[xs * pitch + zs] = w[k] & mb[xs];
                xs += dx; zs += dz; if (zs >= Dc) { zs -= Dc; ++xs; }
            }
        }
    }
    __syncthreads();
    const int nsteps = (abl & 2) ? 0 : nrot;
    u32 cur[NB][8], nxt[NB][8];
    auto load_step = [&](int st, u32 (&w)[NB][8]) {
        const u32* ct = celltab + (i64)d.cell_off + (i64)st * ncell;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int cell = tid + (8 * bb + k) * GTHREADS; w[bb][k] = cell < ncell ? ct[cell] : 0u; }
    };
    if (nsteps > 0) load_step(0, cur);
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_step(s + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);                  // (the loads are issued HERE, not sunk to where their registers are copied)
        int xs = xs0, zs = zs0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tid + (8 * bb + k) * GTHREADS < ncell) {
                    const u32 w = cur[bb][k];
                    const u32* tp = A + (w & 0xffffu);
                    B[xs * pitch + zs] = lut_apply32(w >> 16, tp[0], tp[1], tp[pitch], tp[pitch + 1]) & mb[xs];
                    xs += dx; zs += dz; if (zs >= Dc) { zs -= Dc; ++xs; }
                }
            }
        }
        // (reading the 32 taps of eight cells before the first is evaluated -- no branch per cell -- measured no faster: 218 against 211 us on the
        // Taj dome's crop; the steps are bound by the multiplexer trees and the LDS conflicts of rotated taps, not by the round trips)
        __syncthreads();
        u32* t2 = A; A = B; B = t2;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
#pragma unroll
            for (int k = 0; k < 8; ++k) cur[bb][k] = nxt[bb][k];
    }
    // survivors: count them, and clear the component's own voxels that are not among them (reference :195-201)
    unsigned long long cnt = 0;
    {
        const u32 live = np == 32 ? 0xffffffffu : ((1u << np) - 1u);
        const u32* sl = slices + (i64)d.slice_off + (i64)g * ncell;
        int xs = xs0, zs = zs0;
        for (int cell = tid; cell < ncell; cell += GTHREADS) {
            const u32 R = A[xs * pitch + zs];
            cnt += (unsigned long long)__popc(R);
            const i64 v0 = ((i64)(d.x0 + xs) * H + d.y0 + 32 * g) * D + d.z0 + zs;
            // candidates: voxels of the crop that hold something (the slice word) and did not survive -- empty space, most of a bounding
            // box, needs neither a membership bit nor a label
            const u32 todo = (abl & 4) ? 0u : (~R & live & ((abl & 1) ? 0xffffffffu : sl[cell]));
            if (todo) {
                // all 32 planes of the cell at once: membership words, then labels, each as ONE batch of loads in flight (in groups of eight a dome
                // cell paid up to eight dependent round trips, and the four workgroups of the crop are alone on the chip: 0.11 ms of its 0.21)
                int lab[32];
                if (mbits64) {
                    // the labelling's membership bits (word (row, z / 64), bit z % 64) say which voxels carry a label at all: a label volume
                    // written for the members only (pb3d_label_*_stats_dev with members_only) holds nothing elsewhere
                    const i64 row0 = (i64)(d.x0 + xs) * H + d.y0 + 32 * g, zz = d.z0 + zs;
                    u64 mw[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) mw[k] = ((todo >> k) & 1u) ? mbits64[(row0 + k) * P + (zz >> 6)] : 0ull;
#pragma unroll
                    for (int k = 0; k < 32; ++k) lab[k] = ((mw[k] >> (zz & 63)) & 1ull) ? labels[v0 + (i64)k * D] : 0;
                } else {
#pragma unroll
                    for (int k = 0; k < 32; ++k) lab[k] = ((todo >> k) & 1u) ? labels[v0 + (i64)k * D] : 0;
                }
#pragma unroll
                for (int k = 0; k < 32; ++k)
                    if (lab[k] == d.id) { u8* o = dst_rgb + C * (v0 + (i64)k * D); o[0] = 0; if (C == 3) { o[1] = 0; o[2] = 0; } }
            }
            if (restore) {          // boxes overlap somewhere: an EARLIER component may have cleared a voxel this crop keeps -- write it back (:200-201)
                for (u32 kept = R & live; kept; kept &= kept - 1) {
                    const i64 v = v0 + (i64)__builtin_ctz(kept) * D;
                    const u8* sv = src_rgb + C * v;
                    if (C == 1) { if (sv[0]) dst_rgb[v] = sv[0]; }
                    else if (sv[0] | sv[1] | sv[2]) { u8* o = dst_rgb + 3 * v; o[0] = sv[0]; o[1] = sv[1]; o[2] = sv[2]; }
                }
            }
            xs += dx; zs += dz; if (zs >= Dc) { zs -= Dc; ++xs; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0 && cnt) atomicAdd(&counts[d.id - 1], cnt);
}

static bool boxes_overlap(const i64* a, const i64* b) {
    return a[0] < b[3] && b[0] < a[3] && a[1] < b[4] && b[1] < a[4] && a[2] < b[5] && b[2] < a[5];
}

}  // namespace

static int guided_carve_impl(pb3d_ctx* ctx, uint8_t* d_grid_rgb, const int32_t* d_labels, int64_t W, int64_t H, int64_t D, int64_t ncomp,
                             const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off, int64_t mask_bytes, int angle_interval,
                             int64_t* carved_counts, int* took, int C, int color_index, int64_t* d_counts_out = nullptr) {
    PB3D_REQUIRE(ctx && took && W >= 0 && H >= 0 && D >= 0 && ncomp >= 0, "pb3d_guided_carve: bad argument");
    *took = 0;
    PB3D_REQUIRE(angle_interval > 0, "pb3d_guided_carve: angle_interval must be a positive integer (got %d)", angle_interval);
    if (ncomp == 0 || W * H * D == 0) { *took = 1; return PB3D_OK; }
    PB3D_REQUIRE(d_grid_rgb && d_labels && bbox_lo_hi && masks && mask_off && (carved_counts || d_counts_out), "pb3d_guided_carve: null buffer");
    const bool queued = d_counts_out != nullptr;          // the counts stay on the device, the call does not wait
    const int nrot = 90 / angle_interval;                 // len(range(0, 91, k)) - 1 rotation steps after the 0-degree carve
    // every crop's 32-plane slice (two buffers + mask bits) must fit the LDS and its offsets 16 bits: otherwise the caller's loop
    size_t lds_max = 0;
    for (i64 k = 0; k < ncomp; ++k) {
        const i64* b = bbox_lo_hi + 6 * k;
        const i64 Wc = b[3] - b[0], Hc = b[4] - b[1], Dc = b[5] - b[2];
        PB3D_REQUIRE(b[0] >= 0 && b[1] >= 0 && b[2] >= 0 && b[3] <= W && b[4] <= H && b[5] <= D && Wc > 0 && Hc > 0 && Dc > 0,
                     "pb3d_guided_carve: box outside the grid");
        PB3D_REQUIRE(mask_off[k] >= 0 && mask_off[k] + Wc * Hc <= mask_bytes, "pb3d_guided_carve: mask outside the buffer");
        const i64 pitch = Dc | 1, plane = (Wc + 1) * pitch + 1;
        if (plane >= 65536) return PB3D_OK;
        const size_t lds = (size_t)(2 * plane + Wc) * sizeof(u32);
        if (lds > (size_t)kMaxLds) return PB3D_OK;
        lds_max = lds > lds_max ? lds : lds_max;
    }
    // labels of the last labelling on this context: its membership bits are still on the device
    const pb3d_ctx::CclLast& cl = ctx->ccl_last;
    const bool bits_ok = cl.valid && cl.labels == (const void*)d_labels && cl.rows == W * H && cl.A2 == D && cl.gen == ctx->scratch_slot_gen[42] &&
                         color_index >= 0 && color_index < cl.K;
    // Without the bits the kernel reads labels[] at every voxel of a crop: that is only right for a FULL label volume of ONE colour.  A volume
    // this context wrote for the members only (or for several colours, numbered per colour) must not be read that way.
    PB3D_REQUIRE(bits_ok || (color_index == 0 && !(cl.labels == (const void*)d_labels && (cl.members_only || cl.K > 1))),
                 "pb3d_guided_carve: the membership bits of this label volume are gone (it was labelled for the members only or for several colours): "
                 "label again before carving");
    const u64* mb64 = bits_ok ? (const u64*)cl.bits + (i64)color_index * cl.rows * cl.P : nullptr;
    const int mbP = bits_ok ? (int)cl.P : 0;
    if (!ctx->guided_lds_set) {
        PB3D_HIP(hipFuncSetAttribute((const void*)k_crop_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        PB3D_HIP(hipFuncSetAttribute((const void*)k_crop_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        PB3D_HIP(hipFuncSetAttribute((const void*)k_crop_chain<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        PB3D_HIP(hipFuncSetAttribute((const void*)k_crop_chain<5>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        ctx->guided_lds_set = true;
    }
    const size_t nvox3 = (size_t)(W * H * D) * (size_t)C;
    void *d_masks, *d_counts;
    PB3D_TRY(pb3d_scratch(ctx, 37, (size_t)mask_bytes + 16, &d_masks));
    if (queued) d_counts = d_counts_out;
    else PB3D_TRY(pb3d_scratch(ctx, 39, (size_t)ncomp * sizeof(unsigned long long), &d_counts));
    // every host input goes through the context's pinned staging ring: the caller's buffers are free when the call returns
    PB3D_TRY(pb3d_h2d_async(ctx, d_masks, masks, (size_t)mask_bytes));
    PB3D_HIP(hipMemsetAsync(d_counts, 0, (size_t)ncomp * sizeof(unsigned long long), ctx->stream));
    // batches: consecutive components with pairwise disjoint boxes, at most kBatch of them (descriptor upload size)
    const int kBatch = 48;
    std::vector<std::pair<i64, i64>> batches;
    for (i64 a = 0; a < ncomp;) {
        i64 e = a + 1;
        for (; e < ncomp && e - a < kBatch; ++e) {
            bool hit = false;
            for (i64 j = a; j < e && !hit; ++j) hit = boxes_overlap(bbox_lo_hi + 6 * j, bbox_lo_hi + 6 * e);
            if (hit) break;
        }
        batches.push_back({a, e});
        a = e;
    }
    // do ANY two boxes overlap?  Then later crops must see the grid as it was: read a copy.
    bool any_overlap = false;
    for (i64 i = 0; i < ncomp && !any_overlap; ++i)
        for (i64 j = i + 1; j < ncomp && !any_overlap; ++j) any_overlap = boxes_overlap(bbox_lo_hi + 6 * i, bbox_lo_hi + 6 * j);
    void* copy = nullptr;
    if (any_overlap) {
        PB3D_TRY(pb3d_dev_alloc(ctx, nvox3, &copy));
        hipError_t e = hipMemcpyAsync(copy, d_grid_rgb, nvox3, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { (void)pb3d_dev_free(ctx, copy); PB3D_HIP(e); }
    }
    std::vector<CropDesc> hd;
    std::vector<RotParams> hp;
    // host staging of every batch stays alive until the final synchronisation below
    std::vector<std::vector<CropDesc>> keep_d;
    std::vector<std::vector<RotParams>> keep_p;
    keep_d.reserve(batches.size()); keep_p.reserve(batches.size());
    int rc = PB3D_OK;
    auto run = [&]() -> int {
        size_t desc_bytes = 0, par_bytes = 0, tab_bytes = 0, slice_bytes = 16;
        for (auto& be : batches) {
            size_t cells = 0, swords = 0;
            for (i64 k = be.first; k < be.second; ++k) {
                const i64* b = bbox_lo_hi + 6 * k;
                cells += (size_t)((b[3] - b[0]) * (b[5] - b[2]));
                swords += (size_t)((b[3] - b[0]) * (b[5] - b[2])) * (size_t)((b[4] - b[1] + 31) / 32);
            }
            slice_bytes = std::max(slice_bytes, swords * sizeof(u32));
            const size_t n = (size_t)(be.second - be.first);
            desc_bytes = std::max(desc_bytes, n * sizeof(CropDesc));
            par_bytes = std::max(par_bytes, n * (size_t)(nrot > 0 ? nrot : 1) * sizeof(RotParams));
            tab_bytes = std::max(tab_bytes, cells * (size_t)(nrot > 0 ? nrot : 1) * sizeof(u32));
        }
        void *dd, *dt, *dsl;
        // every batch gets its own slice of the slot ([descriptors | step parameters], ONE upload): the upload of batch b + 1 must not
        // overwrite what batch b's kernels still read
        const size_t nb = batches.size();
        const size_t slice = ((desc_bytes + par_bytes) + 63) & ~(size_t)63;
        PB3D_TRY(pb3d_scratch(ctx, 36, slice * nb, &dd));
        PB3D_TRY(pb3d_scratch(ctx, 38, tab_bytes, &dt));
        PB3D_TRY(pb3d_scratch(ctx, 43, slice_bytes, &dsl));
        for (size_t bi = 0; bi < nb; ++bi) {
            const i64 a = batches[bi].first, e = batches[bi].second;
            keep_d.emplace_back(); keep_p.emplace_back();
            std::vector<CropDesc>& ds = keep_d.back();
            std::vector<RotParams>& ps = keep_p.back();
            int g0 = 0;
            u32 cell_off = 0, slice_off = 0;
            int maxcells = 0;
            for (i64 k = a; k < e; ++k) {
                const i64* b = bbox_lo_hi + 6 * k;
                CropDesc d;
                memset(&d, 0, sizeof(d));
                d.x0 = (int)b[0]; d.y0 = (int)b[1]; d.z0 = (int)b[2];
                d.Wc = (int)(b[3] - b[0]); d.Hc = (int)(b[4] - b[1]); d.Dc = (int)(b[5] - b[2]);
                d.id = (int)(k + 1); d.g0 = g0; d.ng = (d.Hc + 31) / 32; d.pitch = d.Dc | 1;
                d.mask_off = (u32)mask_off[k]; d.cell_off = cell_off; d.slice_off = slice_off;
                slice_off += (u32)(d.Wc * d.Dc) * (u32)d.ng;
                g0 += d.ng;
                cell_off += (u32)(d.Wc * d.Dc) * (u32)(nrot > 0 ? nrot : 1);
                maxcells = std::max(maxcells, d.Wc * d.Dc);
                ds.push_back(d);
                const i64 shape[3] = {d.Wc, d.Hc, d.Dc};
                for (int s = 0; s < nrot; ++s) {
                    double M[9], off[3];
                    PB3D_TRY(pb3d_rotinv((s + 1) * angle_interval, M));
                    PB3D_TRY(pb3d_offset(M, shape, off));
                    ps.push_back(RotParams{M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]});
                }
            }
            static_assert(sizeof(CropDesc) == sizeof(RotParams), "descriptors and parameters share one staging vector");
            const size_t nd = ds.size();
            for (const RotParams& rp : ps) { CropDesc as; memcpy(&as, &rp, sizeof(as)); ds.push_back(as); }     // [descriptors | parameters]
            CropDesc* ddb = (CropDesc*)((char*)dd + slice * bi);
            RotParams* dpb = (RotParams*)(ddb + nd);
            PB3D_TRY(pb3d_h2d_async(ctx, ddb, ds.data(), ds.size() * sizeof(CropDesc)));
            const int n = (int)(e - a);
            if (nrot > 0) {
                dim3 gridc((unsigned)std::min((maxcells + 255) / 256, 64), (unsigned)(n * nrot));
                PB3D_REQUIRE(gridc.y <= 65535u, "pb3d_guided_carve: too many (component, step) pairs in a batch");
                hipLaunchKernelGGL(k_crop_cells, gridc, dim3(256), 0, ctx->stream, (const CropDesc*)ddb, (const RotParams*)dpb, nrot, (u32*)dt);
            }
            PB3D_REQUIRE(g0 <= 65535, "pb3d_guided_carve: too many plane groups in a batch");
            hipLaunchKernelGGL(k_crop_slice, dim3((unsigned)((maxcells + 255) / 256), (unsigned)g0), dim3(256), 0, ctx->stream, (const u8*)(copy ? copy : d_grid_rgb), H, D,
                               (const CropDesc*)ddb, n, (u32*)dsl, C, W * H * D);
            const int nbt = (maxcells + 8 * GTHREADS - 1) / (8 * GTHREADS);
            PB3D_REQUIRE(nbt <= 5, "pb3d_guided_carve: crop of %d cells (internal limit)", maxcells);
            auto chain = nbt <= 1 ? k_crop_chain<1> : nbt == 2 ? k_crop_chain<2> : nbt == 3 ? k_crop_chain<3> : k_crop_chain<5>;
            hipLaunchKernelGGL(chain, dim3((unsigned)g0), dim3(GTHREADS), lds_max, ctx->stream, (const u8*)(copy ? copy : d_grid_rgb), d_grid_rgb,
                               d_labels, H, D, (const CropDesc*)ddb, n, (const u8*)d_masks, (const u32*)dt, nrot, (unsigned long long*)d_counts, copy ? 1 : 0, C, W * H * D, ctx->tune_crop_ablate, (const u32*)dsl, mb64, mbP);
            PB3D_CHECK_LAUNCH();
        }
        static_assert(sizeof(unsigned long long) == sizeof(int64_t), "counts are 64-bit");
        if (!queued) PB3D_HIP(hipMemcpyAsync(carved_counts, d_counts, (size_t)ncomp * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        return PB3D_OK;
    };
    rc = run();
    if (queued) {
        // (a block handed back to the pool is only ever given to later work on the same stream, in order)
        if (copy) (void)pb3d_dev_free(ctx, copy);
        if (rc != PB3D_OK) return rc;
        *took = 1;
        return PB3D_OK;
    }
    ++ctx->sync_count;
    hipError_t es = hipStreamSynchronize(ctx->stream);          // the counts are the caller's (the printed log)
    if (copy) (void)pb3d_dev_free(ctx, copy);
    if (rc != PB3D_OK) return rc;
    PB3D_HIP(es);
    *took = 1;
    return PB3D_OK;
}

extern "C" {

int pb3d_guided_carve_dev(pb3d_ctx* ctx, uint8_t* d_grid_rgb, const int32_t* d_labels, int64_t W, int64_t H, int64_t D, int64_t ncomp,
                          const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off, int64_t mask_bytes, int angle_interval,
                          int64_t* carved_counts, int* took) {
    return guided_carve_impl(ctx, d_grid_rgb, d_labels, W, H, D, ncomp, bbox_lo_hi, masks, mask_off, mask_bytes, angle_interval, carved_counts, took, 3, 0);
}

// ... for colour `color_index` of a labelling of several colours (pb3d_label_colors_stats_dev / pb3d_label_values_stats_dev)
int pb3d_guided_carve_color_dev(pb3d_ctx* ctx, uint8_t* d_grid, const int32_t* d_labels, int color_index, int channels, int64_t W, int64_t H,
                                int64_t D, int64_t ncomp, const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off,
                                int64_t mask_bytes, int angle_interval, int64_t* carved_counts, int* took) {
    PB3D_REQUIRE(channels == 1 || channels == 3, "pb3d_guided_carve_color: channels is 1 (labels) or 3 (colours)");
    PB3D_REQUIRE(ctx && color_index >= 0 && color_index < PB3D_CCL_MAX_COLORS, "pb3d_guided_carve_color: bad colour index");
    return guided_carve_impl(ctx, d_grid, d_labels, W, H, D, ncomp, bbox_lo_hi, masks, mask_off, mask_bytes, angle_interval, carved_counts, took, channels,
                             color_index);
}

// ... QUEUED: the call returns without waiting, the "carved voxels" counts are accumulated in d_counts (device, ncomp entries, cleared by
// the call); every host argument may be reused on return
int pb3d_guided_carve_queue_dev(pb3d_ctx* ctx, uint8_t* d_grid, const int32_t* d_labels, int color_index, int channels, int64_t W, int64_t H,
                                int64_t D, int64_t ncomp, const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off,
                                int64_t mask_bytes, int angle_interval, int64_t* d_counts, int* took) {
    PB3D_REQUIRE(channels == 1 || channels == 3, "pb3d_guided_carve_queue: channels is 1 (labels) or 3 (colours)");
    PB3D_REQUIRE(ctx && color_index >= 0 && color_index < PB3D_CCL_MAX_COLORS && d_counts, "pb3d_guided_carve_queue: bad argument");
    return guided_carve_impl(ctx, d_grid, d_labels, W, H, D, ncomp, bbox_lo_hi, masks, mask_off, mask_bytes, angle_interval, nullptr, took, channels,
                             color_index, d_counts);
}

// the same on a 1-byte LABEL volume (row N3): occupancy = label != 0, cleared voxels get label 0
int pb3d_guided_carve_label_dev(pb3d_ctx* ctx, uint8_t* d_grid_lab, const int32_t* d_labels, int64_t W, int64_t H, int64_t D, int64_t ncomp,
                                const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off, int64_t mask_bytes, int angle_interval,
                                int64_t* carved_counts, int* took) {
    return guided_carve_impl(ctx, d_grid_lab, d_labels, W, H, D, ncomp, bbox_lo_hi, masks, mask_off, mask_bytes, angle_interval, carved_counts, took, 1, 0);
}

}  // extern "C"
