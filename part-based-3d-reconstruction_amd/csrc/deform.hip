// K9: part-wise deformation of voxel coordinates (notebook-3 loop, BASELINE config 5).
//
// reference utils/deformation_estimation.py:70-98 (deform_coords closure): seven jitters of the part's
// points, each centred on its own mean, scaled/shifted per axis in float64, rounded half-to-even,
// then np.unique(axis=0).  On the device the sort-based dedup becomes: evaluate -> bounding box ->
// mark a dense byte volume indexed [x][y][z] -> ordered stream compaction (csrc/points.hip), whose
// C order IS the lexicographic (x,y,z) order np.unique returns.
// The points are voxel indices (integer-valued float32), so every jittered coordinate is a multiple of
// 0.25 and the float64 mean NumPy computes is exact_sum / n whatever the summation order; the exact
// sums are taken with integer atomics.  Non-integer inputs are rejected (PB3D_EUNSUPPORTED).
#include "pb3d_internal.h"
#include "project_point.h"

namespace {

struct DeformParams {
    double sxz, sy, kx, ky, kz;
    double ctr[7][3];
};

__constant__ double k_off[7][3] = {{0, 0, 0}, {0.25, 0, 0}, {-0.25, 0, 0}, {0, 0.25, 0}, {0, -0.25, 0}, {0, 0, 0.25}, {0, 0, -0.25}};

__device__ __forceinline__ double sgn(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : v); }

__device__ __forceinline__ void deform_eval(const DeformParams& P, const float* __restrict__ p, int j, i64 r[3]) {
    const double c0 = __dsub_rn(__dadd_rn((double)p[0], k_off[j][0]), P.ctr[j][0]);
    const double c1 = __dsub_rn(__dadd_rn((double)p[1], k_off[j][1]), P.ctr[j][1]);
    const double c2 = __dsub_rn(__dadd_rn((double)p[2], k_off[j][2]), P.ctr[j][2]);
    const double d0 = __dadd_rn(__dmul_rn(c0, P.sxz), __dmul_rn(P.kx, sgn(c0)));
    const double d1 = __dsub_rn(__dmul_rn(c1, P.sy), P.ky);
    const double d2 = __dadd_rn(__dmul_rn(c2, P.sxz), __dmul_rn(P.kz, sgn(c2)));
    r[0] = (i64)rint(__dadd_rn(d0, P.ctr[j][0]));
    r[1] = (i64)rint(__dadd_rn(d1, P.ctr[j][1]));
    r[2] = (i64)rint(__dadd_rn(d2, P.ctr[j][2]));
}

// acc[0..2] = integer sums per axis, acc[3] = count of rejected (non-integer / huge) coordinates
__global__ __launch_bounds__(256) void k_deform_sum(const float* __restrict__ pts, i64 n, unsigned long long* __restrict__ acc) {
    i64 s0 = 0, s1 = 0, s2 = 0, bad = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const float a = pts[3 * i], b = pts[3 * i + 1], c = pts[3 * i + 2];
        const bool ok = a == rintf(a) && b == rintf(b) && c == rintf(c) && fabsf(a) < 4194304.f && fabsf(b) < 4194304.f &&
                        fabsf(c) < 4194304.f;
        if (ok) { s0 += (i64)a; s1 += (i64)b; s2 += (i64)c; } else ++bad;
    }
    atomicAdd(&acc[0], (unsigned long long)s0); atomicAdd(&acc[1], (unsigned long long)s1);
    atomicAdd(&acc[2], (unsigned long long)s2);
    if (bad) atomicAdd(&acc[3], (unsigned long long)bad);
}

__global__ __launch_bounds__(256) void k_deform_bbox(const float* __restrict__ pts, i64 n, DeformParams P, int* __restrict__ bb) {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    int bad = 0;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < 7 * n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = t / 7; const int j = (int)(t - 7 * i);
        i64 r[3];
        deform_eval(P, pts + 3 * i, j, r);
        for (int a = 0; a < 3; ++a) {
            if (r[a] < -(1ll << 30) || r[a] > (1ll << 30)) { bad = 1; continue; }
            lo[a] = (int)r[a] < lo[a] ? (int)r[a] : lo[a];
            hi[a] = (int)r[a] > hi[a] ? (int)r[a] : hi[a];
        }
    }
    for (int a = 0; a < 3; ++a) {
        if (lo[a] != 0x7fffffff) atomicMin(&bb[a], lo[a]);
        if (hi[a] != (int)0x80000000) atomicMax(&bb[3 + a], hi[a]);
    }
    if (bad) atomicOr(&bb[6], 1);
}

__global__ __launch_bounds__(256) void k_deform_mark(const float* __restrict__ pts, i64 n, DeformParams P, int ox, int oy, int oz,
                                                     i64 Y, i64 Z, u8* __restrict__ mark) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < 7 * n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = t / 7; const int j = (int)(t - 7 * i);
        i64 r[3];
        deform_eval(P, pts + 3 * i, j, r);
        mark[((r[0] - ox) * Y + (r[1] - oy)) * Z + (r[2] - oz)] = 1;
    }
}

// compaction output (a2,a1,a0) = (z,y,x) local float32 -> int64 (x,y,z) global rows
__global__ __launch_bounds__(256) void k_pts_to_coords(const float* __restrict__ pts, i64 m, int ox, int oy, int oz,
                                                       i64* __restrict__ coords) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        coords[3 * i + 0] = (i64)pts[3 * i + 2] + ox;
        coords[3 * i + 1] = (i64)pts[3 * i + 1] + oy;
        coords[3 * i + 2] = (i64)pts[3 * i + 0] + oz;
    }
}

// voxel_def[z,y,x] = rgb for every in-bounds deformed coordinate (grid (A0,A1,A2,3) indexed [z][y][x])
__global__ __launch_bounds__(256) void k_deform_paint(const float* __restrict__ pts, i64 n, DeformParams P, i64 A0, i64 A1, i64 A2,
                                                      u8 cr, u8 cg, u8 cb, u8* __restrict__ grid) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < 7 * n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = t / 7; const int j = (int)(t - 7 * i);
        i64 r[3];
        deform_eval(P, pts + 3 * i, j, r);
        if (r[0] < 0 || r[0] >= A2 || r[1] < 0 || r[1] >= A1 || r[2] < 0 || r[2] >= A0) continue;
        u8* o = grid + ((r[2] * A1 + r[1]) * A2 + r[0]) * 3;
        o[0] = cr; o[1] = cg; o[2] = cb;
    }
}

// grid[z,y,x] = cols[k] for coordinate rows (x,y,z); rows must be unique (np.unique output) and in bounds
__global__ __launch_bounds__(256) void k_scatter_colors(const i64* __restrict__ coords, const u8* __restrict__ cols, i64 m, i64 A0,
                                                        i64 A1, i64 A2, u8* __restrict__ grid, int* __restrict__ oob) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (i64)gridDim.x * blockDim.x) {
        const i64 x = coords[3 * i], y = coords[3 * i + 1], z = coords[3 * i + 2];
        if (x < 0 || x >= A2 || y < 0 || y >= A1 || z < 0 || z >= A0) { atomicOr(oob, 1); continue; }
        u8* o = grid + ((z * A1 + y) * A2 + x) * 3;
        o[0] = cols[3 * i]; o[1] = cols[3 * i + 1]; o[2] = cols[3 * i + 2];
    }
}


// ------------------------------------------------------------------------------------------------
// Row N4, deformation side: K deform tuples per launch.  The part-wise grid search of reference
// utils/deformation_estimation.py:148-258 (and the slider loop :100-146 it automates) evaluates, for ONE part and a FIXED
// camera, deform_coords -> bounds filter -> project_colored_voxels -> compute_partwise_iou for hundreds of tuples.  All points
// of a part carry the part colour, so the projected image is the SET of pixels hit: neither np.unique nor the
// last-writer-wins order can change it.  One launch evaluates every (tuple, point, jitter), projects the in-bounds integer
// coordinates (as float32, :128) and marks the pixel in the tuple's private byte image; a second launch counts
// intersection / union against the part's pixels of the image.  The 7 jitter centres do not depend on the tuple.
// ------------------------------------------------------------------------------------------------
struct DeformTuple { double sxz, sy, kx, ky, kz; };

__global__ __launch_bounds__(256) void k_deform_project_batch(const float* __restrict__ pts, i64 n, const DeformTuple* __restrict__ tuples,
                                                              DeformParams C, pb3d_proj::ProjParams P, i64 A0, i64 A1, i64 A2, i64 npix,
                                                              u8* __restrict__ marks, unsigned long long* __restrict__ nvalid) {
    DeformParams D = C;
    const DeformTuple t = tuples[blockIdx.y];
    D.sxz = t.sxz; D.sy = t.sy; D.kx = t.kx; D.ky = t.ky; D.kz = t.kz;
    u8* mark = marks + (i64)blockIdx.y * npix;
    unsigned long long mine = 0;
    for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < 7 * n; e += (i64)gridDim.x * blockDim.x) {
        const i64 i = e / 7; const int j = (int)(e - 7 * i);
        i64 r[3];
        deform_eval(D, pts + 3 * i, j, r);
        if (r[0] < 0 || r[0] >= A2 || r[1] < 0 || r[1] >= A1 || r[2] < 0 || r[2] >= A0) continue;
        ++mine;
        const double q[3] = {(double)(float)r[0], (double)(float)r[1], (double)(float)r[2]};   // coords_def.astype(np.float32): exact below 2^24
        int ui, vi;
        if (pb3d_proj::project_xyz<0>(P, q, &ui, &vi)) mark[(i64)vi * P.Wimg + ui] = 1;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&nvalid[blockIdx.y], mine);
}

__global__ __launch_bounds__(256) void k_deform_iou_batch(const u8* __restrict__ marks, const u8* __restrict__ seg, i64 npix, u8 c0, u8 c1, u8 c2,
                                                          unsigned long long* __restrict__ counts) {
    const u8* mark = marks + (i64)blockIdx.y * npix;
    const bool black = c0 == 0 && c1 == 0 && c2 == 0;                       // an unprojected pixel is (0,0,0): it "matches" a black part
    unsigned long long ni = 0, nu = 0;
    for (i64 px = (i64)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (i64)gridDim.x * blockDim.x) {
        const bool a = mark[px] != 0 || black;
        const bool b = seg[3 * px] == c0 && seg[3 * px + 1] == c1 && seg[3 * px + 2] == c2;
        ni += (a && b); nu += (a || b);
    }
    for (int o = 32; o > 0; o >>= 1) { ni += __shfl_xor(ni, o); nu += __shfl_xor(nu, o); }
    if ((threadIdx.x & 63) == 0) {
        if (ni) atomicAdd(&counts[2 * blockIdx.y], ni);
        if (nu) atomicAdd(&counts[2 * blockIdx.y + 1], nu);
    }
}

int centers(pb3d_ctx* ctx, const float* d_pts, i64 n, double sxz, double sy, double kx, double ky, double kz, DeformParams* P) {
    void* acc;
    PB3D_TRY(pb3d_scratch(ctx, 11, 8 * sizeof(unsigned long long), &acc));
    PB3D_HIP(hipMemsetAsync(acc, 0, 8 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_deform_sum, dim3(pb3d_stream_blocks(ctx, n, 256, 4)), dim3(256), 0, ctx->stream, d_pts, n,
                       (unsigned long long*)acc);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    const long long* h = (const long long*)ctx->pinned;
    if (h[3] != 0) {
        pb3d_set_error("pb3d_deform: %lld point coordinates are not voxel indices (integer-valued, |v| < 2^22)", h[3]);
        return PB3D_EUNSUPPORTED;
    }
    static const double offs[7][3] = {{0, 0, 0}, {0.25, 0, 0}, {-0.25, 0, 0}, {0, 0.25, 0}, {0, -0.25, 0}, {0, 0, 0.25}, {0, 0, -0.25}};
    P->sxz = sxz; P->sy = sy; P->kx = kx; P->ky = ky; P->kz = kz;
    for (int j = 0; j < 7; ++j)
        for (int a = 0; a < 3; ++a) {
            // sum of (p_i + off): every term and partial sum is a multiple of 0.25 below 2^53 -> exact
            const double total = (double)h[a] + (double)n * offs[j][a];
            P->ctr[j][a] = total / (double)n;
        }
    return PB3D_OK;
}

}  // namespace

extern "C" {

int pb3d_deform_count_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                          int64_t* n_unique) {
    PB3D_REQUIRE(ctx && n_unique && n >= 0, "pb3d_deform_count: bad argument");
    ctx->deform.valid = false;
    *n_unique = 0;
    if (n == 0) { ctx->deform.n = 0; ctx->deform.valid = true; return PB3D_OK; }
    PB3D_REQUIRE(d_pts != nullptr, "pb3d_deform_count: null points");
    DeformParams P;
    PB3D_TRY(centers(ctx, d_pts, n, sxz, sy, kx, ky, kz, &P));
    void* bbv;
    PB3D_TRY(pb3d_scratch(ctx, 11, 8 * sizeof(unsigned long long), &bbv));
    const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
    PB3D_HIP(hipMemcpyAsync(bbv, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = pb3d_stream_blocks(ctx, 7 * n, 256, 8);
    hipLaunchKernelGGL(k_deform_bbox, dim3(blocks), dim3(256), 0, ctx->stream, d_pts, n, P, (int*)bbv);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, bbv, sizeof(init), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    const int* bb = (const int*)ctx->pinned;
    if (bb[6]) {
        pb3d_set_error("pb3d_deform: deformed coordinates exceed +-2^30");
        return PB3D_EUNSUPPORTED;
    }
    const i64 X = (i64)bb[3] - bb[0] + 1, Y = (i64)bb[4] - bb[1] + 1, Z = (i64)bb[5] - bb[2] + 1;
    PB3D_REQUIRE(X > 0 && Y > 0 && Z > 0 && (double)X * (double)Y * (double)Z <= 34359738368.0,
                 "pb3d_deform: deformed bounding box %lld x %lld x %lld is too large", (long long)X, (long long)Y, (long long)Z);
    void* mark;
    PB3D_TRY(pb3d_scratch(ctx, 12, (size_t)(X * Y * Z), &mark));
    PB3D_HIP(hipMemsetAsync(mark, 0, (size_t)(X * Y * Z), ctx->stream));
    hipLaunchKernelGGL(k_deform_mark, dim3(blocks), dim3(256), 0, ctx->stream, d_pts, n, P, bb[0], bb[1], bb[2], Y, Z, (u8*)mark);
    PB3D_CHECK_LAUNCH();
    ctx->deform.ox = bb[0]; ctx->deform.oy = bb[1]; ctx->deform.oz = bb[2];
    ctx->deform.X = X; ctx->deform.Y = Y; ctx->deform.Z = Z;
    PB3D_TRY(pb3d_points_count_dev(ctx, (const u8*)mark, X, Y, Z, 1, nullptr, 0, 1, n_unique));
    ctx->deform.n = *n_unique;
    ctx->deform.valid = true;
    return PB3D_OK;
}

int pb3d_deform_fill_dev(pb3d_ctx* ctx, int64_t n_unique, int64_t* d_coords) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_deform_fill: null context");
    PB3D_REQUIRE(ctx->deform.valid && ctx->deform.n == n_unique, "pb3d_deform_fill: call pb3d_deform_count first");
    ctx->deform.valid = false;
    if (n_unique == 0) return PB3D_OK;
    PB3D_REQUIRE(d_coords != nullptr, "pb3d_deform_fill: null output");
    void *pts, *cols;
    PB3D_TRY(pb3d_scratch(ctx, 13, (size_t)n_unique * 3 * sizeof(float), &pts));
    PB3D_TRY(pb3d_scratch(ctx, 14, (size_t)n_unique, &cols));
    PB3D_TRY(pb3d_points_fill_dev(ctx, (const u8*)ctx->scratch[12], ctx->deform.X, ctx->deform.Y, ctx->deform.Z, 1, nullptr, 0, 1,
                                  n_unique, (float*)pts, (u8*)cols));
    hipLaunchKernelGGL(k_pts_to_coords, dim3(pb3d_stream_blocks(ctx, n_unique, 256, 8)), dim3(256), 0, ctx->stream,
                       (const float*)pts, n_unique, ctx->deform.ox, ctx->deform.oy, ctx->deform.oz, (i64*)d_coords);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_deform_paint_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                          int64_t A0, int64_t A1, int64_t A2, const uint8_t rgb[3], uint8_t* d_grid) {
    PB3D_REQUIRE(ctx && rgb && n >= 0 && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_deform_paint: bad argument");
    if (n == 0 || A0 * A1 * A2 == 0) return PB3D_OK;
    PB3D_REQUIRE(d_pts && d_grid, "pb3d_deform_paint: null buffer");
    DeformParams P;
    PB3D_TRY(centers(ctx, d_pts, n, sxz, sy, kx, ky, kz, &P));
    hipLaunchKernelGGL(k_deform_paint, dim3(pb3d_stream_blocks(ctx, 7 * n, 256, 8)), dim3(256), 0, ctx->stream, d_pts, n, P, A0, A1, A2,
                       rgb[0], rgb[1], rgb[2], d_grid);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_scatter_colors_dev(pb3d_ctx* ctx, const int64_t* d_coords, const uint8_t* d_cols, int64_t m, int64_t A0, int64_t A1,
                            int64_t A2, uint8_t* d_grid) {
    PB3D_REQUIRE(ctx && m >= 0 && A0 >= 0 && A1 >= 0 && A2 >= 0, "pb3d_scatter_colors: bad argument");
    if (m == 0) return PB3D_OK;
    PB3D_REQUIRE(d_coords && d_cols && d_grid, "pb3d_scatter_colors: null buffer");
    void* flag;
    PB3D_TRY(pb3d_scratch(ctx, 11, 8 * sizeof(unsigned long long), &flag));
    PB3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_scatter_colors, dim3(pb3d_stream_blocks(ctx, m, 256, 8)), dim3(256), 0, ctx->stream, (const i64*)d_coords,
                       d_cols, m, A0, A1, A2, d_grid, (int*)flag);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    if (*(const int*)ctx->pinned) {
        pb3d_set_error("index out of bounds in pb3d_scatter_colors (NumPy would raise IndexError)");
        return PB3D_EINVAL;
    }
    return PB3D_OK;
}

int pb3d_deform_iou_batch_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, const double* deforms5, int ntuples, int64_t A0, int64_t A1,
                              int64_t A2, const pb3d_camera* cam, int Himg, int Wimg, const uint8_t* d_seg, const uint8_t color[3],
                              int64_t* inter, int64_t* uni, int64_t* nvalid) {
    PB3D_REQUIRE(ctx && n >= 0 && ntuples >= 0 && A0 >= 0 && A1 >= 0 && A2 >= 0 && Himg >= 0 && Wimg >= 0, "pb3d_deform_iou_batch: bad argument");
    PB3D_REQUIRE(ntuples == 0 || (deforms5 && cam && color && inter && uni && nvalid), "pb3d_deform_iou_batch: null argument");
    PB3D_REQUIRE(A0 < (1 << 24) && A1 < (1 << 24) && A2 < (1 << 24), "pb3d_deform_iou_batch: grid axis too long for float32 coordinates");
    for (int k = 0; k < ntuples; ++k) inter[k] = uni[k] = nvalid[k] = 0;
    const i64 npix = (i64)Himg * Wimg;
    if (ntuples == 0 || npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_seg && (n == 0 || d_pts), "pb3d_deform_iou_batch: null buffer");
    pb3d_proj::ProjParams P;
    PB3D_TRY(pb3d_proj::fill_proj(&P, 0, cam->R, cam->cam, cam->f, cam->cx, cam->cy, cam->prec, Himg, Wimg));
    DeformParams C;
    memset(&C, 0, sizeof(C));
    if (n > 0) PB3D_TRY(centers(ctx, d_pts, n, 0, 0, 0, 0, 0, &C));
    i64 kc = (256ll << 20) / npix;
    kc = kc < 1 ? 1 : (kc > 8192 ? 8192 : kc);
    if (kc > ntuples) kc = ntuples;
    void *marks, *dt, *cnt;
    PB3D_TRY(pb3d_scratch(ctx, 12, (size_t)(kc * npix), &marks));
    PB3D_TRY(pb3d_scratch(ctx, 20, (size_t)kc * sizeof(DeformTuple), &dt));
    PB3D_TRY(pb3d_scratch(ctx, 21, (size_t)kc * 3 * sizeof(unsigned long long), &cnt));
    unsigned long long* hc = (unsigned long long*)malloc((size_t)kc * 3 * sizeof(unsigned long long));
    if (!hc) { pb3d_set_error("pb3d_deform_iou_batch: out of host memory"); return PB3D_ENOMEM; }
    int rc = PB3D_OK;
    auto hipok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == PB3D_OK) { pb3d_set_error("%s failed: %s", what, hipGetErrorString(e)); rc = PB3D_ENODEVICE; }
    };
    for (i64 k0 = 0; k0 < ntuples && rc == PB3D_OK; k0 += kc) {
        const i64 kn = ntuples - k0 < kc ? ntuples - k0 : kc;
        // the caller's array outlives the synchronisation at the end of this pass (struct DeformTuple = 5 doubles)
        hipok(hipMemcpyAsync(dt, deforms5 + 5 * k0, (size_t)kn * sizeof(DeformTuple), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
        hipok(hipMemsetAsync(marks, 0, (size_t)(kn * npix), ctx->stream), "hipMemsetAsync");
        hipok(hipMemsetAsync(cnt, 0, (size_t)kn * 3 * sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
        if (rc != PB3D_OK) break;
        unsigned long long* d_counts = (unsigned long long*)cnt;
        unsigned long long* d_nvalid = d_counts + 2 * kn;
        if (n > 0) {
            i64 per = (7 * n + 255) / 256;
            const i64 want = ((i64)ctx->cus * 16 + kn - 1) / kn;
            if (per > want) per = want;
            if (per < 1) per = 1;
            hipLaunchKernelGGL(k_deform_project_batch, dim3((unsigned)per, (unsigned)kn), dim3(256), 0, ctx->stream, d_pts, n, (const DeformTuple*)dt, C,
                               P, A0, A1, A2, npix, (u8*)marks, d_nvalid);
            hipok(hipGetLastError(), "k_deform_project_batch");
        }
        i64 ib = (npix + 255) / 256;
        const i64 iwant = ((i64)ctx->cus * 8 + kn - 1) / kn;
        if (ib > iwant) ib = iwant;
        if (ib < 1) ib = 1;
        hipLaunchKernelGGL(k_deform_iou_batch, dim3((unsigned)ib, (unsigned)kn), dim3(256), 0, ctx->stream, (const u8*)marks, d_seg, npix, color[0],
                           color[1], color[2], d_counts);
        hipok(hipGetLastError(), "k_deform_iou_batch");
        hipok(hipMemcpyAsync(hc, cnt, (size_t)kn * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
        ++ctx->sync_count; hipok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        if (rc != PB3D_OK) break;
        for (i64 k = 0; k < kn; ++k) { inter[k0 + k] = (int64_t)hc[2 * k]; uni[k0 + k] = (int64_t)hc[2 * k + 1]; nvalid[k0 + k] = (int64_t)hc[2 * kn + k]; }
    }
    free(hc);
    return rc;
}

// host-pointer flavours
int pb3d_deform_count(pb3d_ctx* ctx, const float* pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                      int64_t* n_unique) {
    PB3D_REQUIRE(ctx && n >= 0, "pb3d_deform_count: bad argument");
    void* d = nullptr;
    if (n) {
        PB3D_REQUIRE(pts != nullptr, "pb3d_deform_count: null points");
        PB3D_TRY(pb3d_scratch(ctx, 0, (size_t)n * 3 * sizeof(float), &d));
        PB3D_HIP(hipMemcpyAsync(d, pts, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    return pb3d_deform_count_dev(ctx, (const float*)d, n, sxz, sy, kx, ky, kz, n_unique);
}

int pb3d_deform_fill(pb3d_ctx* ctx, int64_t n_unique, int64_t* coords) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_deform_fill: null context");
    if (n_unique == 0) return pb3d_deform_fill_dev(ctx, 0, nullptr);
    PB3D_REQUIRE(coords != nullptr, "pb3d_deform_fill: null output");
    void* d;
    PB3D_TRY(pb3d_scratch(ctx, 1, (size_t)n_unique * 3 * sizeof(i64), &d));
    PB3D_TRY(pb3d_deform_fill_dev(ctx, n_unique, (int64_t*)d));
    PB3D_HIP(hipMemcpyAsync(coords, d, (size_t)n_unique * 3 * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    return PB3D_OK;
}

}  // extern "C"
