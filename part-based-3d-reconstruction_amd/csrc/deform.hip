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

int centers(pb3d_ctx* ctx, const float* d_pts, i64 n, double sxz, double sy, double kx, double ky, double kz, DeformParams* P) {
    void* acc;
    PB3D_TRY(pb3d_scratch(ctx, 11, 8 * sizeof(unsigned long long), &acc));
    PB3D_HIP(hipMemsetAsync(acc, 0, 8 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_deform_sum, dim3(pb3d_stream_blocks(ctx, n, 256, 4)), dim3(256), 0, ctx->stream, d_pts, n,
                       (unsigned long long*)acc);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, acc, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
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
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
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
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
    if (*(const int*)ctx->pinned) {
        pb3d_set_error("index out of bounds in pb3d_scatter_colors (NumPy would raise IndexError)");
        return PB3D_EINVAL;
    }
    return PB3D_OK;
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
    PB3D_HIP(hipStreamSynchronize(ctx->stream));
    return PB3D_OK;
}

}  // extern "C"
