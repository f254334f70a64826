// Host-pointer entry points: what the NumPy shim binds.  Each stages the caller's buffers through
// the context's device scratch (slots 0-3), runs the device-resident op on the context's stream and
// copies the result back; the call returns when the caller's output buffer is complete.
#include "pb3d_internal.h"

namespace {

int up(pb3d_ctx* ctx, int slot, const void* h, size_t bytes, void** d) {
    PB3D_TRY(pb3d_scratch(ctx, slot, bytes, d));
    if (bytes) PB3D_HIP(hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PB3D_OK;
}

int down(pb3d_ctx* ctx, void* h, const void* d, size_t bytes) {
    if (bytes) PB3D_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    return PB3D_OK;
}

}  // namespace

extern "C" {

int pb3d_carve_mask(pb3d_ctx* ctx, const uint8_t* grid, int64_t W, int64_t H, int64_t D, int C,
                    const uint8_t* mask_wh, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_carve_mask: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0 && (C == 1 || C == 3), "pb3d_carve_mask: bad shape");
    const size_t nb = (size_t)(W * H * D * C);
    if (nb == 0) return PB3D_OK;
    PB3D_REQUIRE(grid && mask_wh && out, "pb3d_carve_mask: null buffer");
    void *dg, *dm, *dout;
    PB3D_TRY(up(ctx, 0, grid, nb, &dg));
    PB3D_TRY(up(ctx, 2, mask_wh, (size_t)(W * H), &dm));
    PB3D_TRY(pb3d_scratch(ctx, 1, nb, &dout));
    PB3D_TRY(pb3d_carve_mask_dev(ctx, (const u8*)dg, W, H, D, C, (const u8*)dm, (u8*)dout));
    return down(ctx, out, dout, nb);
}

int pb3d_rotate_carve(pb3d_ctx* ctx, const uint8_t* occ, int64_t W, int64_t H, int64_t D,
                      const double M[9], const double off[3], const uint8_t* mask_wh, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_rotate_carve: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_rotate_carve: bad shape");
    const size_t nb = (size_t)(W * H * D);
    if (nb == 0) return PB3D_OK;
    PB3D_REQUIRE(occ && out, "pb3d_rotate_carve: null buffer");
    void *dg, *dm = nullptr, *dout;
    PB3D_TRY(up(ctx, 0, occ, nb, &dg));
    if (mask_wh) PB3D_TRY(up(ctx, 2, mask_wh, (size_t)(W * H), &dm));
    PB3D_TRY(pb3d_scratch(ctx, 1, nb, &dout));
    PB3D_TRY(pb3d_rotate_carve_dev(ctx, (const u8*)dg, W, H, D, M, off, (const u8*)dm, (u8*)dout));
    return down(ctx, out, dout, nb);
}

int pb3d_process_grid(pb3d_ctx* ctx, const uint8_t* occ, int64_t W, int64_t H, int64_t D,
                      const uint8_t* mask_wh, int angle_interval, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_process_grid: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_process_grid: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_process_grid: angle_interval must be a positive integer (got %d)", angle_interval);
    const size_t nb = (size_t)(W * H * D);
    if (nb == 0) return PB3D_OK;
    PB3D_REQUIRE(occ && mask_wh && out, "pb3d_process_grid: null buffer");
    void *dg, *dm, *dout, *dtmp;
    PB3D_TRY(up(ctx, 0, occ, nb, &dg));
    PB3D_TRY(up(ctx, 2, mask_wh, (size_t)(W * H), &dm));
    PB3D_TRY(pb3d_scratch(ctx, 1, nb, &dout));
    PB3D_TRY(pb3d_scratch(ctx, 3, nb, &dtmp));
    PB3D_TRY(pb3d_process_grid_dev(ctx, (const u8*)dg, W, H, D, (const u8*)dm, angle_interval, (u8*)dout, (u8*)dtmp));
    return down(ctx, out, dout, nb);
}

int pb3d_occupancy(pb3d_ctx* ctx, const uint8_t* grid_rgb, int64_t nvox, uint8_t* occ) {
    PB3D_REQUIRE(ctx != nullptr && nvox >= 0, "pb3d_occupancy: bad argument");
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(grid_rgb && occ, "pb3d_occupancy: null buffer");
    void *dg, *dout;
    PB3D_TRY(up(ctx, 0, grid_rgb, (size_t)nvox * 3, &dg));
    PB3D_TRY(pb3d_scratch(ctx, 1, (size_t)nvox, &dout));
    PB3D_TRY(pb3d_occupancy_dev(ctx, (const u8*)dg, nvox, (u8*)dout));
    return down(ctx, occ, dout, (size_t)nvox);
}

int pb3d_color_apply(pb3d_ctx* ctx, const uint8_t* carved, int64_t W, int64_t H, int64_t D,
                     const uint8_t* rgb_hw3, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr && W >= 0 && H >= 0 && D >= 0, "pb3d_color_apply: bad shape");
    const size_t nv = (size_t)(W * H * D);
    if (nv == 0) return PB3D_OK;
    PB3D_REQUIRE(carved && rgb_hw3 && out, "pb3d_color_apply: null buffer");
    void *dg, *dm, *dout;
    PB3D_TRY(up(ctx, 0, carved, nv, &dg));
    PB3D_TRY(up(ctx, 2, rgb_hw3, (size_t)(W * H * 3), &dm));
    PB3D_TRY(pb3d_scratch(ctx, 1, nv * 3, &dout));
    PB3D_TRY(pb3d_color_apply_dev(ctx, (const u8*)dg, W, H, D, (const u8*)dm, (u8*)dout));
    return down(ctx, out, dout, nv * 3);
}

int pb3d_global_carve(pb3d_ctx* ctx, const uint8_t* bin_hw, const uint8_t* rgb_hw3, int64_t h, int64_t w,
                      int angle_interval, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr && h >= 0 && w >= 0, "pb3d_global_carve: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_global_carve: angle_interval must be a positive integer (got %d)", angle_interval);
    const size_t nv = (size_t)(w * h * w);
    if (nv == 0) return PB3D_OK;
    PB3D_REQUIRE(bin_hw && rgb_hw3 && out, "pb3d_global_carve: null buffer");
    void *db, *dr, *dout;
    PB3D_TRY(up(ctx, 0, bin_hw, (size_t)(h * w), &db));
    PB3D_TRY(up(ctx, 2, rgb_hw3, (size_t)(h * w * 3), &dr));
    PB3D_TRY(pb3d_scratch(ctx, 1, nv * 3, &dout));
    PB3D_TRY(pb3d_global_carve_dev(ctx, (const u8*)db, (const u8*)dr, h, w, angle_interval, 0, w, (u8*)dout));
    return down(ctx, out, dout, nv * 3);
}

int pb3d_part_carve(pb3d_ctx* ctx, const uint8_t* colored, int64_t W, int64_t H, int64_t D,
                    const uint8_t* mask_sub, const uint8_t* mask_carve, const int* job_angle,
                    const int* job_skip, int njobs, uint8_t* out) {
    PB3D_REQUIRE(ctx != nullptr && W >= 0 && H >= 0 && D >= 0 && njobs >= 0, "pb3d_part_carve: bad shape");
    const size_t nv = (size_t)(W * H * D);
    if (nv == 0) return PB3D_OK;
    PB3D_REQUIRE(colored && out, "pb3d_part_carve: null buffer");
    void *dg, *dms = nullptr, *dmc = nullptr, *dout;
    PB3D_TRY(up(ctx, 0, colored, nv * 3, &dg));
    if (njobs > 0) {
        PB3D_REQUIRE(mask_sub && mask_carve, "pb3d_part_carve: null job masks");
        PB3D_TRY(up(ctx, 2, mask_sub, (size_t)(W * H) * njobs, &dms));
        PB3D_TRY(up(ctx, 3, mask_carve, (size_t)(W * H) * njobs, &dmc));
    }
    PB3D_TRY(pb3d_scratch(ctx, 1, nv * 3, &dout));
    PB3D_TRY(pb3d_part_carve_dev(ctx, (const u8*)dg, W, H, D, (const u8*)dms, (const u8*)dmc, job_angle, job_skip, njobs,
                                 (u8*)dout));
    return down(ctx, out, dout, nv * 3);
}

int pb3d_points_count(pb3d_ctx* ctx, const uint8_t* grid, int64_t A0, int64_t A1, int64_t A2, int C,
                      const uint8_t* colors, int ncolors, int stride, int64_t* n) {
    PB3D_REQUIRE(ctx != nullptr && n != nullptr, "pb3d_points_count: null argument");
    PB3D_REQUIRE(A0 >= 0 && A1 >= 0 && A2 >= 0 && (C == 1 || C == 3), "pb3d_points_count: bad shape");
    PB3D_REQUIRE(ncolors >= 0 && ncolors <= 32, "pb3d_points_count: at most 32 colours");
    ctx->pts.valid = false;
    const size_t nb = (size_t)(A0 * A1 * A2 * C);
    void* dg = nullptr;
    if (nb) {
        PB3D_REQUIRE(grid != nullptr, "pb3d_points_count: null grid");
        PB3D_TRY(up(ctx, 0, grid, nb, &dg));
    }
    // One pass when the worst case fits a modest device buffer: the points are extracted while they are counted (decoupled
    // look-back, csrc/points.hip) and pb3d_points_fill only downloads them -- the grid is read once instead of twice.
    const i64 nvox = A0 * A1 * A2;
    ctx->pts.extracted = false;
    // (measured on MI355X the look-back costs more than the second read of the grid saves -- 4.1 ms against 2.4 ms at 1024^3, flag round
    //  trips between XCDs are microseconds -- so the two-pass protocol stays the default; tune misc2 = 3 selects the one-pass form)
    if (ctx->tune_points_onepass == 1 && stride == 1 && nvox > 0 && (C == 3 || ncolors == 0) && (size_t)nvox * (12 + (size_t)C) <= ((size_t)8 << 30)) {
        void *dp, *dc;
        PB3D_TRY(pb3d_scratch(ctx, 1, (size_t)nvox * 3 * sizeof(float), &dp));
        PB3D_TRY(pb3d_scratch(ctx, 3, (size_t)nvox * C, &dc));
        const int rc = pb3d_points_extract_dev(ctx, (const u8*)dg, A0, A1, A2, C, colors, ncolors, nvox, (float*)dp, (u8*)dc, n);
        if (rc == PB3D_OK) ctx->pts.extracted = true;
        else if (rc != PB3D_EUNSUPPORTED) return rc;
    }
    if (!ctx->pts.extracted) PB3D_TRY(pb3d_points_count_dev(ctx, (const u8*)dg, A0, A1, A2, C, colors, ncolors, stride, n));
    ctx->pts.A0 = A0; ctx->pts.A1 = A1; ctx->pts.A2 = A2; ctx->pts.C = C;
    ctx->pts.ncolors = ncolors; ctx->pts.stride = stride; ctx->pts.n = *n;
    if (ncolors) memcpy(ctx->pts.colors, colors, (size_t)3 * ncolors);
    ctx->pts.valid = true;
    return PB3D_OK;
}

int pb3d_points_fill(pb3d_ctx* ctx, int64_t n, float* pts, uint8_t* cols) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_points_fill: null context");
    PB3D_REQUIRE(ctx->pts.valid, "pb3d_points_fill: call pb3d_points_count first");
    PB3D_REQUIRE(n == ctx->pts.n, "pb3d_points_fill: n=%lld does not match the counted %lld", (long long)n, (long long)ctx->pts.n);
    ctx->pts.valid = false;
    if (n == 0) return PB3D_OK;
    PB3D_REQUIRE(pts && cols, "pb3d_points_fill: null buffer");
    void *dp, *dc;
    if (ctx->pts.extracted) {                    // already in scratch 1 / 3 (pb3d_points_count)
        dp = ctx->scratch[1]; dc = ctx->scratch[3];
    } else {
        PB3D_TRY(pb3d_scratch(ctx, 1, (size_t)n * 3 * sizeof(float), &dp));
        PB3D_TRY(pb3d_scratch(ctx, 3, (size_t)n * ctx->pts.C, &dc));
        PB3D_TRY(pb3d_points_fill_dev(ctx, (const u8*)ctx->scratch[0], ctx->pts.A0, ctx->pts.A1, ctx->pts.A2, ctx->pts.C,
                                      ctx->pts.colors, ctx->pts.ncolors, ctx->pts.stride, n, (float*)dp, (u8*)dc));
    }
    PB3D_HIP(hipMemcpyAsync(pts, dp, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return down(ctx, cols, dc, (size_t)n * ctx->pts.C);
}

int pb3d_project(pb3d_ctx* ctx, const void* pts, int pts_f64, const uint8_t* cols, int64_t n,
                 const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4],
                 int Himg, int Wimg, uint8_t* img) {
    PB3D_REQUIRE(ctx != nullptr && n >= 0 && Himg >= 0 && Wimg >= 0, "pb3d_project: bad argument");
    const size_t npix = (size_t)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(img && (n == 0 || (pts && cols)), "pb3d_project: null buffer");
    void *dp = nullptr, *dc = nullptr, *dimg;
    if (n) {
        PB3D_TRY(up(ctx, 0, pts, (size_t)n * 3 * (pts_f64 ? 8 : 4), &dp));
        PB3D_TRY(up(ctx, 2, cols, (size_t)n * 3, &dc));
    }
    PB3D_TRY(pb3d_scratch(ctx, 1, npix * 3, &dimg));
    PB3D_TRY(pb3d_project_dev(ctx, dp, pts_f64, (const u8*)dc, n, R, cam, f, cx, cy, prec, Himg, Wimg, (u8*)dimg));
    return down(ctx, img, dimg, npix * 3);
}

int pb3d_partwise_iou(pb3d_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t npix,
                      const uint8_t* colors, int ncolors, int64_t* inter, int64_t* uni) {
    PB3D_REQUIRE(ctx != nullptr && npix >= 0, "pb3d_partwise_iou: bad argument");
    void *da = nullptr, *db = nullptr;
    if (npix) {
        PB3D_REQUIRE(a && b, "pb3d_partwise_iou: null buffer");
        PB3D_TRY(up(ctx, 0, a, (size_t)npix * 3, &da));
        PB3D_TRY(up(ctx, 2, b, (size_t)npix * 3, &db));
    }
    return pb3d_partwise_iou_dev(ctx, (const u8*)da, (const u8*)db, npix, colors, ncolors, inter, uni);
}

}  // extern "C"
