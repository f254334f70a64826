// Shared by project.hip and deform.hip: the pinhole arithmetic of reference utils/projection_utils.py:5-23 for ONE point, evaluated
// in the float widths NumPy-2 promotion gives each stage (see the header of project.hip).
#pragma once
#include "pb3d_internal.h"

namespace pb3d_proj {

struct ProjParams {
    double R[9], cam[3], f, cx, cy;
    int t0, tm, tu, tv;
    int Himg, Wimg, pts_f64;
};

__device__ __forceinline__ double rnd(double v, int is64) { return is64 ? v : (double)(float)v; }

// mode 0: project_colored_voxels (Z < 1e-8 clamped to 1e-8); mode 1: the z-buffer functions of
// reference utils/eval_helpers_intra.py:134-190 (points with Z <= 1e-6 are dropped, no clamp)
template <int MODE>
__device__ __forceinline__ bool project_xyz(const ProjParams& P, const double p[3], int* ui, int* vi, double* zout = nullptr) {
    double pc[3];
    if (P.t0) {
        const double d0 = __dsub_rn(p[0], P.cam[0]), d1 = __dsub_rn(p[1], P.cam[1]), d2 = __dsub_rn(p[2], P.cam[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            pc[r] = __fma_rn(d2, P.R[3 * r + 2], __fma_rn(d1, P.R[3 * r + 1], __dmul_rn(d0, P.R[3 * r])));
    } else {
        const float d0 = __fsub_rn((float)p[0], (float)P.cam[0]), d1 = __fsub_rn((float)p[1], (float)P.cam[1]),
                    d2 = __fsub_rn((float)p[2], (float)P.cam[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            pc[r] = (double)__fmaf_rn(d2, (float)P.R[3 * r + 2],
                                      __fmaf_rn(d1, (float)P.R[3 * r + 1], __fmul_rn(d0, (float)P.R[3 * r])));
    }
    const double X = pc[0], Y = pc[1];
    double Z = pc[2];
    if (MODE == 0) {
        const double zmin = P.t0 ? 1e-8 : (double)(float)1e-8;
        if (Z < zmin) Z = zmin;
    } else {
        const double zthr = P.t0 ? 1e-6 : (double)(float)1e-6;
        if (!(Z > zthr)) return false;
        if (zout) *zout = Z;
    }
    const double qx = rnd(__ddiv_rn(X, Z), P.t0);
    const double qy = -rnd(__ddiv_rn(Y, Z), P.t0);
    const double fm = P.tm ? P.f : (double)(float)P.f;
    const double mu = rnd(__dmul_rn(qx, fm), P.tm), mv = rnd(__dmul_rn(qy, fm), P.tm);
    const double u = rnd(__dadd_rn(mu, P.tu ? P.cx : (double)(float)P.cx), P.tu);
    const double v = rnd(__dadd_rn(mv, P.tv ? P.cy : (double)(float)P.cy), P.tv);
    const double ur = rint(u), vr = rint(v);
    if (!(ur >= 0.0 && ur < (double)P.Wimg && vr >= 0.0 && vr < (double)P.Himg)) return false;  // NaN -> false
    *ui = (int)ur; *vi = (int)vr;
    return true;
}

template <int MODE>
__device__ __forceinline__ bool project_point(const ProjParams& P, const void* __restrict__ pts, i64 i, int* ui, int* vi,
                                              double* zout = nullptr) {
    double p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        p[k] = P.pts_f64 ? ((const double*)pts)[3 * i + k] : (double)((const float*)pts)[3 * i + k];
    return project_xyz<MODE>(P, p, ui, vi, zout);
}


// host: validate the promotion flags and fill the kernel parameter block
inline int fill_proj(ProjParams* P, int pts_f64, const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4],
                     int Himg, int Wimg) {
    for (int k = 0; k < 4; ++k) PB3D_REQUIRE(prec[k] == 0 || prec[k] == 1, "pb3d projection: prec[%d] must be 0 or 1", k);
    PB3D_REQUIRE(prec[1] >= prec[0] && prec[2] >= prec[1] && prec[3] >= prec[1], "pb3d projection: precision may only widen");
    memcpy(P->R, R, sizeof(P->R)); memcpy(P->cam, cam, sizeof(P->cam));
    P->f = f; P->cx = cx; P->cy = cy;
    P->t0 = prec[0]; P->tm = prec[1]; P->tu = prec[2]; P->tv = prec[3];
    P->Himg = Himg; P->Wimg = Wimg; P->pts_f64 = pts_f64 ? 1 : 0;
    return PB3D_OK;
}

}  // namespace pb3d_proj
