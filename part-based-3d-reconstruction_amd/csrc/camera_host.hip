// Host half of the batched camera objective (row N4): K look-at rotations per call.
//
// reference utils/camera_geometry.py:3-14 (look_at_rotation) is ten NumPy calls on 3-vectors -- 30-40 us of interpreter time
// per camera, more than the whole GPU side of a batched evaluation.  This file restates it operation by operation in the
// caller's float width (every NumPy step here is an element-wise IEEE operation except the 3-element dot product inside
// numpy.linalg.norm, whose rounding depends on the BLAS kernel of the host: `dot_mode` selects it and the Python shim finds
// the mode that reproduces this host's NumPy bit for bit, or stays on NumPy when none does).  -ffp-contract=off: nothing
// here may be fused except the explicit fma() calls.
#include <cmath>

#include "pb3d_internal.h"

namespace {

template <class T>
inline T dot3(const T x[3], int mode) {
    if (mode == 1) return std::fma(x[2], x[2], std::fma(x[1], x[1], x[0] * x[0]));
    if (mode == 3) return std::fma(x[0], x[0], std::fma(x[1], x[1], x[2] * x[2]));
    if (mode == 4) return (T)(((double)(T)(x[0] * x[0]) + (double)(T)(x[1] * x[1])) + (double)(T)(x[2] * x[2]));   // OpenBLAS sdot tail loop
    if (mode == 2) return (T)(((double)x[0] * (double)x[0] + (double)x[1] * (double)x[1]) + (double)x[2] * (double)x[2]);
    return (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2];
}

// numpy.cross for two 3-vectors: each component is (rounded product) - (rounded product)
template <class T>
inline void cross3(const T a[3], const T b[3], T c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

template <class T>
void look_at(const T* eye, const T* target, int mode, double* R) {
    T f[3] = {(T)(target[0] - eye[0]), (T)(target[1] - eye[1]), (T)(target[2] - eye[2])};
    const T nf = std::sqrt(dot3(f, mode));
    for (int k = 0; k < 3; ++k) f[k] = f[k] / nf;
    // np.dot(forward, [0,1,0]) == forward[1] whatever the summation order.  np.allclose(|d|, 1.0): |d| is a NumPy scalar of
    // width T, 1.0 and the tolerance (atol + rtol * |1.0|, Python floats) are weak scalars, so the whole test runs in T
    const T d = std::fabs(f[1]);
    const T thr = (T)(1e-8 + 1e-5 * std::fabs(1.0));
    T up[3] = {(T)0, (T)1, (T)0};
    if (std::fabs(d - (T)1) <= thr || d == (T)1) { up[1] = (T)0; up[2] = (T)1; }
    T r[3], u[3];
    cross3(up, f, r);
    const T nr = std::sqrt(dot3(r, mode));
    for (int k = 0; k < 3; ++k) r[k] = r[k] / nr;
    cross3(f, r, u);
    for (int k = 0; k < 3; ++k) { R[k] = (double)r[k]; R[3 + k] = (double)u[k]; R[6 + k] = (double)f[k]; }
}

}  // namespace

extern "C" int pb3d_look_at_batch(const void* eye, const void* target, int is_f64, int64_t count, int dot_mode, double* R9) {
    PB3D_REQUIRE(count >= 0 && (count == 0 || (eye && target && R9)), "pb3d_look_at_batch: null argument");
    PB3D_REQUIRE(dot_mode >= 0 && dot_mode <= 4 && (is_f64 == 0 || (dot_mode != 2 && dot_mode != 4)), "pb3d_look_at_batch: bad dot_mode");
    for (int64_t k = 0; k < count; ++k) {
        if (is_f64) look_at<double>((const double*)eye + 3 * k, (const double*)target + 3 * k, dot_mode, R9 + 9 * k);
        else look_at<float>((const float*)eye + 3 * k, (const float*)target + 3 * k, dot_mode, R9 + 9 * k);
    }
    return PB3D_OK;
}
