/*
 * pb3d_oracle.c -- CPU ORACLE for the semantic voxel-carving / re-projection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker.  The product
 * (part-based-3d-reconstruction_amd/) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of what the reference's Python computes, loop by loop:
 *   reference utils/voxel_carving_utils.py  (carve / rotate-carve / colour / part jobs)
 *   reference utils/voxel_utils.py          (grid -> points)
 *   reference utils/projection_utils.py     (pinhole scatter projection)
 *   reference utils/camera_estimation.py:770-787 (per-part IoU)
 * plus the arithmetic of a third-party dependency that is NOT under /root/reference:
 *   SciPy 1.15.3 scipy.ndimage.affine_transform(order=1, mode="constant", cval=0)
 *   (reference pins scipy==1.10.1 in requirements.txt; the container has 1.15.3) --
 *   restated from its published algorithm (NI_GeometricTransform + linear spline
 *   weights) and pinned against SciPy itself and against the imported reference by
 *   tools/pin_oracle.py and the golden vectors under tests/golden/ (see DESIGN.md).
 *
 * PARITY PIN: every function here is checked (tests/test_oracle_golden.py, CPU-only)
 * against golden input/output vectors captured by running the reference's own
 * functions in the build container (tools/gen_golden.py), and the 90-degree carve path
 * additionally against the reference's stored artefact
 * results/1.Orthographic_Voxel_Carving/Taj_voxel_grid.npz (digests in tests/golden/).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared  (oracle/Makefile).
 * -ffp-contract=off matters: SciPy's coordinate arithmetic is separate IEEE multiply
 * and add; an FMA would change border decisions.
 *
 * Grid axes are always (W=x, H=y, D=z[,C]) in C order, uint8.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint8_t u8;
typedef int64_t i64;

static int g_threads = 0; /* 0 = OpenMP default */

void orc_set_threads(int n) { g_threads = n; }

int orc_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

#ifdef _OPENMP
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(orc_get_threads())")
#else
#define ORC_PAR
#endif

/* ------------------------------------------------------------------------------------
 * A3: Rinv(angle) -- reference utils/voxel_carving_utils.py:65-69.
 * numpy.linalg.inv's LAPACK rounding is platform dependent, so the 91 matrices the
 * reference produces in the build container are pinned as bit patterns.
 * ---------------------------------------------------------------------------------- */
static const uint64_t k_rotinv_bits[91][9] = {
#include "rotinv_table.inc"
};

int orc_rotinv(int angle_deg, double M[9]) {
    if (angle_deg < 0 || angle_deg > 90) return -1;
    memcpy(M, k_rotinv_bits[angle_deg], 9 * sizeof(double));
    return 0;
}

/* offset = center - Rinv @ center, center = shape / 2  (voxel_carving_utils.py:108,119).
 * NumPy's `M @ c` goes through OpenBLAS dgemv whose kernel accumulates each row as an
 * FMA chain fma(M2,c2, fma(M1,c1, M0*c0)) (verified on 3952 shape x angle cases). */
void orc_offset(const double M[9], const i64 shape[3], double off[3]) {
    double c0 = (double)shape[0] / 2.0, c1 = (double)shape[1] / 2.0, c2 = (double)shape[2] / 2.0;
    double c[3] = {c0, c1, c2};
    for (int h = 0; h < 3; ++h) {
        double p = M[3 * h + 0] * c0;
        p = fma(M[3 * h + 1], c1, p);
        p = fma(M[3 * h + 2], c2, p);
        off[h] = c[h] - p;
    }
}

/* SciPy edge rule for an out-of-range tap index in mode="constant" (only reached with
 * weight exactly 0, at cc == n-1): mirror about the last sample. */
static inline i64 orc_mirror(i64 idx, i64 len) {
    if (len <= 1) return 0;
    i64 s2 = 2 * len - 2;
    if (idx < 0) {
        idx = s2 * (i64)(-idx / s2) + idx;
        return idx <= 1 - len ? idx + s2 : -idx;
    }
    if (idx >= len) {
        idx -= s2 * (i64)(idx / s2);
        if (idx >= len) idx = s2 - idx;
    }
    return idx;
}

/* ------------------------------------------------------------------------------------
 * scipy.ndimage.affine_transform(in, M, offset=off, order=1, mode="constant", cval=0)
 * on a uint8 volume -- call site voxel_carving_utils.py:116-123.
 *   cc[h] = (((0 + x*M[h][0]) + y*M[h][1]) + z*M[h][2]) + off[h]     (no FMA)
 *   outside [0, n-1] on any axis -> 0
 *   w0 = 1 - (cc - floor cc), w1 = 1 - w0
 *   acc += ((v * w[0][i]) * w[1][j]) * w[2][k]   taps in C order
 *   uint8 store: acc > 0 ? trunc(min(acc + 0.5, 255)) : 0
 * ---------------------------------------------------------------------------------- */
void orc_affine_u8(const u8* in, i64 W, i64 H, i64 D, const double M[9], const double off[3], u8* out) {
    const i64 n[3] = {W, H, D};
    ORC_PAR
    for (i64 x = 0; x < W; ++x) {
        for (i64 y = 0; y < H; ++y) {
            for (i64 z = 0; z < D; ++z) {
                const double o[3] = {(double)x, (double)y, (double)z};
                double w[3][2];
                i64 s[3];
                int outside = 0;
                for (int h = 0; h < 3; ++h) {
                    double cc = 0.0;
                    for (int l = 0; l < 3; ++l) cc += o[l] * M[3 * h + l];
                    cc += off[h];
                    if (cc < 0.0 || cc > (double)(n[h] - 1)) { outside = 1; break; }
                    double fl = floor(cc);
                    s[h] = (i64)fl;
                    double t = cc - fl;
                    w[h][0] = 1.0 - t;
                    w[h][1] = 1.0 - w[h][0];
                }
                u8 r = 0;
                if (!outside) {
                    double acc = 0.0;
                    for (int i = 0; i < 2; ++i) {
                        i64 ix = s[0] + i; if (ix >= W) ix = orc_mirror(ix, W);
                        for (int j = 0; j < 2; ++j) {
                            i64 iy = s[1] + j; if (iy >= H) iy = orc_mirror(iy, H);
                            for (int k = 0; k < 2; ++k) {
                                i64 iz = s[2] + k; if (iz >= D) iz = orc_mirror(iz, D);
                                double v = (double)in[(ix * H + iy) * D + iz];
                                v *= w[0][i]; v *= w[1][j]; v *= w[2][k];
                                acc += v;
                            }
                        }
                    }
                    if (acc > 0.0) {
                        acc += 0.5;
                        if (acc > 255.0) acc = 255.0;
                        r = (u8)acc;
                    }
                }
                out[(x * H + y) * D + z] = r;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A4: carve_voxel_grid_with_masks -- voxel_carving_utils.py:76-97, after _mask_to_wh.
 * mask_wh is (W,H) [mask_c==1: truthiness gates the whole (z,c) column, :83-87]
 * or (W,H,3) [mask_c==3: channel c gated by mask[...,c], C must be 3, :90-95].
 * ---------------------------------------------------------------------------------- */
int orc_carve_mask(const u8* grid, i64 W, i64 H, i64 D, int C, const u8* mask_wh, int mask_c, u8* out) {
    if (mask_c != 1 && mask_c != 3) return -1;
    if (mask_c == 3 && C != 3) return -2;
    const i64 col = D * C;
    ORC_PAR
    for (i64 xy = 0; xy < W * H; ++xy) {
        const u8* g = grid + xy * col;
        u8* o = out + xy * col;
        if (mask_c == 1) {
            if (mask_wh[xy]) memcpy(o, g, (size_t)col); else memset(o, 0, (size_t)col);
        } else {
            const u8* m = mask_wh + xy * 3;
            for (i64 z = 0; z < D; ++z)
                for (int c = 0; c < 3; ++c) o[z * 3 + c] = m[c] ? g[z * 3 + c] : 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A5: process_voxel_grid -- voxel_carving_utils.py:104-126.  Cumulative: each step
 * rotates the already rotated-and-carved grid by `angle` about Y; never rotated back.
 * ---------------------------------------------------------------------------------- */
int orc_process_grid(const u8* occ, i64 W, i64 H, i64 D, const u8* mask_wh, int angle_interval, u8* out) {
    if (angle_interval <= 0) return -1; /* range(0, 91, k) with k <= 0: empty or ValueError upstream */
    const i64 shape[3] = {W, H, D};
    const size_t nbytes = (size_t)(W * H * D);
    u8* cur = (u8*)malloc(nbytes ? nbytes : 1);
    u8* rot = (u8*)malloc(nbytes ? nbytes : 1);
    if (!cur || !rot) { free(cur); free(rot); return -2; }
    memcpy(cur, occ, nbytes);
    for (int angle = 0; angle < 91; angle += angle_interval) {
        double M[9], off[3];
        orc_rotinv(angle, M);
        orc_offset(M, shape, off);
        orc_affine_u8(cur, W, H, D, M, off, rot);
        orc_carve_mask(rot, W, H, D, 1, mask_wh, 1, cur);
    }
    memcpy(out, cur, nbytes);
    free(cur); free(rot);
    return 0;
}

/* A2: _occupancy -- voxel_carving_utils.py:32-33: any(grid > 0, axis=-1) as uint8. */
void orc_occupancy(const u8* grid4, i64 nvox, u8* out) {
    ORC_PAR
    for (i64 i = 0; i < nvox; ++i) out[i] = (grid4[3 * i] | grid4[3 * i + 1] | grid4[3 * i + 2]) ? 1 : 0;
}

/* A6: apply_colored_mask_to_voxel_grid -- voxel_carving_utils.py:128-136.
 * out[x,y,z,:] = rgb[y,x,:] where carved[x,y,z] == 1 (exactly 1), else 0.  rgb is (H,W,3). */
void orc_color_apply(const u8* carved, i64 W, i64 H, i64 D, const u8* rgb_hw3, u8* out) {
    ORC_PAR
    for (i64 x = 0; x < W; ++x)
        for (i64 y = 0; y < H; ++y) {
            const u8* px = rgb_hw3 + (y * W + x) * 3;
            const u8* cv = carved + (x * H + y) * D;
            u8* o = out + (x * H + y) * D * 3;
            for (i64 z = 0; z < D; ++z) {
                int on = cv[z] == 1;
                o[3 * z + 0] = on ? px[0] : 0;
                o[3 * z + 1] = on ? px[1] : 0;
                o[3 * z + 2] = on ? px[2] : 0;
            }
        }
}

/* A7: global_carve -- voxel_carving_utils.py:269-298.  ones((w,h,w)) -> A5 -> A6.
 * bin_hw is the (h,w) binary mask; since the grid is (w,h,w), _mask_to_wh transposes it
 * (also when h == w, where the (H,W) test wins). */
int orc_global_carve(const u8* bin_hw, const u8* rgb_hw3, i64 h, i64 w, int angle_interval, u8* out) {
    const i64 W = w, H = h, D = w;
    const size_t n = (size_t)(W * H * D);
    u8* ones = (u8*)malloc(n ? n : 1);
    u8* carved = (u8*)malloc(n ? n : 1);
    const size_t npx = (size_t)(W * H);
    u8* m_wh = (u8*)malloc(npx ? npx : 1);
    if (!ones || !carved || !m_wh) { free(ones); free(carved); free(m_wh); return -2; }
    memset(ones, 1, n);
    for (i64 x = 0; x < W; ++x)
        for (i64 y = 0; y < H; ++y) m_wh[x * H + y] = bin_hw[y * W + x] ? 1 : 0;
    int rc = orc_process_grid(ones, W, H, D, m_wh, angle_interval, carved);
    if (rc == 0) orc_color_apply(carved, W, H, D, rgb_hw3, out);
    free(ones); free(carved); free(m_wh);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * A8: part_carve -- voxel_carving_utils.py:139-160.
 * Per job j (already reduced by the caller to its 2-D pixel masks, both (W,H) uint8 0/1):
 *   mask_sub   = mask2d.T                          (:151, gates `sub`)
 *   mask_carve = _mask_to_wh(mask2d.T, W, H)       (what process_voxel_grid really uses;
 *                differs from mask_sub only when W == H, where :24 transposes it again)
 *   sub  = colored * mask_sub ; occ = any(sub > 0) ; carved = A5(occ, mask_carve, angle)
 *   part = sub * carved ; final[any(part > 0)] = part[...]       (later jobs overwrite)
 * Jobs whose mask2d is all-false are skipped (:148-149): pass job_skip[j] != 0.
 * ---------------------------------------------------------------------------------- */
int orc_part_carve(const u8* colored, i64 W, i64 H, i64 D, const u8* mask_sub, const u8* mask_carve,
                   const int* job_angle, const int* job_skip, int nj, u8* out) {
    const i64 nvox = W * H * D;
    u8* occ = (u8*)malloc((size_t)nvox ? (size_t)nvox : 1);
    u8* carved = (u8*)malloc((size_t)nvox ? (size_t)nvox : 1);
    if (!occ || !carved) { free(occ); free(carved); return -2; }
    memset(out, 0, (size_t)nvox * 3);
    int rc = 0;
    for (int j = 0; j < nj && rc == 0; ++j) {
        if (job_skip[j]) continue;
        const u8* ms = mask_sub + (i64)j * W * H;
        const u8* mc = mask_carve + (i64)j * W * H;
        ORC_PAR
        for (i64 xy = 0; xy < W * H; ++xy)
            for (i64 z = 0; z < D; ++z) {
                const u8* p = colored + (xy * D + z) * 3;
                occ[xy * D + z] = (ms[xy] && (p[0] | p[1] | p[2])) ? 1 : 0;
            }
        rc = orc_process_grid(occ, W, H, D, mc, job_angle[j], carved);
        if (rc) break;
        ORC_PAR
        for (i64 xy = 0; xy < W * H; ++xy)
            for (i64 z = 0; z < D; ++z) {
                const i64 v = xy * D + z;
                /* part = (colored * m) * carved, uint8 wrap-around product as NumPy computes it */
                u8 p0 = (u8)(colored[3 * v + 0] * ms[xy]), p1 = (u8)(colored[3 * v + 1] * ms[xy]),
                   p2 = (u8)(colored[3 * v + 2] * ms[xy]);
                p0 = (u8)(p0 * carved[v]); p1 = (u8)(p1 * carved[v]); p2 = (u8)(p2 * carved[v]);
                if (p0 | p1 | p2) { out[3 * v] = p0; out[3 * v + 1] = p1; out[3 * v + 2] = p2; }
            }
    }
    free(occ); free(carved);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * A15/A16: grid -> points.  voxel_utils.py:7-21 (colour-select) and :35-51 (occupancy,
 * strided).  Selection over a (A0,A1,A2[,C]) grid; points are emitted in np.where order
 * (a0 major, a2 minor) as float32 (a2, a1, a0) * stride, colours as the voxel's C bytes.
 *   ncolors > 0 : voxel selected iff its RGB equals any of colors[ncolors][3]  (C == 3)
 *   ncolors == 0: voxel selected iff any channel non-zero (C == 3) / value non-zero (C == 1)
 * Only indices that are multiples of `stride` on every axis are visited ([::s,::s,::s]).
 * pts == NULL -> count only.  Returns the number of selected voxels.
 * ---------------------------------------------------------------------------------- */
i64 orc_points(const u8* grid, i64 A0, i64 A1, i64 A2, int C, const u8* colors, int ncolors, int stride,
               float* pts, u8* cols) {
    i64 n = 0;
    for (i64 a0 = 0; a0 < A0; a0 += stride)
        for (i64 a1 = 0; a1 < A1; a1 += stride)
            for (i64 a2 = 0; a2 < A2; a2 += stride) {
                const u8* p = grid + ((a0 * A1 + a1) * A2 + a2) * C;
                int sel = 0;
                if (ncolors > 0) {
                    for (int k = 0; k < ncolors && !sel; ++k)
                        sel = p[0] == colors[3 * k] && p[1] == colors[3 * k + 1] && p[2] == colors[3 * k + 2];
                } else {
                    for (int c = 0; c < C; ++c) sel |= p[c] != 0;
                }
                if (!sel) continue;
                if (pts) {
                    /* index // stride as float32, then * stride in float32 (voxel_utils.py:42) */
                    pts[3 * n + 0] = (float)(a2 / stride) * (float)stride;
                    pts[3 * n + 1] = (float)(a1 / stride) * (float)stride;
                    pts[3 * n + 2] = (float)(a0 / stride) * (float)stride;
                    if (cols) for (int c = 0; c < C; ++c) cols[C * n + c] = p[c];
                }
                ++n;
            }
    return n;
}

/* ------------------------------------------------------------------------------------
 * A14: project_colored_voxels -- projection_utils.py:5-23, given R = look_at_rotation.
 *   pc = (p - cam) @ R.T   each component an FMA chain fma(d2,r2, fma(d1,r1, d0*r0))
 *        (what NumPy's gemm does for realistic N, both dtypes)
 *   Z < 1e-8 -> 1e-8 ; u = (X/Z)*f + cx ; v = -(Y/Z)*f + cy  (separate mul / add)
 *   ui, vi = rint (half to even) ; in-bounds points written in input order, so for
 *   duplicate pixels the LAST point in input order wins.
 * prec[4] = {T0, Tmul, Taddu, Taddv}: 0 = float32, 1 = float64 arithmetic for the matmul/
 * divide stage, the multiply by f, and the two adds (NumPy-2 promotion is decided by the
 * caller from the Python types of f, cx, cy).
 * ---------------------------------------------------------------------------------- */
static inline double orc_rnd(double v, int is64) { return is64 ? v : (double)(float)v; }

void orc_project(const void* pts, int pts_f64, const u8* cols, i64 n, const double R[9], const double cam[3],
                 double f, double cx, double cy, const int prec[4], int Himg, int Wimg, u8* img) {
    memset(img, 0, (size_t)Himg * Wimg * 3);
    const int t0 = prec[0], tm = prec[1], tu = prec[2], tv = prec[3];
    for (i64 i = 0; i < n; ++i) {
        double p[3];
        for (int k = 0; k < 3; ++k)
            p[k] = pts_f64 ? ((const double*)pts)[3 * i + k] : (double)((const float*)pts)[3 * i + k];
        double d[3], pc[3];
        if (t0) {
            for (int k = 0; k < 3; ++k) d[k] = p[k] - cam[k];
            for (int r = 0; r < 3; ++r)
                pc[r] = fma(d[2], R[3 * r + 2], fma(d[1], R[3 * r + 1], d[0] * R[3 * r + 0]));
        } else {
            float df[3];
            for (int k = 0; k < 3; ++k) df[k] = (float)p[k] - (float)cam[k];
            for (int r = 0; r < 3; ++r)
                pc[r] = (double)fmaf(df[2], (float)R[3 * r + 2],
                                     fmaf(df[1], (float)R[3 * r + 1], df[0] * (float)R[3 * r + 0]));
        }
        double X = pc[0], Y = pc[1], Z = pc[2];
        const double zmin = t0 ? 1e-8 : (double)(float)1e-8;
        if (Z < zmin) Z = zmin;
        /* +,-,*,/ of two float32 values evaluated in double and rounded once to float32 are
         * exactly the float32 operation (53 >= 2*24+2), so orc_rnd() reproduces each stage. */
        double qx = orc_rnd(X / Z, t0), qy = orc_rnd(-(orc_rnd(Y / Z, t0)), t0);
        double fm = tm ? f : (double)(float)f;
        double mu = orc_rnd(qx * fm, tm), mv = orc_rnd(qy * fm, tm);
        double u = orc_rnd(mu + (tu ? cx : (double)(float)cx), tu);
        double v = orc_rnd(mv + (tv ? cy : (double)(float)cy), tv);
        double ur = nearbyint(u), vr = nearbyint(v);
        if (!(ur >= 0 && ur < (double)Wimg && vr >= 0 && vr < (double)Himg)) continue;
        i64 ui = (i64)ur, vi = (i64)vr;
        u8* o = img + (vi * Wimg + ui) * 3;
        o[0] = cols[3 * i]; o[1] = cols[3 * i + 1]; o[2] = cols[3 * i + 2];
    }
}

/* A17: compute_partwise_iou -- camera_estimation.py:770-787.  Per palette colour:
 * inter = #(a == colour & b == colour), union = #(a == colour | b == colour). */
void orc_partwise_iou(const u8* a, const u8* b, i64 npix, const u8* colors, int ncolors, i64* inter, i64* uni) {
    for (int k = 0; k < ncolors; ++k) {
        i64 in = 0, un = 0;
        const u8 c0 = colors[3 * k], c1 = colors[3 * k + 1], c2 = colors[3 * k + 2];
        for (i64 i = 0; i < npix; ++i) {
            int ma = a[3 * i] == c0 && a[3 * i + 1] == c1 && a[3 * i + 2] == c2;
            int mb = b[3 * i] == c0 && b[3 * i + 1] == c1 && b[3 * i + 2] == c2;
            in += ma & mb; un += ma | mb;
        }
        inter[k] = in; uni[k] = un;
    }
}

/* ------------------------------------------------------------------------------------
 * A18: deform_coords -- reference utils/deformation_estimation.py:70-98 (closure of the
 * notebook-3 viewer).  Seven jitters (0, +-0.25 per axis) of the float32 points, each
 * centred on its own mean, scaled / shifted per axis in float64, rounded half-to-even to
 * integers, then np.unique(axis=0): the lexicographically sorted set of (x,y,z) rows.
 *   kx = shift_xz * pix2vox_x, ky = shift_y * pix2vox_y, kz = shift_xz * pix2vox_z are formed by the
 *   caller as Python floats exactly as upstream (:76-81).
 * Returns the number of unique rows; out (capacity 7n rows of int64[3]) receives them.
 * ---------------------------------------------------------------------------------- */
static int orc_cmp3(const void* a, const void* b) {
    const i64* p = (const i64*)a; const i64* q = (const i64*)b;
    for (int k = 0; k < 3; ++k) { if (p[k] < q[k]) return -1; if (p[k] > q[k]) return 1; }
    return 0;
}

static inline double orc_sign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : v); }

i64 orc_deform_coords(const float* pts, i64 n, double sxz, double sy, double kx, double ky, double kz, i64* out) {
    static const double offs[7][3] = {{0, 0, 0}, {0.25, 0, 0}, {-0.25, 0, 0}, {0, 0.25, 0}, {0, -0.25, 0}, {0, 0, 0.25}, {0, 0, -0.25}};
    if (n == 0) return 0;
    for (int j = 0; j < 7; ++j) {
        double sum[3] = {0, 0, 0};
        for (i64 i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) sum[a] += (double)pts[3 * i + a] + offs[j][a];   /* np.mean: rows added in order */
        double ctr[3] = {sum[0] / (double)n, sum[1] / (double)n, sum[2] / (double)n};
        for (i64 i = 0; i < n; ++i) {
            double c0 = ((double)pts[3 * i + 0] + offs[j][0]) - ctr[0];
            double c1 = ((double)pts[3 * i + 1] + offs[j][1]) - ctr[1];
            double c2 = ((double)pts[3 * i + 2] + offs[j][2]) - ctr[2];
            double d0 = c0 * sxz + kx * orc_sign(c0);
            double d1 = c1 * sy - ky;
            double d2 = c2 * sxz + kz * orc_sign(c2);
            i64* o = out + 3 * (j * n + i);
            o[0] = (i64)nearbyint(d0 + ctr[0]); o[1] = (i64)nearbyint(d1 + ctr[1]); o[2] = (i64)nearbyint(d2 + ctr[2]);
        }
    }
    qsort(out, (size_t)(7 * n), 3 * sizeof(i64), orc_cmp3);
    i64 m = 0;
    for (i64 i = 0; i < 7 * n; ++i)
        if (m == 0 || orc_cmp3(out + 3 * i, out + 3 * (m - 1)) != 0) { if (m != i) memcpy(out + 3 * m, out + 3 * i, 3 * sizeof(i64)); ++m; }
    return m;
}

/* ------------------------------------------------------------------------------------
 * scipy.ndimage.label(mask) with the default 6-connected structure (call sites
 * voxel_carving_utils.py:175, :254): labels 1..n in raster order of each component's first
 * voxel.  Two-pass union-find with "smaller index is the root".
 * ---------------------------------------------------------------------------------- */
static i64 orc_uf_find(i64* p, i64 v) { while (p[v] != v) { p[v] = p[p[v]]; v = p[v]; } return v; }

i64 orc_label6(const u8* mask, i64 A0, i64 A1, i64 A2, int32_t* labels) {
    const i64 n = A0 * A1 * A2;
    i64* p = (i64*)malloc((size_t)(n ? n : 1) * sizeof(i64));
    if (!p) return -1;
    for (i64 v = 0; v < n; ++v) p[v] = v;
    for (i64 v = 0; v < n; ++v) {
        if (!mask[v]) continue;
        const i64 a2 = v % A2, a1 = (v / A2) % A1;
        const i64 nb[3] = {a2 > 0 ? v - 1 : -1, a1 > 0 ? v - A2 : -1, v >= A1 * A2 ? v - A1 * A2 : -1};
        for (int k = 0; k < 3; ++k) {
            if (nb[k] < 0 || !mask[nb[k]]) continue;
            i64 ra = orc_uf_find(p, v), rb = orc_uf_find(p, nb[k]);
            if (ra == rb) continue;
            if (ra < rb) p[rb] = ra; else p[ra] = rb;
        }
    }
    i64 next = 0;
    for (i64 v = 0; v < n; ++v) {
        if (!mask[v]) { labels[v] = 0; continue; }
        const i64 r = orc_uf_find(p, v);
        if (r == v) labels[v] = (int32_t)(++next);      /* roots are met first in raster order */
        else labels[v] = labels[r];
    }
    free(p);
    return next;
}

/* ------------------------------------------------------------------------------------
 * N5: z-buffer visibility -- reference utils/eval_helpers_intra.py:134-163 (compute_global_depth_buffer)
 * and :168-190 (project_part_visible).  Same pinhole arithmetic as orc_project, except that points
 * with Z <= 1e-6 are dropped instead of clamped.  Sequential loops exactly as upstream.
 * ---------------------------------------------------------------------------------- */
static int orc_pin(const void* pts, int pts_f64, i64 i, const double R[9], const double cam[3], double f, double cx, double cy,
                   const int prec[4], int Himg, int Wimg, i64* ui, i64* vi, double* zout) {
    const int t0 = prec[0], tm = prec[1], tu = prec[2], tv = prec[3];
    double p[3];
    for (int k = 0; k < 3; ++k)
        p[k] = pts_f64 ? ((const double*)pts)[3 * i + k] : (double)((const float*)pts)[3 * i + k];
    double pc[3];
    if (t0) {
        double d[3];
        for (int k = 0; k < 3; ++k) d[k] = p[k] - cam[k];
        for (int r = 0; r < 3; ++r) pc[r] = fma(d[2], R[3 * r + 2], fma(d[1], R[3 * r + 1], d[0] * R[3 * r + 0]));
    } else {
        float df[3];
        for (int k = 0; k < 3; ++k) df[k] = (float)p[k] - (float)cam[k];
        for (int r = 0; r < 3; ++r)
            pc[r] = (double)fmaf(df[2], (float)R[3 * r + 2], fmaf(df[1], (float)R[3 * r + 1], df[0] * (float)R[3 * r + 0]));
    }
    const double X = pc[0], Y = pc[1], Z = pc[2];
    if (!(Z > (t0 ? 1e-6 : (double)(float)1e-6))) return 0;
    double qx = orc_rnd(X / Z, t0), qy = -orc_rnd(Y / Z, t0);
    double fm = tm ? f : (double)(float)f;
    double mu = orc_rnd(qx * fm, tm), mv = orc_rnd(qy * fm, tm);
    double u = orc_rnd(mu + (tu ? cx : (double)(float)cx), tu);
    double v = orc_rnd(mv + (tv ? cy : (double)(float)cy), tv);
    double ur = nearbyint(u), vr = nearbyint(v);
    if (!(ur >= 0 && ur < (double)Wimg && vr >= 0 && vr < (double)Himg)) return 0;
    *ui = (i64)ur; *vi = (i64)vr; *zout = Z;
    return 1;
}

void orc_depth_buffer(const void* pts, int pts_f64, i64 n, const double R[9], const double cam[3], double f, double cx, double cy,
                      const int prec[4], int Himg, int Wimg, float* zbuf) {
    for (i64 k = 0; k < (i64)Himg * Wimg; ++k) zbuf[k] = INFINITY;
    for (i64 i = 0; i < n; ++i) {
        i64 ui, vi; double z;
        if (!orc_pin(pts, pts_f64, i, R, cam, f, cx, cy, prec, Himg, Wimg, &ui, &vi, &z)) continue;
        if (z < (double)zbuf[vi * Wimg + ui]) zbuf[vi * Wimg + ui] = (float)z;
    }
}

void orc_visible_mask(const void* pts, int pts_f64, i64 n, const double R[9], const double cam[3], double f, double cx, double cy,
                      const int prec[4], const float* zbuf, int Himg, int Wimg, double eps, int eps_f32, u8* mask) {
    memset(mask, 0, (size_t)Himg * Wimg);
    for (i64 i = 0; i < n; ++i) {
        i64 ui, vi; double z;
        if (!orc_pin(pts, pts_f64, i, R, cam, f, cx, cy, prec, Himg, Wimg, &ui, &vi, &z)) continue;
        const float zb = zbuf[vi * Wimg + ui];
        int hit;
        if (prec[0]) hit = fabs(z - (double)zb) < eps;
        else { const float dz = fabsf((float)z - zb); hit = eps_f32 ? dz < (float)eps : (double)dz < eps; }
        if (hit) mask[vi * Wimg + ui] = 1;
    }
}
