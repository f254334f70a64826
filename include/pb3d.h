/*
 * pb3d.h -- C ABI of libpb3d.so: MI355X (gfx950) semantic voxel carving & re-projection.
 *
 * The reference (BarnitaSharma/Part-based-3D-Reconstruction) has no FFI: its boundary for
 * this path is the Python function surface of utils/voxel_carving_utils.py,
 * utils/voxel_utils.py, utils/projection_utils.py and utils/camera_estimation.py, called
 * with NumPy arrays.  This header is what a ctypes binding of those functions binds; each
 * entry point cites the reference function (file:line) it replaces.  INTEGRATION.md shows
 * the reference-side stub.
 *
 * Conventions
 *   - plain C, no C++/torch types; every buffer is caller-owned and C-contiguous;
 *   - grids are uint8, axes (W=x, H=y, D=z[,3]) in C order, exactly the reference's layout;
 *   - every function returns 0 on success or a negative PB3D_E* code; pb3d_last_error()
 *     gives the thread-local message of the last failure;
 *   - "*_dev" entry points take DEVICE pointers, enqueue on the context's HIP stream and
 *     return without synchronising (call pb3d_sync); the un-suffixed entry points take
 *     HOST pointers, stage through device scratch and return when the result is in the
 *     caller's buffer (these are what the NumPy shim calls);
 *   - a context belongs to one GPU and one HIP stream; use it from one thread at a time.
 *   - There is NO CPU fallback: every op fails with PB3D_ENODEVICE when no GPU is present.
 */
#ifndef PB3D_H
#define PB3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pb3d_ctx pb3d_ctx;
typedef struct pb3d_event pb3d_event;

enum {
    PB3D_OK = 0,
    PB3D_EINVAL = -1,       /* bad argument (shape, null pointer, angle ...)           */
    PB3D_ENODEVICE = -2,    /* no HIP device / HIP runtime error                        */
    PB3D_ENOMEM = -3,       /* device allocation failed                                  */
    PB3D_EUNSUPPORTED = -4, /* argument combination outside what the path needs          */
    PB3D_ECOMM = -5         /* RCCL not available / collective failed                    */
};

/* ---- lifecycle, device memory, timing (plumbing) ------------------------------------- */
int pb3d_version(void);
const char* pb3d_last_error(void);
int pb3d_device_count(int* n);
int pb3d_create(int device, pb3d_ctx** out);
void pb3d_destroy(pb3d_ctx* ctx);
/* A context belongs to ONE device; its stream, scratch and pools live there.  HIP's current device is a property of the host THREAD:
 * pb3d_create makes `device` current for the calling thread, so "one thread (or one process) per GPU, each with its own context" needs
 * nothing else.  A single thread that alternates between contexts of different devices calls pb3d_make_current(ctx) before each run of
 * calls on that context (it is hipSetDevice(ctx's device); the entry points do not switch devices themselves). */
int pb3d_make_current(pb3d_ctx* ctx);
int pb3d_device_info(pb3d_ctx* ctx, char* name, int name_cap, int* compute_units, int64_t* hbm_bytes);
int pb3d_sync(pb3d_ctx* ctx);
/* Development knobs (results never depend on them; they select between kernels that are all bit-exact, and the parity tests use them to
 * run both forms of a kernel on the same grids).  Every knob has a name -- "sliced", "rot90_wide", "rot90_flat", "rot90_fill",
 * "rot90_mask_block", "global_composed", "per_job", "points_fill", "points_onepass", "orient_tile", "ccl_*", "s32_*", "no_table_cache",
 * "uncap", "crop_ablate" (csrc/ctx.hip lists them with their ranges); an unknown name or a value out of range is PB3D_EINVAL.  Initial
 * values come from the environment, read ONCE in pb3d_create: PB3D_KNOBS="name=value,name=value" (any knob), and PB3D_SLICED,
 * PB3D_ROT90_WIDE, PB3D_S32_GPW, PB3D_UNCAP. */
int pb3d_set_tuning(pb3d_ctx* ctx, const char* name, int value);
int pb3d_dev_alloc(pb3d_ctx* ctx, size_t bytes, void** dptr);
int pb3d_dev_free(pb3d_ctx* ctx, void* dptr);
int pb3d_dev_memset(pb3d_ctx* ctx, void* dptr, int value, size_t bytes);
int pb3d_h2d(pb3d_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
int pb3d_d2h(pb3d_ctx* ctx, void* hptr, const void* dptr, size_t bytes);
/* Host -> device without waiting: the bytes are copied into a pinned ring of the context first, so hptr may be reused when the call
 * returns and the transfer is ordered on the context's stream like a kernel (inputs above 4 MiB take the blocking path of pb3d_h2d).
 * What a resident pipeline uploads between its stages are 2-D masks and descriptors of a few hundred KB. */
int pb3d_h2d_async(pb3d_ctx* ctx, void* dptr, const void* hptr, size_t bytes);
/* number of times the host has waited for the context's stream so far (tests bound the waits of a resident pipeline) */
int64_t pb3d_sync_count(pb3d_ctx* ctx);
int pb3d_d2d(pb3d_ctx* ctx, void* dst, const void* src, size_t bytes);
/* HIP events recorded on the context's stream (the stream every kernel is launched on). */
int pb3d_event_create(pb3d_ctx* ctx, pb3d_event** ev);
int pb3d_event_record(pb3d_ctx* ctx, pb3d_event* ev);
int pb3d_event_elapsed_ms(pb3d_ctx* ctx, pb3d_event* start, pb3d_event* stop, float* ms); /* syncs on stop */
void pb3d_event_destroy(pb3d_event* ev);

/* ---- host shim H1: pinned rotation data ------------------------------------------------
 * Rinv(angle) = numpy.linalg.inv of the Y-rotation, reference utils/voxel_carving_utils.py:65-69
 * (bit patterns pinned for angle = 0..90, see csrc/rotinv_table.inc), and
 * offset = c - Rinv@c with c = shape/2 as NumPy evaluates it (:108,:119; FMA chain). */
int pb3d_rotinv(int angle_deg, double M[9]);
int pb3d_offset(const double M[9], const int64_t shape[3], double off[3]);

/* ---- carve_voxel_grid_with_masks, reference utils/voxel_carving_utils.py:76-87 ----------
 * out = grid where mask_wh[x,y] != 0 else 0, broadcast over z (and channel).  C = 1 or 3.
 * mask_wh is the (W,H) uint8 truthiness image, i.e. after _mask_to_wh (:19-28), which is
 * host logic in the shim.  (The reference's RGB-mask branch :90-95 cannot broadcast for
 * any non-degenerate shape and always raises; the shim raises the same ValueError.) */
int pb3d_carve_mask_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t W, int64_t H, int64_t D, int C,
                        const uint8_t* d_mask_wh, uint8_t* d_out);
int pb3d_carve_mask(pb3d_ctx* ctx, const uint8_t* grid, int64_t W, int64_t H, int64_t D, int C,
                    const uint8_t* mask_wh, uint8_t* out);

/* ---- one rotate-about-Y + carve step = scipy.ndimage.affine_transform(order=1,
 * mode="constant", cval=0) followed by the mask carve; call site
 * reference utils/voxel_carving_utils.py:116-124.  M must have row 1 == [+-0, 1, +-0]
 * and off[1] == 0 (true for every Rinv).  d_mask_wh may be NULL (no carve). */
int pb3d_rotate_carve_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const double M[9], const double off[3], const uint8_t* d_mask_wh, uint8_t* d_out);
int pb3d_rotate_carve(pb3d_ctx* ctx, const uint8_t* occ, int64_t W, int64_t H, int64_t D,
                      const double M[9], const double off[3], const uint8_t* mask_wh, uint8_t* out);

/* ---- process_voxel_grid, reference utils/voxel_carving_utils.py:104-126 ------------------
 * for angle in range(0, 91, angle_interval): rotate-carve; cumulative.  The loop runs on
 * the device.  d_tmp: W*H*D bytes of scratch (ping-pong); d_out may not alias d_occ.
 * Chains of two and more rotation steps (angle_interval <= 45) keep the volume BIT-SLICED between the steps (32 planes per dword,
 * 1/4 B/voxel per middle step instead of 2; csrc/sliced.hip): the first kernel slices and checks that the data is 0 / 1; the whole chain
 * is queued behind it and the call then waits on the host for that ONE kernel's verdict (not for the chain): grids with other values
 * take the byte chain, which overwrites what the queued steps left in d_out -- same results.  Everything else about the call is
 * asynchronous on the context's stream, as for the other *_dev entries. */
int pb3d_process_grid_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const uint8_t* d_mask_wh, int angle_interval, uint8_t* d_out, uint8_t* d_tmp);
int pb3d_process_grid(pb3d_ctx* ctx, const uint8_t* occ, int64_t W, int64_t H, int64_t D,
                      const uint8_t* mask_wh, int angle_interval, uint8_t* out);

/* The same loop for grids that are NOT uint8.  The reference never looks at the dtype: scipy.ndimage.affine_transform(order=1) returns the
 * dtype it was given and np.where(mask, grid, 0) keeps it (float16 and anything else SciPy's interpolation refuses: "data type not
 * supported"; a bool grid becomes int64 in upstream's first np.where, so the host side passes it as PB3D_I64).  A plain
 * kernel per step (csrc/rotate_typed.hip: f64 accumulation in SciPy's tap order, SciPy's store rule of the type, the step's carve in the
 * same store); the notebooks only pass uint8, which keeps its own kernels above.  d_grid / d_out / d_tmp: W*H*D elements of the dtype
 * (complex: interleaved parts), none aliased.  64-bit integers go through a double exactly as in SciPy. */
enum {
    PB3D_I8 = 1, PB3D_U8 = 2, PB3D_I16 = 3, PB3D_U16 = 4, PB3D_I32 = 5, PB3D_U32 = 6, PB3D_I64 = 7, PB3D_U64 = 8,
    PB3D_F32 = 9, PB3D_F64 = 10, PB3D_C64 = 11, PB3D_C128 = 12
};
size_t pb3d_dtype_bytes(int dtype);      /* bytes per element, 0 for an unknown code */
int pb3d_process_grid_typed_dev(pb3d_ctx* ctx, const void* d_grid, int dtype, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                                int angle_interval, void* d_out, void* d_tmp);

/* ---- _occupancy, reference utils/voxel_carving_utils.py:32-33: any(grid > 0, axis=-1) ---- */
int pb3d_occupancy_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t nvox, uint8_t* d_occ);
int pb3d_occupancy(pb3d_ctx* ctx, const uint8_t* grid_rgb, int64_t nvox, uint8_t* occ);

/* ---- apply_colored_mask_to_voxel_grid, reference utils/voxel_carving_utils.py:128-136 ----
 * out[x,y,z,:] = rgb_hw3[y,x,:] where carved[x,y,z] == 1 else 0. */
int pb3d_color_apply_dev(pb3d_ctx* ctx, const uint8_t* d_carved, int64_t W, int64_t H, int64_t D,
                         const uint8_t* d_rgb_hw3, uint8_t* d_out);
int pb3d_color_apply(pb3d_ctx* ctx, const uint8_t* carved, int64_t W, int64_t H, int64_t D,
                     const uint8_t* rgb_hw3, uint8_t* out);

/* ---- global_carve, reference utils/voxel_carving_utils.py:269-298 -------------------------
 * ones((w,h,w)) -> process_voxel_grid -> colour.  bin_hw: (h,w) truthiness uint8,
 * rgb_hw3: (h,w,3).  out: (w,h,w,3).  The x-range [x0,x1) variant computes only that slab
 * of the output (slab pointer = start of the slab), for sharded runs. */
int pb3d_global_carve_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_rgb_hw3, int64_t h, int64_t w,
                          int angle_interval, int64_t x0, int64_t x1, uint8_t* d_out_slab);
int pb3d_global_carve(pb3d_ctx* ctx, const uint8_t* bin_hw, const uint8_t* rgb_hw3, int64_t h, int64_t w,
                      int angle_interval, uint8_t* out);

/* ---- part_carve, reference utils/voxel_carving_utils.py:139-160 ---------------------------
 * Job j is described by two (W,H) uint8 0/1 images (the shim derives them from the
 * semantic mask, :143-151): mask_sub[j] = mask2d.T gates `sub`; mask_carve[j] =
 * _mask_to_wh(mask2d.T) is what process_voxel_grid uses (differs only when W == H).
 * job_skip[j] != 0 marks jobs whose mask2d is empty (:148).  out is zero where no job keeps. */
int pb3d_part_carve_dev(pb3d_ctx* ctx, const uint8_t* d_colored, int64_t W, int64_t H, int64_t D,
                        const uint8_t* d_mask_sub, const uint8_t* d_mask_carve, const int* job_angle,
                        const int* job_skip, int njobs, uint8_t* d_out);
int pb3d_part_carve(pb3d_ctx* ctx, const uint8_t* colored, int64_t W, int64_t H, int64_t D,
                    const uint8_t* mask_sub, const uint8_t* mask_carve, const int* job_angle,
                    const int* job_skip, int njobs, uint8_t* out);

/* ---- grid -> points: get_voxel_points_by_parts (reference utils/voxel_utils.py:7-21) and
 * voxel_grid_to_points (:35-51).  Grid (A0,A1,A2[,C]); a voxel on the stride lattice is
 * selected if its RGB equals one of colors[ncolors][3] (ncolors > 0, C == 3) or if any of
 * its C channels is non-zero (ncolors == 0).  Points come out in numpy.where order as
 * float32 (a2,a1,a0)*stride plus the voxel's C bytes.  count first, then fill with
 * buffers of exactly n rows. */
int pb3d_points_count_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C,
                          const uint8_t* colors, int ncolors, int stride, int64_t* n);
int pb3d_points_fill_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C,
                         const uint8_t* colors, int ncolors, int stride, int64_t n, float* d_pts, uint8_t* d_cols);
/* One-pass form (stride 1) FOR CAPACITY-BOUNDED CALLERS ONLY -- it is the SLOWER form on MI355X: count + fill above takes 2.3 ms at
 * 1024^3 / 416 M points, this entry 4.1-4.5 ms (a decoupled look-back waits for flags that cross XCDs).  Use
 * it when the output buffers exist before the count is known and a second sweep cannot be scheduled; otherwise count, then fill.
 * The selected voxels are counted AND written in a single sweep of the grid (ordered stream compaction).
 * capacity = rows d_pts (n x 3 float32) / d_cols (n x C) can take; *n is the number of selected voxels.  If *n > capacity the
 * buffers hold only whole blocks that fitted and the call must be repeated with capacity >= *n.  Synchronises (returns *n). */
int pb3d_points_extract_dev(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, int C, const uint8_t* colors, int ncolors,
                            int64_t capacity, float* d_pts, uint8_t* d_cols, int64_t* n);
int pb3d_points_count(pb3d_ctx* ctx, const uint8_t* grid, int64_t A0, int64_t A1, int64_t A2, int C,
                      const uint8_t* colors, int ncolors, int stride, int64_t* n);
int pb3d_points_fill(pb3d_ctx* ctx, int64_t n, float* pts, uint8_t* cols); /* after pb3d_points_count on the same ctx */

/* ---- project_colored_voxels, reference utils/projection_utils.py:5-23 ---------------------
 * R = look_at_rotation(cam, target) (host, reference utils/camera_geometry.py:3-14) is
 * passed in.  pts: (n,3) float32 (pts_f64 = 0) or float64 (1); cols (n,3) uint8.
 * prec[4] = arithmetic width (0 = float32, 1 = float64) of {matmul+divide, *f, +cx, +cy}
 * as NumPy-2 promotion decides it from the caller's types.  img (Himg,Wimg,3) is fully
 * written; among points landing on one pixel the last in input order wins. */
int pb3d_project_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n,
                     const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4],
                     int Himg, int Wimg, uint8_t* d_img);
int pb3d_project(pb3d_ctx* ctx, const void* pts, int pts_f64, const uint8_t* cols, int64_t n,
                 const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4],
                 int Himg, int Wimg, uint8_t* img);

/* ---- z-buffer visibility (row N5), reference utils/eval_helpers_intra.py:134-190 -------------------------
 * Same pinhole arithmetic as pb3d_project, but points with Z <= 1e-6 are dropped (no clamp).
 * depth_buffer: zbuf[v,u] = min Z of the points landing on the pixel as float32, +inf where none (:134-163).
 * visible_mask: mask[v,u] = 1 where some point has |Z - zbuf[v,u]| < eps (:168-190); eps_f32 != 0 when the
 * comparison is made in float32 (float32 camera and a Python-float eps). */
int pb3d_depth_buffer_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, int64_t n, const double R[9], const double cam[3], double f,
                          double cx, double cy, const int prec[4], int Himg, int Wimg, float* d_zbuf);
int pb3d_visible_mask_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, int64_t n, const double R[9], const double cam[3], double f,
                          double cx, double cy, const int prec[4], const float* d_zbuf, int Himg, int Wimg, double eps, int eps_f32,
                          uint8_t* d_mask);

/* ---- compute_partwise_iou, reference utils/camera_estimation.py:770-787 -------------------
 * per colour k: inter[k] = #(a==c & b==c), uni[k] = #(a==c | b==c) over npix RGB pixels. */
int pb3d_partwise_iou_dev(pb3d_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, int64_t npix,
                          const uint8_t* colors, int ncolors, int64_t* inter, int64_t* uni);
int pb3d_partwise_iou(pb3d_ctx* ctx, const uint8_t* a, const uint8_t* b, int64_t npix,
                      const uint8_t* colors, int ncolors, int64_t* inter, int64_t* uni);

/* ---- K cameras per launch (row N4): the objective of the camera aligner, reference utils/camera_estimation.py:597-603
 * (`evaluate` = project_colored_voxels + compute_partwise_iou), as the random / coordinate / Powell loops (:606-725) call it
 * hundreds of times on the same points.  For every camera k: inter[k*ncolors + c], uni[k*ncolors + c] are the counts
 * pb3d_partwise_iou would return for pb3d_project's image of camera k against the (Himg,Wimg,3) part image d_seg.  The K
 * images are never materialised; one counter download per call.  pb3d_look_at_batch is the host half: K look-at rotations
 * (reference utils/camera_geometry.py:3-14) in the caller's float width; dot_mode says how this host's NumPy rounds the
 * 3-element dot product inside numpy.linalg.norm (0: separate multiplies and adds, 1: one FMA chain from the first product,
 * 2: exact products accumulated in double then rounded -- float32 only, 3: the FMA chain from the last product, 4: float32 products
 * accumulated in double -- float32 only, OpenBLAS's sdot tail loop); the Python shim calibrates it against NumPy itself. */
typedef struct pb3d_camera {
    double R[9], cam[3], f, cx, cy;
    int prec[4];
} pb3d_camera;
int pb3d_project_iou_batch_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n, const pb3d_camera* cams,
                               int ncams, int Himg, int Wimg, const uint8_t* d_seg, const uint8_t* colors, int ncolors, int64_t* inter,
                               int64_t* uni);
int pb3d_look_at_batch(const void* eye, const void* target, int is_f64, int64_t count, int dot_mode, double* R9);
/* K deform tuples per launch: the part-wise grid search of reference utils/deformation_estimation.py:148-258 (the slider loop
 * :100-146 / save_params :262-284 automated).  For tuple k = deforms5[5k .. 5k+4] = (scale_xz, scale_y, kx, ky, kz -- the scalars of
 * pb3d_deform_count): deform_coords of the part's points, the bounds filter against the (A0,A1,A2) grid, the float32 projection
 * with ONE fixed camera and the IoU of the part's colour against the (Himg,Wimg,3) image d_seg.  inter[k] / uni[k] are the counts
 * compute_partwise_iou forms; nvalid[k] counts the in-bounds (point, jitter) evaluations before np.unique (0 <=> upstream's
 * "No deformed voxels within bounds").  Every point of a part carries the part colour, so the projected image is the set of
 * pixels hit and neither np.unique nor the write order can change the counts. */
int pb3d_deform_iou_batch_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, const double* deforms5, int ntuples, int64_t A0, int64_t A1,
                              int64_t A2, const pb3d_camera* cam, int Himg, int Wimg, const uint8_t* d_seg, const uint8_t color[3],
                              int64_t* inter, int64_t* uni, int64_t* nvalid);

/* ---- part-wise deformation, reference utils/deformation_estimation.py:70-98 (deform_coords) -------
 * Seven jitters (0, +-0.25 per axis) of the part's points (voxel indices as float32), each centred on
 * its own mean, x/z scaled by sxz and pushed by kx/kz * sign, y scaled by sy and shifted by -ky (the
 * caller forms kx = shift_xz*W/W_img, ky = shift_y*H/H_img, kz = shift_xz*D/W_img as Python floats, :76-78),
 * rounded half-to-even; the result is np.unique(axis=0): unique rows in lexicographic (x,y,z) order as
 * int64.  count, then fill with a buffer of n_unique rows.  paint writes rgb at grid[z,y,x] of every
 * in-bounds deformed coordinate (:120-124, :306-309 for a uniformly coloured part); scatter_colors is the
 * general grid[z,y,x] = cols[k] assignment for unique rows. */
int pb3d_deform_count_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                          int64_t* n_unique);
int pb3d_deform_fill_dev(pb3d_ctx* ctx, int64_t n_unique, int64_t* d_coords);
int pb3d_deform_count(pb3d_ctx* ctx, const float* pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                      int64_t* n_unique);
int pb3d_deform_fill(pb3d_ctx* ctx, int64_t n_unique, int64_t* coords);
int pb3d_deform_paint_dev(pb3d_ctx* ctx, const float* d_pts, int64_t n, double sxz, double sy, double kx, double ky, double kz,
                          int64_t A0, int64_t A1, int64_t A2, const uint8_t rgb[3], uint8_t* d_grid);
int pb3d_scatter_colors_dev(pb3d_ctx* ctx, const int64_t* d_coords, const uint8_t* d_cols, int64_t m, int64_t A0, int64_t A1,
                            int64_t A2, uint8_t* d_grid);

/* ---- connected components and the steps built on them (rows N1/N2) ---------------------------------
 * pb3d_label_color_dev: scipy.ndimage.label(all(grid == color, axis=-1)) with the default 6-connected
 * structure (call sites reference utils/voxel_carving_utils.py:175, :254): int32 labels 1..ncomp numbered in
 * raster order of each component's first voxel, 0 elsewhere.  pb3d_component_stats_dev: per component the
 * bounding box (lo inclusive, hi exclusive -- :184-185), voxel count and coordinate sums (for the means of
 * :259), returned in host arrays.  crop_occupancy / component_paste are the per-component steps of
 * left_right_guided_carve (:190-192, :197-201); recolor_components is :263-265; extrude is
 * extrude_from_surface (:213-248): out = grid with `depth` cells from the first occupied voxel along the
 * axis painted fill_color (NULL = zeros) where the 2-D mask allows (axis 2: valid[x*H+y]; axis 0:
 * valid[y*valid_w + z], the upstream indexing of its (H,W) mask). */
int pb3d_label_color_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                         int32_t* d_labels, int64_t* ncomp);
/* label_color + component_stats in one pass and ONE host round trip: the statistics are gathered by the labelling's last kernel
 * (the labels are in registers there).  The host arrays hold `cap` components; *stats_valid = 0 when there are more (only *ncomp and
 * the labels are then valid: call pb3d_component_stats_dev). */
int pb3d_label_color_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3],
                               int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                               int64_t* coord_sum, int* stats_valid);
/* members_only = 1: the entries of d_labels at voxels that do NOT carry the colour are left UNSPECIFIED (the zeros are more than half of the
 * labelling's traffic).  Such a volume may only be consumed, before the next pb3d_label_* call on the context, by the entries that
 * consult the labelling's membership bits: pb3d_guided_carve[_label | _color]_dev and pb3d_recolor_last_labelled_dev (and by reading the
 * labels of voxels known to carry the colour).  members_only = 0: a full label volume (0 elsewhere), as pb3d_label_color_dev writes.
 * The bits stay valid until the next pb3d_label_* call on the context (calls that only grow OTHER scratch buffers, e.g.
 * pb3d_component_stats_dev, do not invalidate them). */
/* The components of SEVERAL colours in ONE labelling sequence (reference utils/voxel_carving_utils.py:338 calls :175 once per part colour
 * on a grid whose membership of the OTHER colours left_right_guided_carve never changes: it only clears or restores voxels of its own
 * colour, :199-201): the colour grid is read once, one forest serves every colour (a run = a maximal run of one colour).  colors =
 * ncolors x 3 bytes, pairwise different, 1 <= ncolors <= PB3D_CCL_MAX_COLORS.  ncomp[k], stats_valid[k] per colour; the statistics arrays
 * are [ncolors][cap][...]; labels are numbered PER COLOUR (as one scipy.ndimage.label call per colour numbers them), so the label of a
 * voxel means something only together with its colour: d_labels is for the consumers that take a colour index
 * (pb3d_guided_carve_color_dev) or, with ncolors == 1, for anyone. */
#define PB3D_CCL_MAX_COLORS 8
int pb3d_label_colors_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const uint8_t* colors, int ncolors,
                                int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                int64_t* coord_sum, int* stats_valid);
int pb3d_component_stats_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t A0, int64_t A1, int64_t A2, int64_t ncomp,
                             int64_t* bbox_lo_hi, int64_t* count, int64_t* coord_sum);
int pb3d_crop_occupancy_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3],
                            const int64_t hi[3], uint8_t* d_occ);
int pb3d_component_paste_dev(pb3d_ctx* ctx, const uint8_t* d_colored, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                             int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved);
/* The whole component loop of left_right_guided_carve (reference utils/voxel_carving_utils.py:178-201) in one call, IN PLACE on the
 * resident colour grid: for component k + 1 of d_labels with box bbox_lo_hi[6k..] (lo inclusive, hi exclusive) and crop mask
 * masks[mask_off[k] ..] ((Wc,Hc) uint8 truthiness, host memory, mask_bytes in all) the crop's occupancy goes through
 * process_voxel_grid(., crop mask, angle_interval) and the component's voxels that do not survive are cleared; carved_counts[k] (host)
 * = the log's "carved voxels" (:195).  One launch pair per batch of components (all 32-plane slices of all crops resident in LDS for
 * every rotation step).  *took = 0 (nothing done): a crop's slice does not fit the LDS -- run the per-component entries below.
 * The call returns after the device has finished (the counts are host values). */
int pb3d_guided_carve_dev(pb3d_ctx* ctx, uint8_t* d_grid_rgb, const int32_t* d_labels, int64_t W, int64_t H, int64_t D, int64_t ncomp,
                          const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off, int64_t mask_bytes, int angle_interval,
                          int64_t* carved_counts, int* took);
/* ... for colour `color_index` of the last pb3d_label_colors_stats_dev / pb3d_label_values_stats_dev call on the context (d_labels is the
 * volume that call wrote; channels = 3: colour grid, 1: label volume).  PB3D_EINVAL when the labelling's membership bits are gone. */
int pb3d_guided_carve_color_dev(pb3d_ctx* ctx, uint8_t* d_grid, const int32_t* d_labels, int color_index, int channels, int64_t W, int64_t H,
                                int64_t D, int64_t ncomp, const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off,
                                int64_t mask_bytes, int angle_interval, int64_t* carved_counts, int* took);
/* ... QUEUED: the call returns without waiting; the "carved voxels" counts are accumulated in d_counts (device memory, ncomp entries, cleared
 * by the call) and every host argument may be reused on return (they are staged through pb3d_h2d_async's ring).  partwise_carve
 * (reference :338-346) queues the component loops of all its parts behind ONE labelling and reads the counts once, at the end. */
int pb3d_guided_carve_queue_dev(pb3d_ctx* ctx, uint8_t* d_grid, const int32_t* d_labels, int color_index, int channels, int64_t W, int64_t H,
                                int64_t D, int64_t ncomp, const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off,
                                int64_t mask_bytes, int angle_interval, int64_t* d_counts, int* took);
/* recolor_backward_components (reference utils/voxel_carving_utils.py:252-266) on a resident grid WITHOUT a host round trip: the
 * components of `color` are labelled (into d_labels, members only) with their statistics left on the device, the keep_k components with
 * the smallest mean coordinate on sort_axis are kept (float64 means as np.mean gives them; equal means keep their numbering order, as
 * the stable sorted() of :261 does), the others painted new_color -- all queued on the context's stream.  d_status (device, two int64,
 * may be NULL): [0] = number of components, [1] = 1 when the device could not decide (more than 2048 components: nothing was painted;
 * run pb3d_label_color_stats_dev + pb3d_recolor_last_labelled_dev instead).  channels = 3: colour grid; 1: label volume (color[0],
 * new_color[0] are label values). */
int pb3d_recolor_backward_dev(pb3d_ctx* ctx, uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3], const uint8_t new_color[3],
                              int keep_k, int sort_axis, int channels, int32_t* d_labels, int64_t* d_status);
/* The rest of the notebook-1 chain on the 1-byte LABEL form of a palette grid (row N3; label 0 = empty, the others index a palette):
 * the same kernels with one byte per voxel -- components of the voxels that carry `value`, the fused component loop, extrusion
 * (fill_label < 0: clear), recolouring and the output orientation.  Expanding a result with the palette gives the bytes of the RGB
 * entry (tests: label chain == RGB chain == the reference's stage digests). */
int pb3d_label_value_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, uint8_t value, int32_t* d_labels,
                               int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count, int64_t* coord_sum,
                               int* stats_valid);
int pb3d_label_values_stats_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, const uint8_t* values, int nvalues,
                                int32_t* d_labels, int64_t* ncomp, int64_t cap, int members_only, int64_t* bbox_lo_hi, int64_t* count,
                                int64_t* coord_sum, int* stats_valid);
int pb3d_guided_carve_label_dev(pb3d_ctx* ctx, uint8_t* d_grid_lab, const int32_t* d_labels, int64_t W, int64_t H, int64_t D, int64_t ncomp,
                                const int64_t* bbox_lo_hi, const uint8_t* masks, const int64_t* mask_off, int64_t mask_bytes, int angle_interval,
                                int64_t* carved_counts, int* took);
/* the per-component steps (crops too large for the fused loop) on a label volume */
int pb3d_crop_occupancy_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3],
                                  const int64_t hi[3], uint8_t* d_occ);
int pb3d_component_paste_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, const int32_t* d_labels, int32_t id, const uint8_t* d_carved_occ,
                                   int64_t A0, int64_t A1, int64_t A2, const int64_t lo[3], const int64_t hi[3], uint8_t* d_carved);
int pb3d_extrude_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                           int axis, int plus, int depth, int fill_label, uint8_t* d_out);
int pb3d_recolor_components_label_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                      uint8_t new_label, uint8_t* d_grid_lab);
int pb3d_orient_label_dev(pb3d_ctx* ctx, const uint8_t* d_grid_lab, int64_t W, int64_t H, int64_t D, uint8_t* d_out);
/* *d_count (a device int64 the caller has zeroed) += number of non-zero bytes of d_bytes[0..n): the "carved voxels" figure of
 * left_right_guided_carve's log (reference utils/voxel_carving_utils.py:197), without a host round trip per component. */
int pb3d_count_nonzero_dev(pb3d_ctx* ctx, const uint8_t* d_bytes, int64_t n, int64_t* d_count);
int pb3d_recolor_components_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                const uint8_t new_color[3], uint8_t* d_grid_rgb);
/* The same for the label volume the LAST pb3d_label_* call on this context wrote, untouched since (PB3D_EINVAL otherwise): the
 * labelling's 1-bit-per-voxel membership array is still on the device, so only the members' labels are read (9 MB of bits instead of
 * 292 MB of labels at 512 x 278 x 512).  channels = 3: d_grid is a colour grid; 1: a label volume and new_color[0] the new label. */
int pb3d_recolor_last_labelled_dev(pb3d_ctx* ctx, const int32_t* d_labels, int64_t nvox, const uint8_t* comp_flag, int64_t ncomp,
                                   const uint8_t new_color[3], uint8_t* d_grid, int channels);
/* d_out may be d_grid_rgb itself (in place: nothing but the painted cells is written). */
int pb3d_extrude_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, const uint8_t* d_valid, int64_t valid_w,
                     int axis, int plus, int depth, const uint8_t* fill_color, uint8_t* d_out);
/* notebook-1 output orientation, reference utils/voxel_carving_utils.py:384-385:
 * out (D,H,W,3) = flip(grid.transpose(2,1,0,3), axis=1) of the (W,H,D,3) grid, C-contiguous. */
int pb3d_orient_dev(pb3d_ctx* ctx, const uint8_t* d_grid_rgb, int64_t W, int64_t H, int64_t D, uint8_t* d_out);

/* ---- the 1-byte LABEL form of a semantic grid (row N3: what sits either side of the path) -------------------------------------
 * A semantic grid / mask only holds (0,0,0) and the colours of a small palette (reference utils/config.py:29-43, masks from
 * reference utils/mask_utils.py:14-87), so a voxel or pixel is one byte: label 0 <-> (0,0,0), label k <-> palette[k-1]
 * (npal <= 254 distinct non-black colours).  rgb_to_label fails with PB3D_EINVAL on a colour outside palette + black;
 * label_to_rgb on a label above npal.  The carve ops take label volumes as they are:
 *   carve_voxel_grid_with_masks  = pb3d_carve_mask_dev with C = 1 (reference utils/voxel_carving_utils.py:76-87);
 *   global_carve                 = pb3d_global_carve_label_dev: (w,h,w) labels = label_hw[y,x] where the carve keeps (:269-298);
 *   part_carve                   = pb3d_part_carve_label_dev, same job description as pb3d_part_carve_dev (:139-160).
 * Expanding any of their results with pb3d_label_to_rgb_dev gives the bytes of the RGB entry points. */
int pb3d_rgb_to_label_dev(pb3d_ctx* ctx, const uint8_t* d_rgb, int64_t nvox, const uint8_t* palette, int npal, uint8_t* d_label);
int pb3d_label_to_rgb_dev(pb3d_ctx* ctx, const uint8_t* d_label, int64_t nvox, const uint8_t* palette, int npal, uint8_t* d_rgb);
int pb3d_global_carve_label_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_label_hw, int64_t h, int64_t w, int angle_interval,
                                uint8_t* d_out);
int pb3d_part_carve_label_dev(pb3d_ctx* ctx, const uint8_t* d_label, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_sub,
                              const uint8_t* d_mask_carve, const int* job_angle, const int* job_skip, int njobs, uint8_t* d_out);

/* ---- seeded synthetic inputs generated on the device ---------------------
 * mask16: labels (S,S) uint8 in 0..15 by the closed formula scaled from S=1024; binary and
 * rgb derive from it.  Any output pointer may be NULL.  sem grid: palette[label16 of a
 * splitmix64 hash of the voxel index] (kind 0) -- worst case for any sparsity trick. */
int pb3d_synth_mask16_dev(pb3d_ctx* ctx, int64_t S, uint8_t* d_label_hw, uint8_t* d_binary_hw, uint8_t* d_rgb_hw3,
                          uint8_t* d_binary_wh);
int pb3d_synth_sem_dev(pb3d_ctx* ctx, int64_t x0, int64_t x1, int64_t H, int64_t D, uint64_t seed, uint8_t* d_slab_rgb);
int pb3d_synth_occ_dev(pb3d_ctx* ctx, int64_t x0, int64_t x1, int64_t H, int64_t D, uint64_t seed, uint8_t* d_slab);
int pb3d_synth_palette16(uint8_t palette[48]);

/* ---- multi-GPU: slab reassembly with ONE RCCL all-gather over xGMI -----------------------
 * One process per GPU.  Rank 0 obtains a 128-byte id, the launcher distributes it, every
 * rank calls pb3d_comm_init.  pb3d_allgather_dev gathers `bytes_per_rank` from each rank's
 * d_send into d_recv[rank * bytes_per_rank ...] (d_send may be the rank's own slot: in place). */
int pb3d_comm_unique_id(uint8_t id[128]);
int pb3d_comm_init(pb3d_ctx* ctx, const uint8_t id[128], int rank, int nranks);
/* what the communicator itself reports (ncclCommUserRank / ncclCommCount): lets a launcher confirm RCCL saw every rank */
int pb3d_comm_info(pb3d_ctx* ctx, int* rank, int* nranks);
int pb3d_allgather_dev(pb3d_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
int pb3d_comm_destroy(pb3d_ctx* ctx);
/* Sharded forms of the carve: one process per GPU, communicator from pb3d_comm_init.  Rank r of n owns the X-planes
 * [r W/n, (r+1) W/n) (W % n == 0): it carves ITS slab (d_grid_slab: planes x H x D x C bytes; d_mask_wh: the full (W,H) mask) into its
 * slot of d_out_full and ONE in-place ncclAllGather leaves the whole carved volume on every rank.  Asynchronous like every *_dev
 * entry.  carve_labels: the same on a label slab (1 B/voxel over xGMI instead of 3), then -- if d_rgb_full is not NULL -- the
 * local expansion of the reassembled label volume to RGB (labels above npal expand to black; no read-back). */
int pb3d_carve_mask_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_grid_slab, int64_t W, int64_t H, int64_t D, int C, const uint8_t* d_mask_wh,
                                uint8_t* d_out_full);
int pb3d_global_carve_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_rgb_hw3, int64_t h, int64_t w, int angle_interval,
                                  uint8_t* d_out_full);
int pb3d_carve_labels_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_label_slab, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                                  uint8_t* d_label_full, const uint8_t* palette, int npal, uint8_t* d_rgb_full);
/* points partition of the projection (project_colored_voxels, reference utils/projection_utils.py:5-23): a rank projects its
 * contiguous range [index_base, index_base + n) of the point list into 64-bit keys (global index + 1) << 24 | rgb (zero = no
 * point); pb3d_allreduce_max_u64_dev (in place, ncclMax) merges the ranks' key images; resolve writes the (H,W,3) image. */
int pb3d_project_keys_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n, int64_t index_base,
                          const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4], int Himg, int Wimg,
                          uint64_t* d_keys);
int pb3d_project_resolve_keys_dev(pb3d_ctx* ctx, const uint64_t* d_keys, int Himg, int Wimg, uint8_t* d_img);
int pb3d_allreduce_max_u64_dev(pb3d_ctx* ctx, void* d_buf, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* PB3D_H */
