// K7 pinhole scatter projection (deterministic last-writer-wins) and K8 per-part IoU counts.
//
// reference utils/projection_utils.py:5-23:  pc = (p - cam) @ R.T ; Z<1e-8 -> 1e-8 ;
//   u = (X/Z)*f + cx ; v = -(Y/Z)*f + cy ; rint ; bounds ; img[v,u] = colour.
// NumPy's fancy assignment resolves duplicate pixels by "last in input order wins"; that is
// reproduced with an atomicMax on (point index + 1) per pixel followed by a resolve pass, so the
// image is identical from run to run and to the CPU path.
// Arithmetic: each component of pc is the FMA chain fma(d2,r2, fma(d1,r1, d0*r0)) NumPy's gemm
// produces (float32 or float64 per prec[0]); the later stages are separate IEEE ops whose width is
// prec[1..3].  float32 +,-,*,/ are evaluated in double and rounded once (exact since 53 >= 2*24+2);
// the float32 FMA chain uses the hardware's single-rounded v_fma_f32.
#include "pb3d_internal.h"
#include "project_point.h"

namespace {

using namespace pb3d_proj;

// Points are visited from the LAST index down: the winner of a pixel is its largest point index, so once the
// high indices have claimed their pixels the remaining points mostly lose on a plain read and skip the atomic.
__global__ __launch_bounds__(256) void k_project_points(const void* __restrict__ pts, i64 n, ProjParams P,
                                                        u32* __restrict__ winner) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = n - 1 - t;
        int ui, vi;
        if (project_point<0>(P, pts, i, &ui, &vi)) {
            u32* w = &winner[(i64)vi * P.Wimg + ui];
            const u32 mine = (u32)(i + 1);
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(w, mine);
        }
    }
}

__global__ __launch_bounds__(256) void k_project_resolve(const u32* __restrict__ winner, const u8* __restrict__ cols,
                                                         u8* __restrict__ img, i64 npix) {
    for (i64 px = (i64)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (i64)gridDim.x * blockDim.x) {
        const u32 w = winner[px];
        u8 r = 0, g = 0, b = 0;
        if (w) {
            const u8* c = cols + (i64)(w - 1) * 3;
            r = c[0]; g = c[1]; b = c[2];
        }
        img[3 * px] = r; img[3 * px + 1] = g; img[3 * px + 2] = b;
    }
}

// z-buffer: nearest depth per pixel.  Depths are positive, so float32 order == order of their bit patterns and
// the sequential "if z < zbuf: zbuf = z" loop of the reference equals an atomicMin on the float32 bits of z
// (for float64 cameras the stored value is float32(z); float32 rounding is monotone, so min and rounding commute).
__global__ __launch_bounds__(256) void k_depth_points(const void* __restrict__ pts, i64 n, ProjParams P, u32* __restrict__ zbits) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        int ui, vi; double z;
        if (project_point<1>(P, pts, i, &ui, &vi, &z)) atomicMin(&zbits[(i64)vi * P.Wimg + ui], __float_as_uint((float)z));
    }
}

__global__ __launch_bounds__(256) void k_visible_points(const void* __restrict__ pts, i64 n, ProjParams P, const float* __restrict__ zbuf,
                                                        double eps, int eps_f32, u8* __restrict__ mask) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        int ui, vi; double z;
        if (!project_point<1>(P, pts, i, &ui, &vi, &z)) continue;
        const i64 px = (i64)vi * P.Wimg + ui;
        const float zb = zbuf[px];
        bool hit;
        if (P.t0) hit = fabs(__dsub_rn(z, (double)zb)) < eps;                       // float64 z - float32 zbuf -> float64
        else {
            const float dz = fabsf(__fsub_rn((float)z, zb));                         // float32 z - float32 zbuf
            hit = eps_f32 ? dz < (float)eps : (double)dz < eps;                     // weak Python float -> float32 compare
        }
        if (hit) mask[px] = 1;
    }
}

// Sharded form of the scatter (points partition): every rank projects its contiguous range of the
// point list into a private image of 64-bit keys ((global index + 1) << 24 | b << 16 | g << 8 | r); the last-writer-wins
// rule is a max over the key, so ONE all-reduce(max) over the ranks followed by a local resolve gives every rank the
// image the unsharded call produces, with no colour look-up across ranks.
__global__ __launch_bounds__(256) void k_project_keys(const void* __restrict__ pts, const u8* __restrict__ cols, i64 n, i64 index_base,
                                                      ProjParams P, unsigned long long* __restrict__ keys) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = n - 1 - t;
        int ui, vi;
        if (project_point<0>(P, pts, i, &ui, &vi)) {
            unsigned long long* w = &keys[(i64)vi * P.Wimg + ui];
            const unsigned long long mine = ((unsigned long long)(index_base + i + 1) << 24) | ((unsigned long long)cols[3 * i + 2] << 16) |
                                            ((unsigned long long)cols[3 * i + 1] << 8) | (unsigned long long)cols[3 * i];
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(w, mine);
        }
    }
}

__global__ __launch_bounds__(256) void k_resolve_keys(const unsigned long long* __restrict__ keys, u8* __restrict__ img, i64 npix) {
    for (i64 px = (i64)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (i64)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[px];
        img[3 * px] = (u8)(k & 0xff); img[3 * px + 1] = (u8)((k >> 8) & 0xff); img[3 * px + 2] = (u8)((k >> 16) & 0xff);
    }
}

// ------------------------------------------------------------------------------------------------
// All-float32 fast path (float32 points, float32 camera: prec = {0,0,0,0} -- what notebooks 2/3 pass).
// Every stage of project_point is then one IEEE float32 operation, so it is evaluated directly in
// float32 (v_fma_f32 chains, correctly rounded v_div sequence) instead of "double, rounded once":
// ~3x fewer VALU cycles, which is what bounded the generic kernel.  Points are read four at a time as
// three 16-byte loads per lane.
// ------------------------------------------------------------------------------------------------
struct ProjF32 {
    float R[9], cam[3], f, cx, cy;
    int Himg, Wimg;
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ bool project_f32(const ProjF32& P, float x, float y, float z, int* ui, int* vi, float* zout) {
    const float d0 = __fsub_rn(x, P.cam[0]), d1 = __fsub_rn(y, P.cam[1]), d2 = __fsub_rn(z, P.cam[2]);
    float pc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = __fmaf_rn(d2, P.R[3 * r + 2], __fmaf_rn(d1, P.R[3 * r + 1], __fmul_rn(d0, P.R[3 * r])));
    float Z = pc[2];
    if (MODE == 0) {
        if (Z < 1e-8f) Z = 1e-8f;
    } else {
        if (!(Z > 1e-6f)) return false;
        *zout = Z;
    }
    const float qx = __fdiv_rn(pc[0], Z), qy = -__fdiv_rn(pc[1], Z);
    const float u = __fadd_rn(__fmul_rn(qx, P.f), P.cx), v = __fadd_rn(__fmul_rn(qy, P.f), P.cy);
    // 0 <= r < n as ONE unsigned compare of float bit patterns: for r >= +0 the patterns order like the values, negative
    // values / NaN / inf have patterns above every finite n; "+ 0.0f" turns the legitimate -0.0 (u in [-0.5, -0]) into +0.0
    const float ur = __fadd_rn(rintf(u), 0.0f), vr = __fadd_rn(rintf(v), 0.0f);
    if (!(__float_as_uint(ur) < __float_as_uint((float)P.Wimg) && __float_as_uint(vr) < __float_as_uint((float)P.Himg))) return false;
    *ui = (int)ur; *vi = (int)vr;
    return true;
}

// sink(i0, ok[4], px[4], z[4]) once per group of four points (i0 = index of the group's first point; ok = lands in the
// image; px = pixel index): the four projections are finished before the sink runs, so its four dependent image
// reads / atomics are issued back to back instead of one L2 round trip after another.  REVERSE visits the list last-first.
template <int MODE, class Sink>
__device__ __forceinline__ void group_f32(const ProjF32& P, const float q[12], int live, i64 i0, Sink& sink) {
    bool ok[4]; u32 px[4]; float z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int ui = 0, vi = 0;
        z[k] = 0.0f;
        ok[k] = k < live && project_f32<MODE>(P, q[3 * k], q[3 * k + 1], q[3 * k + 2], &ui, &vi, &z[k]);
        px[k] = (u32)vi * (u32)P.Wimg + (u32)ui;
    }
    sink(i0, ok, px, z);
}

template <int MODE, bool REVERSE, class Sink>
__device__ __forceinline__ void sweep_f32(const float* __restrict__ pts, i64 n, const ProjF32& P, bool vec, Sink sink) {
    const i64 nfull = n >> 2;                      // groups of four whole points
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x, gsz = (i64)gridDim.x * blockDim.x;
    float q[12];
    if (REVERSE && gtid == 0 && (n & 3)) {        // the ragged tail holds the highest indices: first in a reverse sweep
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = k < 3 * (int)(n & 3) ? pts[12 * nfull + k] : 0.0f;
        group_f32<MODE>(P, q, (int)(n & 3), 4 * nfull, sink);
    }
    if (vec) {
        // the next group's three vectors are requested before this group is projected (software prefetch)
        f32x4 nx[3];
        auto fetch = [&](i64 t) {
            const i64 c = REVERSE ? nfull - 1 - t : t;
            const f32x4* g = (const f32x4*)(pts + 12 * c);
#pragma unroll
            for (int k = 0; k < 3; ++k) nx[k] = g[k];   // plain loads: each 128-byte line is shared by three instructions (nontemporal: 1.4x slower)
        };
        if (gtid < nfull) fetch(gtid);
        for (i64 t = gtid; t < nfull; t += gsz) {
            const i64 c = REVERSE ? nfull - 1 - t : t;
#pragma unroll
            for (int k = 0; k < 3; ++k) { q[4 * k] = nx[k].x; q[4 * k + 1] = nx[k].y; q[4 * k + 2] = nx[k].z; q[4 * k + 3] = nx[k].w; }
            if (t + gsz < nfull) fetch(t + gsz);
            group_f32<MODE>(P, q, 4, 4 * c, sink);
        }
    } else {
        for (i64 t = gtid; t < nfull; t += gsz) {
            const i64 c = REVERSE ? nfull - 1 - t : t;
#pragma unroll
            for (int k = 0; k < 12; ++k) q[k] = pts[12 * c + k];
            group_f32<MODE>(P, q, 4, 4 * c, sink);
        }
    }
    if (!REVERSE && gtid == 0 && (n & 3)) {
#pragma unroll
        for (int k = 0; k < 12; ++k) q[k] = k < 3 * (int)(n & 3) ? pts[12 * nfull + k] : 0.0f;
        group_f32<MODE>(P, q, (int)(n & 3), 4 * nfull, sink);
    }
}

__global__ __launch_bounds__(256) void k_project_points_f32(const float* __restrict__ pts, i64 n, ProjF32 P, int vec, u32* __restrict__ winner) {
    sweep_f32<0, true>(pts, n, P, vec != 0, [&](i64 i0, const bool* ok, const u32* px, const float*) {
        u32 cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = ok[k] ? __hip_atomic_load(&winner[px[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
#pragma unroll
        for (int k = 3; k >= 0; --k)
            if (cur[k] < (u32)(i0 + k + 1)) atomicMax(&winner[px[k]], (u32)(i0 + k + 1));
    });
}

__global__ __launch_bounds__(256) void k_project_keys_f32(const float* __restrict__ pts, const u8* __restrict__ cols, i64 n, i64 index_base,
                                                          ProjF32 P, int vec, unsigned long long* __restrict__ keys) {
    sweep_f32<0, true>(pts, n, P, vec != 0, [&](i64 i0, const bool* ok, const u32* px, const float*) {
        unsigned long long cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = ok[k] ? __hip_atomic_load(&keys[px[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const i64 i = i0 + k;
            if ((cur[k] >> 24) < (unsigned long long)(index_base + i + 1)) {
                const unsigned long long mine = ((unsigned long long)(index_base + i + 1) << 24) | ((unsigned long long)cols[3 * i + 2] << 16) |
                                                ((unsigned long long)cols[3 * i + 1] << 8) | (unsigned long long)cols[3 * i];
                atomicMax(&keys[px[k]], mine);
            }
        }
    });
}

__global__ __launch_bounds__(256) void k_depth_points_f32(const float* __restrict__ pts, i64 n, ProjF32 P, int vec, u32* __restrict__ zbits) {
    sweep_f32<1, false>(pts, n, P, vec != 0, [&](i64, const bool* ok, const u32* px, const float* z) {
        u32 cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = ok[k] ? __hip_atomic_load(&zbits[px[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cur[k] > __float_as_uint(z[k])) atomicMin(&zbits[px[k]], __float_as_uint(z[k]));
    });
}

__global__ __launch_bounds__(256) void k_visible_points_f32(const float* __restrict__ pts, i64 n, ProjF32 P, int vec, const float* __restrict__ zbuf,
                                                            double eps, int eps_f32, u8* __restrict__ mask) {
    sweep_f32<1, false>(pts, n, P, vec != 0, [&](i64, const bool* ok, const u32* px, const float* z) {
        float zb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) zb[k] = ok[k] ? zbuf[px[k]] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = fabsf(__fsub_rn(z[k], zb[k]));
            if (ok[k] && (eps_f32 ? dz < (float)eps : (double)dz < eps)) mask[px[k]] = 1;
        }
    });
}

// the fast path applies when nothing in the call is float64
bool f32_path(const ProjParams& P, ProjF32* F) {
    if (P.pts_f64 || P.t0 || P.tm || P.tu || P.tv || P.Himg >= (1 << 24) || P.Wimg >= (1 << 24) || (i64)P.Himg * P.Wimg > 0xffffffffll) return false;
    for (int k = 0; k < 9; ++k) F->R[k] = (float)P.R[k];
    for (int k = 0; k < 3; ++k) F->cam[k] = (float)P.cam[k];
    F->f = (float)P.f; F->cx = (float)P.cx; F->cy = (float)P.cy;
    F->Himg = P.Himg; F->Wimg = P.Wimg;
    return true;
}
inline int vec_ok(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
inline unsigned f32_blocks(pb3d_ctx* ctx, i64 n) { return pb3d_stream_blocks(ctx, n / 4 + 1, 256, 0); }      // one workgroup per 1024 points

struct IouParams {
    int ncolors;
    u8 colors[3 * 32];
};

__global__ __launch_bounds__(256) void k_partwise_iou(const u8* __restrict__ a, const u8* __restrict__ b, i64 npix,
                                                      IouParams P, unsigned long long* __restrict__ counts) {
    __shared__ u32 acc[64];
    if (threadIdx.x < 64) acc[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 nloop = (npix + stride - 1) / stride;  // same trip count for every lane: ballots stay convergent
    for (i64 it = 0; it < nloop; ++it) {
        const i64 px = it * stride + (i64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = px < npix;
        u8 a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (live) {
            a0 = a[3 * px]; a1 = a[3 * px + 1]; a2 = a[3 * px + 2];
            b0 = b[3 * px]; b1 = b[3 * px + 1]; b2 = b[3 * px + 2];
        }
        for (int k = 0; k < P.ncolors; ++k) {
            const u8 c0 = P.colors[3 * k], c1 = P.colors[3 * k + 1], c2 = P.colors[3 * k + 2];
            const bool ma = live && a0 == c0 && a1 == c1 && a2 == c2;
            const bool mb = live && b0 == c0 && b1 == c1 && b2 == c2;
            const u32 ni = (u32)__popcll(__ballot(ma && mb)), nu = (u32)__popcll(__ballot(ma || mb));
            if (lane == 0) {
                if (ni) atomicAdd(&acc[2 * k], ni);
                if (nu) atomicAdd(&acc[2 * k + 1], nu);
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * P.ncolors && acc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}


// ------------------------------------------------------------------------------------------------
// Row N4: K cameras per launch.  The camera aligner (reference utils/camera_estimation.py:597-603 `evaluate`, driven by the
// random / coordinate / Powell loops :606-725) re-projects the SAME resident points for every candidate camera and keeps only
// the per-part IoU against the part image.  One launch projects all K cameras (blockIdx.y = camera; private winner image per
// camera), a second one resolves every camera's winners to colours on the fly and counts intersections / unions against the
// part image -- the K projected images are never written -- and ONE copy brings the K x P counters back.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project_points_batch(const void* __restrict__ pts, i64 n, const ProjParams* __restrict__ cams,
                                                              i64 npix, u32* __restrict__ winners) {
    const ProjParams P = cams[blockIdx.y];
    u32* winner = winners + (i64)blockIdx.y * npix;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 i = n - 1 - t;
        int ui, vi;
        if (project_point<0>(P, pts, i, &ui, &vi)) {
            u32* w = &winner[(i64)vi * P.Wimg + ui];
            const u32 mine = (u32)(i + 1);
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(w, mine);
        }
    }
}

__global__ __launch_bounds__(256) void k_project_points_batch_f32(const float* __restrict__ pts, i64 n, const ProjF32* __restrict__ cams, int vec,
                                                                  i64 npix, u32* __restrict__ winners) {
    const ProjF32 P = cams[blockIdx.y];
    u32* winner = winners + (i64)blockIdx.y * npix;
    sweep_f32<0, true>(pts, n, P, vec != 0, [&](i64 i0, const bool* ok, const u32* px, const float*) {
        u32 cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = ok[k] ? __hip_atomic_load(&winner[px[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
#pragma unroll
        for (int k = 3; k >= 0; --k)
            if (cur[k] < (u32)(i0 + k + 1)) atomicMax(&winner[px[k]], (u32)(i0 + k + 1));
    });
}

__global__ __launch_bounds__(256) void k_iou_winners_batch(const u32* __restrict__ winners, const u8* __restrict__ cols, const u8* __restrict__ seg,
                                                           i64 npix, int ncolors, const u8* __restrict__ colors, unsigned long long* __restrict__ counts) {
    __shared__ u32 acc[64];
    __shared__ u8 cl[96];
    if (threadIdx.x < 64) acc[threadIdx.x] = 0;
    if (threadIdx.x < 96) cl[threadIdx.x] = (int)threadIdx.x < 3 * ncolors ? colors[threadIdx.x] : (u8)0;
    __syncthreads();
    const u32* winner = winners + (i64)blockIdx.y * npix;
    const int lane = threadIdx.x & 63;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    const i64 nloop = (npix + stride - 1) / stride;  // same trip count for every lane: ballots stay convergent
    for (i64 it = 0; it < nloop; ++it) {
        const i64 px = it * stride + (i64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = px < npix;
        u8 a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (live) {
            const u32 w = winner[px];
            if (w) { const u8* c = cols + (i64)(w - 1) * 3; a0 = c[0]; a1 = c[1]; a2 = c[2]; }
            b0 = seg[3 * px]; b1 = seg[3 * px + 1]; b2 = seg[3 * px + 2];
        }
        for (int k = 0; k < ncolors; ++k) {
            const u8 c0 = cl[3 * k], c1 = cl[3 * k + 1], c2 = cl[3 * k + 2];
            const bool ma = live && a0 == c0 && a1 == c1 && a2 == c2;
            const bool mb = live && b0 == c0 && b1 == c1 && b2 == c2;
            const u32 ni = (u32)__popcll(__ballot(ma && mb)), nu = (u32)__popcll(__ballot(ma || mb));
            if (lane == 0) {
                if (ni) atomicAdd(&acc[2 * k], ni);
                if (nu) atomicAdd(&acc[2 * k + 1], nu);
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * ncolors && acc[threadIdx.x]) atomicAdd(&counts[(i64)blockIdx.y * 64 + threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}

}  // namespace

extern "C" {

int pb3d_project_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n,
                     const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4],
                     int Himg, int Wimg, uint8_t* d_img) {
    PB3D_REQUIRE(ctx && R && cam && prec, "pb3d_project: null argument");
    PB3D_REQUIRE(n >= 0 && n < 0xffffffffll, "pb3d_project: point count out of range");
    PB3D_REQUIRE(Himg >= 0 && Wimg >= 0, "pb3d_project: bad image shape");
    const i64 npix = (i64)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_img && (n == 0 || (d_pts && d_cols)), "pb3d_project: null buffer");
    for (int k = 0; k < 4; ++k) PB3D_REQUIRE(prec[k] == 0 || prec[k] == 1, "pb3d_project: prec[%d] must be 0 or 1", k);
    PB3D_REQUIRE(prec[1] >= prec[0] && prec[2] >= prec[1] && prec[3] >= prec[1], "pb3d_project: precision may only widen");
    void* winner;
    PB3D_TRY(pb3d_scratch(ctx, 8, (size_t)npix * sizeof(u32), &winner));
    PB3D_HIP(hipMemsetAsync(winner, 0, (size_t)npix * sizeof(u32), ctx->stream));
    ProjParams P;
    memcpy(P.R, R, sizeof(P.R)); memcpy(P.cam, cam, sizeof(P.cam));
    P.f = f; P.cx = cx; P.cy = cy;
    P.t0 = prec[0]; P.tm = prec[1]; P.tu = prec[2]; P.tv = prec[3];
    P.Himg = Himg; P.Wimg = Wimg; P.pts_f64 = pts_f64 ? 1 : 0;
    ProjF32 F;
    if (n > 0) {
        if (f32_path(P, &F))
            hipLaunchKernelGGL(k_project_points_f32, dim3(f32_blocks(ctx, n)), dim3(256), 0, ctx->stream, (const float*)d_pts, n, F,
                               vec_ok(d_pts), (u32*)winner);
        else
            hipLaunchKernelGGL(k_project_points, dim3(pb3d_stream_blocks(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, d_pts, n, P,
                               (u32*)winner);
        PB3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_project_resolve, dim3(pb3d_stream_blocks(ctx, npix, 256, 8)), dim3(256), 0, ctx->stream,
                       (const u32*)winner, d_cols, d_img, npix);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_project_keys_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n, int64_t index_base,
                          const double R[9], const double cam[3], double f, double cx, double cy, const int prec[4], int Himg, int Wimg,
                          uint64_t* d_keys) {
    PB3D_REQUIRE(ctx && R && cam && prec && n >= 0 && index_base >= 0 && Himg >= 0 && Wimg >= 0, "pb3d_project_keys: bad argument");
    PB3D_REQUIRE(index_base + n < (1ll << 39), "pb3d_project_keys: point index does not fit the key");
    const i64 npix = (i64)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_keys && (n == 0 || (d_pts && d_cols)), "pb3d_project_keys: null buffer");
    ProjParams P;
    PB3D_TRY(fill_proj(&P, pts_f64, R, cam, f, cx, cy, prec, Himg, Wimg));
    PB3D_HIP(hipMemsetAsync(d_keys, 0, (size_t)npix * sizeof(uint64_t), ctx->stream));
    ProjF32 F;
    if (n > 0) {
        if (f32_path(P, &F))
            hipLaunchKernelGGL(k_project_keys_f32, dim3(f32_blocks(ctx, n)), dim3(256), 0, ctx->stream, (const float*)d_pts, d_cols, n,
                               index_base, F, vec_ok(d_pts), (unsigned long long*)d_keys);
        else
            hipLaunchKernelGGL(k_project_keys, dim3(pb3d_stream_blocks(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, d_pts, d_cols, n,
                               index_base, P, (unsigned long long*)d_keys);
        PB3D_CHECK_LAUNCH();
    }
    return PB3D_OK;
}

int pb3d_project_resolve_keys_dev(pb3d_ctx* ctx, const uint64_t* d_keys, int Himg, int Wimg, uint8_t* d_img) {
    PB3D_REQUIRE(ctx && Himg >= 0 && Wimg >= 0, "pb3d_project_resolve_keys: bad argument");
    const i64 npix = (i64)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_keys && d_img, "pb3d_project_resolve_keys: null buffer");
    hipLaunchKernelGGL(k_resolve_keys, dim3(pb3d_stream_blocks(ctx, npix, 256, 8)), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)d_keys, d_img, npix);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_depth_buffer_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, int64_t n, const double R[9], const double cam[3], double f,
                          double cx, double cy, const int prec[4], int Himg, int Wimg, float* d_zbuf) {
    PB3D_REQUIRE(ctx && R && cam && prec && n >= 0 && Himg >= 0 && Wimg >= 0, "pb3d_depth_buffer: bad argument");
    const i64 npix = (i64)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_zbuf && (n == 0 || d_pts), "pb3d_depth_buffer: null buffer");
    ProjParams P;
    PB3D_TRY(fill_proj(&P, pts_f64, R, cam, f, cx, cy, prec, Himg, Wimg));
    PB3D_HIP(hipMemsetD32Async((hipDeviceptr_t)d_zbuf, 0x7f800000, (size_t)npix, ctx->stream));   // +inf
    ProjF32 F;
    if (n > 0) {
        if (f32_path(P, &F))
            hipLaunchKernelGGL(k_depth_points_f32, dim3(f32_blocks(ctx, n)), dim3(256), 0, ctx->stream, (const float*)d_pts, n, F,
                               vec_ok(d_pts), (u32*)d_zbuf);
        else
            hipLaunchKernelGGL(k_depth_points, dim3(pb3d_stream_blocks(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, d_pts, n, P, (u32*)d_zbuf);
        PB3D_CHECK_LAUNCH();
    }
    return PB3D_OK;
}

int pb3d_visible_mask_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, int64_t n, const double R[9], const double cam[3], double f,
                          double cx, double cy, const int prec[4], const float* d_zbuf, int Himg, int Wimg, double eps, int eps_f32,
                          uint8_t* d_mask) {
    PB3D_REQUIRE(ctx && R && cam && prec && n >= 0 && Himg >= 0 && Wimg >= 0, "pb3d_visible_mask: bad argument");
    const i64 npix = (i64)Himg * Wimg;
    if (npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_zbuf && d_mask && (n == 0 || d_pts), "pb3d_visible_mask: null buffer");
    ProjParams P;
    PB3D_TRY(fill_proj(&P, pts_f64, R, cam, f, cx, cy, prec, Himg, Wimg));
    PB3D_HIP(hipMemsetAsync(d_mask, 0, (size_t)npix, ctx->stream));
    ProjF32 F;
    if (n > 0) {
        if (f32_path(P, &F))
            hipLaunchKernelGGL(k_visible_points_f32, dim3(f32_blocks(ctx, n)), dim3(256), 0, ctx->stream, (const float*)d_pts, n, F,
                               vec_ok(d_pts), d_zbuf, eps, eps_f32, d_mask);
        else
            hipLaunchKernelGGL(k_visible_points, dim3(pb3d_stream_blocks(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, d_pts, n, P, d_zbuf, eps,
                               eps_f32, d_mask);
        PB3D_CHECK_LAUNCH();
    }
    return PB3D_OK;
}

int pb3d_partwise_iou_dev(pb3d_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, int64_t npix,
                          const uint8_t* colors, int ncolors, int64_t* inter, int64_t* uni) {
    PB3D_REQUIRE(ctx && inter && uni, "pb3d_partwise_iou: null argument");
    PB3D_REQUIRE(ncolors >= 0 && ncolors <= 32 && npix >= 0, "pb3d_partwise_iou: bad sizes");
    for (int k = 0; k < ncolors; ++k) inter[k] = uni[k] = 0;
    if (ncolors == 0 || npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_a && d_b && colors, "pb3d_partwise_iou: null buffer");
    void* counts;
    PB3D_TRY(pb3d_scratch(ctx, 9, 64 * sizeof(unsigned long long), &counts));
    PB3D_HIP(hipMemsetAsync(counts, 0, 64 * sizeof(unsigned long long), ctx->stream));
    IouParams P;
    P.ncolors = ncolors;
    memset(P.colors, 0, sizeof(P.colors));
    memcpy(P.colors, colors, (size_t)3 * ncolors);
    hipLaunchKernelGGL(k_partwise_iou, dim3(pb3d_stream_blocks(ctx, npix, 256, 4)), dim3(256), 0, ctx->stream, d_a, d_b, npix, P,
                       (unsigned long long*)counts);
    PB3D_CHECK_LAUNCH();
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, counts, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    const unsigned long long* h = (const unsigned long long*)ctx->pinned;
    for (int k = 0; k < ncolors; ++k) { inter[k] = (int64_t)h[2 * k]; uni[k] = (int64_t)h[2 * k + 1]; }
    return PB3D_OK;
}


int pb3d_project_iou_batch_dev(pb3d_ctx* ctx, const void* d_pts, int pts_f64, const uint8_t* d_cols, int64_t n, const pb3d_camera* cams, int ncams,
                               int Himg, int Wimg, const uint8_t* d_seg, const uint8_t* colors, int ncolors, int64_t* inter, int64_t* uni) {
    PB3D_REQUIRE(ctx && (ncams == 0 || (cams && inter && uni)), "pb3d_project_iou_batch: null argument");
    PB3D_REQUIRE(ncams >= 0 && n >= 0 && n < 0xffffffffll && Himg >= 0 && Wimg >= 0, "pb3d_project_iou_batch: bad sizes");
    PB3D_REQUIRE(ncolors >= 0 && ncolors <= 32, "pb3d_project_iou_batch: at most 32 colours");
    for (i64 k = 0; k < (i64)ncams * ncolors; ++k) inter[k] = uni[k] = 0;
    const i64 npix = (i64)Himg * Wimg;
    if (ncams == 0 || ncolors == 0 || npix == 0) return PB3D_OK;
    PB3D_REQUIRE(d_seg && colors && (n == 0 || (d_pts && d_cols)), "pb3d_project_iou_batch: null buffer");
    // cameras per pass: winner images of one pass stay below 512 MiB, the grid's y dimension below 65536
    i64 kc = (512ll << 20) / (npix * (i64)sizeof(u32));
    kc = kc < 1 ? 1 : (kc > 4096 ? 4096 : kc);
    if (kc > ncams) kc = ncams;
    void *winners, *dcams, *counts, *dcolors;
    PB3D_TRY(pb3d_scratch(ctx, 8, (size_t)(kc * npix) * sizeof(u32), &winners));
    PB3D_TRY(pb3d_scratch(ctx, 20, (size_t)kc * sizeof(ProjParams), &dcams));
    PB3D_TRY(pb3d_scratch(ctx, 21, (size_t)kc * 64 * sizeof(unsigned long long), &counts));
    PB3D_TRY(pb3d_scratch(ctx, 22, 128, &dcolors));
    PB3D_HIP(hipMemcpyAsync(dcolors, colors, (size_t)3 * ncolors, hipMemcpyHostToDevice, ctx->stream));
    ProjParams* hp = (ProjParams*)malloc((size_t)kc * sizeof(ProjParams));
    unsigned long long* hc = (unsigned long long*)malloc((size_t)kc * 64 * sizeof(unsigned long long));
    if (!hp || !hc) { free(hp); free(hc); pb3d_set_error("pb3d_project_iou_batch: out of host memory"); return PB3D_ENOMEM; }
    int rc = PB3D_OK;
    for (i64 k0 = 0; k0 < ncams && rc == PB3D_OK; k0 += kc) {
        const i64 kn = ncams - k0 < kc ? ncams - k0 : kc;
        bool all_f32 = true;
        ProjF32* hf = (ProjF32*)hp;                      // the float32 records are smaller: they share the staging area
        for (i64 k = 0; k < kn && rc == PB3D_OK; ++k) {
            const pb3d_camera* c = cams + k0 + k;
            ProjParams P;
            rc = fill_proj(&P, pts_f64, c->R, c->cam, c->f, c->cx, c->cy, c->prec, Himg, Wimg);
            ProjF32 F;
            if (rc == PB3D_OK && !f32_path(P, &F)) all_f32 = false;
        }
        if (rc != PB3D_OK) break;
        for (i64 k = 0; k < kn; ++k) {
            const pb3d_camera* c = cams + k0 + k;
            ProjParams P;
            fill_proj(&P, pts_f64, c->R, c->cam, c->f, c->cx, c->cy, c->prec, Himg, Wimg);
            if (all_f32) f32_path(P, &hf[k]); else hp[k] = P;
        }
        auto hipok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && rc == PB3D_OK) { pb3d_set_error("%s failed: %s", what, hipGetErrorString(e)); rc = PB3D_ENODEVICE; }
        };
        hipok(hipMemcpyAsync(dcams, hp, (size_t)kn * (all_f32 ? sizeof(ProjF32) : sizeof(ProjParams)), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
        hipok(hipMemsetAsync(winners, 0, (size_t)(kn * npix) * sizeof(u32), ctx->stream), "hipMemsetAsync");
        hipok(hipMemsetAsync(counts, 0, (size_t)kn * 64 * sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
        if (rc != PB3D_OK) break;
        if (n > 0) {
            // enough blocks per camera to fill the chip between them, never more than the points need
            i64 per_cam = all_f32 ? (n / 4 + 256) / 256 : (n + 255) / 256;
            const i64 want = ((i64)ctx->cus * 16 + kn - 1) / kn;
            if (per_cam > want) per_cam = want;
            if (per_cam < 1) per_cam = 1;
            if (all_f32)
                hipLaunchKernelGGL(k_project_points_batch_f32, dim3((unsigned)per_cam, (unsigned)kn), dim3(256), 0, ctx->stream, (const float*)d_pts, n,
                                   (const ProjF32*)dcams, vec_ok(d_pts), npix, (u32*)winners);
            else
                hipLaunchKernelGGL(k_project_points_batch, dim3((unsigned)per_cam, (unsigned)kn), dim3(256), 0, ctx->stream, d_pts, n,
                                   (const ProjParams*)dcams, npix, (u32*)winners);
            hipok(hipGetLastError(), "k_project_points_batch");
        }
        i64 ib = (npix + 255) / 256;
        const i64 iwant = ((i64)ctx->cus * 8 + kn - 1) / kn;
        if (ib > iwant) ib = iwant;
        if (ib < 1) ib = 1;
        hipLaunchKernelGGL(k_iou_winners_batch, dim3((unsigned)ib, (unsigned)kn), dim3(256), 0, ctx->stream, (const u32*)winners, d_cols, d_seg, npix,
                           ncolors, (const u8*)dcolors, (unsigned long long*)counts);
        hipok(hipGetLastError(), "k_iou_winners_batch");
        hipok(hipMemcpyAsync(hc, counts, (size_t)kn * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
        ++ctx->sync_count; hipok(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        if (rc != PB3D_OK) break;
        for (i64 k = 0; k < kn; ++k)
            for (int c = 0; c < ncolors; ++c) {
                inter[(k0 + k) * ncolors + c] = (int64_t)hc[k * 64 + 2 * c];
                uni[(k0 + k) * ncolors + c] = (int64_t)hc[k * 64 + 2 * c + 1];
            }
    }
    free(hp); free(hc);
    return rc;
}

}  // extern "C"
