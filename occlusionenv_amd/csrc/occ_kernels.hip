// occ_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the batched OcclusionEnv step().
//
// What the reference does per environment through PyTorch3D (4 renders, ~200 autograd nodes,
// /root/reference/environment.py:352-396) is done here for N environments in a fixed handful of launches:
//
//   occ_camera_kernel   one thread per env : action -> (el, az) -> C -> look_at R,T, carrying
//                       forward-mode tangents d/d(el,az) of R and T (environment.py:356-368)
//   occ_setup_kernel    one block per (env, object): gather pool verts, world->view->NDC
//                       (MeshRasterizer.transform), z-clip (clip_faces), cull, per-face record
//                       with NDC verts + tangents + invariants, ORDERED compaction (face order is
//                       PyTorch3D's tie-break order), packed tile bbox, object tile rect
//   occ_scan_kernel     prefix sum of the rect areas -> work items (env, object, block inside the rect)
//   occ_raster_kernel   persistent wave64 per work item (env, object, 4x4-pixel block): 16 pixels x 4 face
//                       slots per wave (on-the-fly binning by ballot over pixel bboxes, records staged in
//                       LDS by cooperative 16-B loads): soft silhouette (K nearest-z
//                       sigmoid product) + hard nearest face of ONE object in one sweep, candidate K-buffer
//                       (z, 1-p, grad) streamed to HBM/L2 in 1-KiB coalesced rows, exact top-K-by-z
//                       selection (LDS-histogram radix select) when a pixel has more than K candidates
//   occ_combine_kernel  one thread per pixel: occlusion image of the three silhouettes, loss and
//                       d loss/d(el,az) partials, nearest-object pick + flat shading, outputs
//   occ_reduce_kernel   one wave per env: fixed-order sum of the per-block partials
//   occ_finish_kernel   reward bookkeeping + action Jacobian (environment.py:381-392,356-361)
//
// The gradient is carried in FORWARD mode (two tangent directions, el and az) through the very
// same sweep that renders: d alpha/d theta = -(A/sigma) * sum_k p_k * d dist_k/d theta needs no
// second pass over the K-buffer, no atomics into grad_face_verts and no saved fragments.  It
// equals what autograd + _C.rasterize_meshes_backward produce.
//
// No MFMA: the path is rasterisation.  fp32 throughout.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "occ_constants.h"
#include "occlusionenv_amd.h"

namespace occ {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float clamp01(float t) { return fminf(fmaxf(t, 0.0f), 1.0f); }

// Read-only data produced by an EARLIER launch (face records, bboxes, rects) is read through the
// constant address space: with a wave-uniform address hipcc then emits s_load_dwordx8/x16 into
// SGPRs (scalar cache) instead of 64 redundant vector loads.
typedef const __attribute__((address_space(4))) float* cfptr;
typedef const __attribute__((address_space(4))) int* ciptr;
__device__ __forceinline__ cfptr as_const(const float* p) { return (cfptr)(uintptr_t)p; }
__device__ __forceinline__ ciptr as_const(const int* p) { return (ciptr)(uintptr_t)p; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// camera: dual numbers with two tangent directions (d/d el, d/d az)
// ------------------------------------------------------------------------------------------
struct D2 {
    float v, e, a;
};
__device__ __forceinline__ D2 dconst(float v) { return {v, 0.f, 0.f}; }
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return {x.v + y.v, x.e + y.e, x.a + y.a}; }
__device__ __forceinline__ D2 operator-(D2 x, D2 y) { return {x.v - y.v, x.e - y.e, x.a - y.a}; }
__device__ __forceinline__ D2 operator-(D2 x) { return {-x.v, -x.e, -x.a}; }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) {
    return {x.v * y.v, x.e * y.v + x.v * y.e, x.a * y.v + x.v * y.a};
}
__device__ __forceinline__ D2 operator/(D2 x, D2 y) {
    const float q = x.v / y.v;
    return {q, (x.e - q * y.e) / y.v, (x.a - q * y.a) / y.v};
}
__device__ __forceinline__ D2 dsin(D2 x) {
    const float s = sinf(x.v), c = cosf(x.v);
    return {s, c * x.e, c * x.a};
}
__device__ __forceinline__ D2 dcos(D2 x) {
    const float s = sinf(x.v), c = cosf(x.v);
    return {c, -s * x.e, -s * x.a};
}
// F.normalize(v, eps): v / max(||v||, eps)    [P3D look_at_rotation, SURVEY A.1]
__device__ __forceinline__ void dnormalize3(D2* v, float eps) {
    const D2 n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float n = sqrtf(n2.v);
    D2 nn;
    if (n > eps) {
        const float h = 0.5f / n;
        nn = {n, h * n2.e, h * n2.a};
    } else {
        nn = dconst(eps);
    }
    v[0] = v[0] / nn;
    v[1] = v[1] / nn;
    v[2] = v[2] / nn;
}
__device__ __forceinline__ void dcross(const D2* a, const D2* b, D2* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(64) void occ_camera_kernel(int mode, const float* __restrict__ action,
                                                        float* __restrict__ el_io, float* __restrict__ az_io,
                                                        const float* __restrict__ radius, float* __restrict__ cam,
                                                        float* __restrict__ cam_pos_out, int n_env) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_env) return;
    float* c = cam + (size_t)n * OCC_CAM_STRIDE;
    D2 C[3];
    float J[4] = {0.f, 0.f, 0.f, 0.f};
    float el_new = 0.f, az_new = 0.f;
    if (mode == OCC_CAM_STEP) {
        // environment.py:356-365
        const float a0 = action[2 * n], a1 = action[2 * n + 1];
        const float nrm = sqrtf(a0 * a0 + a1 * a1);
        float n0 = a0, n1 = a1;
        float j00 = 1.f, j01 = 0.f, j10 = 0.f, j11 = 1.f;  // d n_i / d a_j
        if (nrm != 0.0f) {
            n0 = a0 / nrm;
            n1 = a1 / nrm;
            j00 = (1.f - n0 * n0) / nrm;
            j01 = (-n0 * n1) / nrm;
            j10 = j01;
            j11 = (1.f - n1 * n1) / nrm;
        }
        el_new = el_io[n] + n0 * kStepSize;
        az_new = az_io[n] + n1 * kStepSize;
        el_io[n] = el_new;
        az_io[n] = az_new;
        J[0] = kStepSize * j00;
        J[1] = kStepSize * j01;
        J[2] = kStepSize * j10;
        J[3] = kStepSize * j11;
        const D2 el = {el_new, 1.f, 0.f}, az = {az_new, 0.f, 1.f};
        const D2 r = dconst(radius[n]);
        const D2 rs = r * dsin(az);
        C[0] = rs * dcos(el);
        C[1] = rs * dsin(el);
        C[2] = r * dcos(az);
    } else if (mode == OCC_CAM_LOOKAT) {
        // environment.py:308 -> [P3D] camera_position_from_spherical_angles(degrees=False)
        el_new = el_io[n];
        az_new = az_io[n];
        const float r = radius[n];
        C[0] = dconst(r * cosf(el_new) * sinf(az_new));
        C[1] = dconst(r * sinf(el_new));
        C[2] = dconst(r * cosf(el_new) * cosf(az_new));
    } else {
        C[0] = dconst(action[3 * n]);
        C[1] = dconst(action[3 * n + 1]);
        C[2] = dconst(action[3 * n + 2]);
    }
    // [P3D] look_at_rotation(C, at=0, up=+Y)
    D2 z[3] = {-C[0], -C[1], -C[2]};
    dnormalize3(z, kLookAtEps);
    const D2 up[3] = {dconst(0.f), dconst(1.f), dconst(0.f)};
    D2 x[3], y[3];
    dcross(up, z, x);
    dnormalize3(x, kLookAtEps);
    dcross(z, x, y);
    dnormalize3(y, kLookAtEps);
    if (fabsf(x[0].v) <= kLookAtClose && fabsf(x[1].v) <= kLookAtClose && fabsf(x[2].v) <= kLookAtClose) {
        dcross(y, z, x);
        dnormalize3(x, kLookAtEps);
    }
    // R[i][j]: columns are x, y, z ; T = -R^T C
    const D2* ax[3] = {x, y, z};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const D2 t = -(ax[j][0] * C[0] + ax[j][1] * C[1] + ax[j][2] * C[2]);
        c[C_T + j] = t.v;
        c[C_DT_EL + j] = t.e;
        c[C_DT_AZ + j] = t.a;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c[C_R + i * 3 + j] = ax[j][i].v;
            c[C_DR_EL + i * 3 + j] = ax[j][i].e;
            c[C_DR_AZ + i * 3 + j] = ax[j][i].a;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) c[C_C + i] = C[i].v;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[C_J + i] = J[i];
    c[C_EL] = el_new;
    c[C_AZ] = az_new;
    c[45] = c[46] = c[47] = 0.f;
    if (cam_pos_out) {
        cam_pos_out[3 * n] = C[0].v;
        cam_pos_out[3 * n + 1] = C[1].v;
        cam_pos_out[3 * n + 2] = C[2].v;
    }
}

// ------------------------------------------------------------------------------------------
// setup: projection, z-clipping, culling, record build with ordered compaction
// ------------------------------------------------------------------------------------------
struct PVert {      // one projected vertex
    float x, y, z;  // x_ndc, y_ndc, z_view
    float t[4];     // d x/d el, d y/d el, d x/d az, d y/d az
};

struct VVert {  // view-space vertex with tangents
    float v[3];
    float de[3], da[3];
};

template <bool GRAD>
__device__ __forceinline__ PVert project(const VVert& q) {
    // [P3D] x_ndc = x_view * s / z_view  
    PVert p;
    const float iz = 1.0f / q.v[2];
    p.x = q.v[0] * kProjScale * iz;
    p.y = q.v[1] * kProjScale * iz;
    p.z = q.v[2];
    if (GRAD) {
        p.t[0] = (kProjScale * q.de[0] - p.x * q.de[2]) * iz;
        p.t[1] = (kProjScale * q.de[1] - p.y * q.de[2]) * iz;
        p.t[2] = (kProjScale * q.da[0] - p.x * q.da[2]) * iz;
        p.t[3] = (kProjScale * q.da[1] - p.y * q.da[2]) * iz;
    } else {
        p.t[0] = p.t[1] = p.t[2] = p.t[3] = 0.f;
    }
    return p;
}

// [P3D] clip_faces: intersection of edge (a -> b) with z = kZClip, weight detached
template <bool GRAD>
__device__ __forceinline__ PVert cut_edge(const VVert& a, const VVert& b) {
    PVert p;
    const float w = (a.v[2] - kZClip) / (a.v[2] - b.v[2]);
    const float iw = 1.0f - w;
    const float ic = 1.0f / kZClip;
    p.z = a.v[2] * iw + b.v[2] * w;
    p.x = (kProjScale * a.v[0] * iw + kProjScale * b.v[0] * w) * ic;
    p.y = (kProjScale * a.v[1] * iw + kProjScale * b.v[1] * w) * ic;
    if (GRAD) {
        p.t[0] = (kProjScale * a.de[0] * iw + kProjScale * b.de[0] * w) * ic;
        p.t[1] = (kProjScale * a.de[1] * iw + kProjScale * b.de[1] * w) * ic;
        p.t[2] = (kProjScale * a.da[0] * iw + kProjScale * b.da[0] * w) * ic;
        p.t[3] = (kProjScale * a.da[1] * iw + kProjScale * b.da[1] * w) * ic;
    } else {
        p.t[0] = p.t[1] = p.t[2] = p.t[3] = 0.f;
    }
    return p;
}

struct Tri {
    PVert v[3];
    uint4 bbox;  // conservative pixel bbox x = xl | yl << 16, y = xh | yh << 16; z = key of the smallest vertex depth
    int tx0, ty0, tx1, ty1;
};

// Returns false if the triangle can never be matched to a pixel (culled / degenerate / off screen).  EVERY field
// is filled with in-range values either way: occ_setup_kernel evaluates a surviving face twice (once to count it,
// once to write it) and the two inlined copies need not round alike (fp contraction), so the second evaluation
// must be safe to use even where it would, by a hair, have decided differently.
__device__ __forceinline__ bool finish_tri(Tri& t, int S) {
    const float x0 = t.v[0].x, y0 = t.v[0].y, x1 = t.v[1].x, y1 = t.v[1].y, x2 = t.v[2].x, y2 = t.v[2].y;
    // [P3D] face_area = EdgeFunction(v0; v1, v2); back faces are culled (environment.py:253,271)
    const float area = (x0 - x1) * (y2 - y1) - (y0 - y1) * (x2 - x1);
    bool vis = area > kEpsilon;  // false for a back face, zero area or NaN
    vis = vis && !(fmax3(t.v[0].z, t.v[1].z, t.v[2].z) < 0.0f);
    const float bx0 = fmin3(x0, x1, x2) - kSqrtBlur, bx1 = fmax3(x0, x1, x2) + kSqrtBlur;
    const float by0 = fmin3(y0, y1, y2) - kSqrtBlur, by1 = fmax3(y0, y1, y2) + kSqrtBlur;
    const float lim = 1.0f - 1.0f / (float)S;  // outermost pixel centre
    vis = vis && !(bx1 < -lim || bx0 > lim || by1 < -lim || by0 > lim);
    // pixel index of an NDC coordinate: u(f) = (S-1) - ((f+1)*S - 1)/2   (decreasing)
    const float fS = (float)S;
    auto u = [&](float f) { return (fS - 1.0f) - ((f + 1.0f) * fS - 1.0f) * 0.5f; };
    // pixels whose centre can pass the exact float test bx0 <= xf <= bx1 (u is decreasing); 1e-3 px of slack
    // covers the rounding of u() - the per-pixel float test in eval_face stays the authority
    int xl = (int)ceilf(u(bx1) - 1e-3f), xh = (int)floorf(u(bx0) + 1e-3f);
    int yl = (int)ceilf(u(by1) - 1e-3f), yh = (int)floorf(u(by0) + 1e-3f);
    vis = vis && (max(xl, 0) <= min(xh, S - 1)) && (max(yl, 0) <= min(yh, S - 1));
    xl = min(max(xl, 0), S - 1);
    yl = min(max(yl, 0), S - 1);
    xh = min(max(xh, xl), S - 1);
    yh = min(max(yh, yl), S - 1);
    t.tx0 = xl / OCC_BLOCK;
    t.tx1 = xh / OCC_BLOCK;
    t.ty0 = yl / OCC_BLOCK;
    t.ty1 = yh / OCC_BLOCK;
    const uint32_t zb = __float_as_uint(fmin3(t.v[0].z, t.v[1].z, t.v[2].z));
    t.bbox = make_uint4((uint32_t)xl | ((uint32_t)yl << 16), (uint32_t)xh | ((uint32_t)yh << 16),
                        (zb & 0x80000000u) ? ~zb : (zb | 0x80000000u), 0u);
    return vis;
}

// [P3D] HardFlatShader terms of ONE face: flat shading uses the face normal and the face centre only,
// so (ambient + diffuse) and the specular term are per-face constants of the current camera.  w0..w2 = the
// ORIGINAL face's world-space corners (also for z-clipped pieces), cpos = camera centre.  Computed once per
// visible face by the setup kernel; the combine kernel then shades a pixel with one gather.
struct Shade {
    float amb_diff, spec;
};
__device__ __forceinline__ Shade flat_shade(const float* w0, const float* w1, const float* w2, float cx, float cy, float cz) {
    // hardware sqrt / rcp (1 ulp) instead of the IEEE sequences: ~1e-7 relative on a colour in [0.5, 1]
    auto inv_len = [](float x, float y, float z) { return frcp(fmaxf(__builtin_amdgcn_sqrtf(x * x + y * y + z * z), kShadeEps)); };
    const float ax = w1[0] - w0[0], ay = w1[1] - w0[1], az = w1[2] - w0[2];
    const float bx = w2[0] - w0[0], by = w2[1] - w0[1], bz = w2[2] - w0[2];
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    float in_ = inv_len(nx, ny, nz);
    nx *= in_; ny *= in_; nz *= in_;
    in_ = inv_len(nx, ny, nz);  // F.normalize again in diffuse()/specular()
    nx *= in_; ny *= in_; nz *= in_;
    const float third = 1.0f / 3.0f;
    const float ccx = (w0[0] + w1[0] + w2[0]) * third, ccy = (w0[1] + w1[1] + w2[1]) * third,
                ccz = (w0[2] + w1[2] + w2[2]) * third;
    float lx = kLightX - ccx, ly = kLightY - ccy, lz = kLightZ - ccz;
    const float il = inv_len(lx, ly, lz);
    lx *= il; ly *= il; lz *= il;
    const float cosang = nx * lx + ny * ly + nz * lz;
    const float diffuse = kDiffuse * fmaxf(cosang, 0.f);
    float vx = cx - ccx, vy = cy - ccy, vz = cz - ccz;
    const float iv = inv_len(vx, vy, vz);
    vx *= iv; vy *= iv; vz *= iv;
    const float rx = -lx + 2.f * (cosang * nx), ry = -ly + 2.f * (cosang * ny), rz = -lz + 2.f * (cosang * nz);
    float sa = fmaxf(vx * rx + vy * ry + vz * rz, 0.f) * (cosang > 0.f ? 1.f : 0.f);
    sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa;  // ^64
    Shade sh;
    sh.amb_diff = kAmbient + diffuse;
    sh.spec = kSpecular * sa;
    return sh;
}

template <bool GRAD>
__device__ __forceinline__ void write_record(float* __restrict__ r, uint4* __restrict__ bb, uint4* __restrict__ scan_row,
                                             int pos, const Tri& t, int face_id, int flags, Shade sh) {
    const float x0 = t.v[0].x, y0 = t.v[0].y, x1 = t.v[1].x, y1 = t.v[1].y, x2 = t.v[2].x, y2 = t.v[2].y;
    // [P3D] BarycentricCoordsForward: area = EdgeFunction(v2; v0, v1) + kEpsilon
    const float area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0) + kEpsilon;
    const float l01 = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
    const float l02 = (x2 - x0) * (x2 - x0) + (y2 - y0) * (y2 - y0);
    const float l12 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
    float4* r4 = reinterpret_cast<float4*>(r);
    // slot map: occ_constants.h (R_X0 .. R_TAN)
    r4[0] = make_float4(x0, y0, t.v[0].z, x1);
    r4[1] = make_float4(y1, t.v[1].z, x2, y2);
    r4[2] = make_float4(t.v[2].z, __int_as_float(face_id), __int_as_float(flags), 1.0f / area);
    r4[3] = make_float4(fmin3(x0, x1, x2) - kSqrtBlur, fmax3(x0, x1, x2) + kSqrtBlur, fmin3(y0, y1, y2) - kSqrtBlur,
                        fmax3(y0, y1, y2) + kSqrtBlur);
    r4[4] = make_float4(l01 <= kEpsilon ? -1.0f : 1.0f / l01, l02 <= kEpsilon ? -1.0f : 1.0f / l02,
                        l12 <= kEpsilon ? -1.0f : 1.0f / l12, sh.spec);
    if (GRAD) {
        r4[5] = make_float4(t.v[0].t[0], t.v[0].t[1], t.v[0].t[2], t.v[0].t[3]);
        r4[6] = make_float4(t.v[1].t[0], t.v[1].t[1], t.v[1].t[2], t.v[1].t[3]);
        r4[7] = make_float4(t.v[2].t[0], t.v[2].t[1], t.v[2].t[2], t.v[2].t[3]);
    }
    *bb = make_uint4(t.bbox.x, t.bbox.y, t.bbox.z, __float_as_uint(sh.amb_diff));  // .w: ambient + diffuse of the face
    // scan row in face order (occ_sort_kernel re-orders dense objects): (pixel bbox, nearest depth key, record index)
    *scan_row = make_uint4(t.bbox.x, t.bbox.y, t.bbox.z, (uint32_t)pos);
}

// union pixel bbox and smallest depth key of every 64-entry chunk of the scan order (two-level scan)
__device__ __forceinline__ void chunk_boxes(const uint4* __restrict__ scan, uint4* __restrict__ cbx, int nr, int wave,
                                            int lane) {
    const int nch = (nr + 63) >> 6;
    for (int c = wave; c < nch; c += 4) {
        const int j = c * 64 + lane;
        uint4 bb = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
        if (j < nr) bb = scan[j];
        int xl = bb.x & 0xFFFF, yl = bb.x >> 16, xh = bb.y & 0xFFFF, yh = bb.y >> 16;
        uint32_t zk = bb.z;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            xl = min(xl, __shfl_xor(xl, m, 64));
            yl = min(yl, __shfl_xor(yl, m, 64));
            xh = max(xh, __shfl_xor(xh, m, 64));
            yh = max(yh, __shfl_xor(yh, m, 64));
            zk = min(zk, (uint32_t)__shfl_xor((int)zk, m, 64));
        }
        if (lane == 0)
            cbx[c] = make_uint4((uint32_t)xl | ((uint32_t)yl << 16), (uint32_t)xh | ((uint32_t)yh << 16), zk, 0u);
    }
}

// Objects with many visible faces (>= kSortMin records: a pixel then collects far more than K candidates) get
// their scan order sorted front to back - bitonic sort of (depth key, record index) in LDS - so that the raster
// kernel reaches "every pixel of the block holds its K nearest" after the nearest faces and skips the rest.
constexpr int kSortMin = 4096;
__global__ __launch_bounds__(256) void occ_sort_kernel(OccScene sc, OccWorkspace ws, int sort_cap) {
    extern __shared__ unsigned long long s_keys[];  // sort_cap keys: depth key << 32 | record index
    const int eo = blockIdx.x;
    const int nr = ws.nrec[eo];
    if (nr < kSortMin) return;
    int p2 = 1;
    while (p2 < nr) p2 <<= 1;
    if (p2 > sort_cap) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4* __restrict__ bbs = reinterpret_cast<const uint4*>(ws.rec_bbox) + (size_t)eo * sc.rec_cap;
    uint4* __restrict__ scan = reinterpret_cast<uint4*>(ws.scan) + (size_t)eo * sc.rec_cap;
    for (int i = tid; i < p2; i += 256) s_keys[i] = i < nr ? (((unsigned long long)bbs[i].z << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[ixj];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        s_keys[i] = b;
                        s_keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < nr; i += 256) {
        const int j = (int)(s_keys[i] & 0xFFFFFFFFu);
        const uint4 bb = bbs[j];
        scan[i] = make_uint4(bb.x, bb.y, bb.z, (uint32_t)j);
    }
    __syncthreads();
    chunk_boxes(scan, reinterpret_cast<uint4*>(ws.rec_cbox) + (size_t)eo * ((sc.rec_cap + 63) >> 6), nr, wave, lane);
}

struct CamRT {
    float R[9], T[3], dRe[9], dTe[3], dRa[9], dTa[3];
};

// world-space vertex k of face f: pool vertex + object offset in f32 (environment.py:148,171)
__device__ __forceinline__ void world_vertex(const int* __restrict__ pool_faces, const float* __restrict__ pool_verts,
                                             int vo, int fo, int f, int k, float ox, float oy, float oz, float* w) {
    const int vi = pool_faces[(size_t)(fo + f) * 3 + k];
    const float* pv = pool_verts + (size_t)(vo + vi) * 3;
    w[0] = pv[0] + ox;
    w[1] = pv[1] + oy;
    w[2] = pv[2] + oz;
}

__device__ __forceinline__ void world_corner(const float* __restrict__ pool_verts, int vo, int vi, float ox, float oy,
                                             float oz, float* w) {
    const float* pv = pool_verts + (size_t)(vo + vi) * 3;
    w[0] = pv[0] + ox;
    w[1] = pv[1] + oy;
    w[2] = pv[2] + oz;
}

template <bool GRAD>
__device__ __forceinline__ void view_from_world(const CamRT& c, const float* w, VVert& q) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        q.v[j] = w[0] * c.R[j] + w[1] * c.R[3 + j] + w[2] * c.R[6 + j] + c.T[j];
        if (GRAD) {
            q.de[j] = w[0] * c.dRe[j] + w[1] * c.dRe[3 + j] + w[2] * c.dRe[6 + j] + c.dTe[j];
            q.da[j] = w[0] * c.dRa[j] + w[1] * c.dRa[3 + j] + w[2] * c.dRa[6 + j] + c.dTa[j];
        }
    }
}

template <bool GRAD>
__device__ __forceinline__ void load_camera(const float* __restrict__ c, CamRT& C) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        C.R[i] = c[C_R + i];
        C.dRe[i] = GRAD ? c[C_DR_EL + i] : 0.f;
        C.dRa[i] = GRAD ? c[C_DR_AZ + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        C.T[i] = c[C_T + i];
        C.dTe[i] = GRAD ? c[C_DT_EL + i] : 0.f;
        C.dTa[i] = GRAD ? c[C_DT_AZ + i] : 0.f;
    }
}

// Faces that straddle z = kZClip ([P3D] clip_faces cases 3 and 4).  Rare (the camera must be within
// ~0.5 of the geometry), so this lives out of line: it re-derives everything from the face index, both when the
// face is counted and when its records are written, and keeps its dynamically indexed arrays off the hot path.
// Everything arrives by value (the camera is re-read from memory) so that nothing of the caller's state has its
// address taken - that would pin the kernel arguments and the camera in scratch for the fast path as well.
template <bool GRAD>
__device__ __attribute__((noinline)) int clip_face_slow(const int* __restrict__ pool_faces, const float* __restrict__ pool_verts,
                                                        const float* __restrict__ camp, int S, int vo, int fo, int f,
                                                        float ox, float oy, float oz, Tri* out, int* flags) {
    CamRT c;
    load_camera<GRAD>(camp, c);
    VVert q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w[3];
        world_vertex(pool_faces, pool_verts, vo, fo, f, k, ox, oy, oz, w);
        view_from_world<GRAD>(c, w, q[k]);
    }
    const bool b0 = q[0].v[2] < kZClip, b1 = q[1].v[2] < kZClip, b2 = q[2].v[2] < kZClip;
    const int nb = (int)b0 + (int)b1 + (int)b2;
    flags[0] = flags[1] = FLAG_CLIPPED;
    if (nb == 2) {
        // case 3: p1 = the vertex in front; new triangle (p4, p5, p1)
        const int i1 = !b0 ? 0 : (!b1 ? 1 : 2);
        const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
        out[0].v[0] = cut_edge<GRAD>(q[i1], q[i2]);
        out[0].v[1] = cut_edge<GRAD>(q[i1], q[i3]);
        out[0].v[2] = project<GRAD>(q[i1]);
        return finish_tri(out[0], S) ? 1 : 0;
    }
    if (nb == 1) {
        // case 4: p1 = the vertex behind; quad -> (p4, p2, p5), (p5, p2, p3)
        const int i1 = b0 ? 0 : (b1 ? 1 : 2);
        const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
        const PVert p4 = cut_edge<GRAD>(q[i1], q[i2]);
        const PVert p5 = cut_edge<GRAD>(q[i1], q[i3]);
        const PVert p2 = project<GRAD>(q[i2]);
        const PVert p3 = project<GRAD>(q[i3]);
        Tri ta, tb;
        ta.v[0] = p4; ta.v[1] = p2; ta.v[2] = p5;
        tb.v[0] = p5; tb.v[1] = p2; tb.v[2] = p3;
        const bool oka = finish_tri(ta, S), okb = finish_tri(tb, S);
        if (oka && okb) {
            out[0] = ta; out[1] = tb;
            flags[0] = FLAG_PAIR_FIRST | FLAG_CLIPPED; flags[1] = FLAG_PAIR_SECOND | FLAG_CLIPPED;
            return 2;
        }
        if (oka) { out[0] = ta; return 1; }
        if (okb) { out[0] = tb; return 1; }
    }
    return 0;  // nb == 3: the whole face is behind the clip plane
}

template <bool GRAD>
__global__ __launch_bounds__(256, 4) void occ_setup_kernel(OccScene sc, const float* __restrict__ cam, OccWorkspace ws) {
    __shared__ int s_wcnt[2][4];  // double-buffered: one barrier per 256-face round
    __shared__ int s_rect[4];
    __shared__ float4 s_rec[4 * 64 * kRecPad];  // per wave: the records of one round, staged for coalesced stores
    // (LDS stride 9 parts = 36 dwords: a 32-dword stride would put every lane's write on the same banks)
    const int eo = blockIdx.x;  // env*3 + object
    const int env = eo / 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (sc.skip && sc.skip[env]) {  // scene row not rendered in this launch: no records, no work items
        if (tid == 0) {
            ws.nrec[eo] = 0;
            ws.objrect[eo * 4 + 0] = 1 << 20;
            ws.objrect[eo * 4 + 1] = 1 << 20;
            ws.objrect[eo * 4 + 2] = -1;
            ws.objrect[eo * 4 + 3] = -1;
        }
        return;
    }
    const int mesh = sc.scene_mesh[eo];
    const int vo = sc.mesh_vert_off[mesh];
    const int fo = sc.mesh_face_off[mesh];
    const int nF = sc.mesh_face_off[mesh + 1] - fo;
    const float ox = sc.scene_offset[eo * 3], oy = sc.scene_offset[eo * 3 + 1], oz = sc.scene_offset[eo * 3 + 2];
    const float* __restrict__ c = cam + (size_t)env * OCC_CAM_STRIDE;
    CamRT C;
    load_camera<GRAD>(c, C);
    const int* __restrict__ pool_faces = sc.pool_faces;
    const float* __restrict__ pool_verts = sc.pool_verts;
    const int S = sc.img, rec_cap = sc.rec_cap;
    if (tid == 0) {
        s_rect[0] = 1 << 20;
        s_rect[1] = 1 << 20;
        s_rect[2] = -1;
        s_rect[3] = -1;
    }
    __syncthreads();
    float* __restrict__ rec = ws.rec + (size_t)eo * rec_cap * OCC_REC_STRIDE;
    uint4* __restrict__ bbs = reinterpret_cast<uint4*>(ws.rec_bbox) + (size_t)eo * rec_cap;
    uint4* __restrict__ scan = reinterpret_cast<uint4*>(ws.scan) + (size_t)eo * rec_cap;
    int total = 0;
    bool overflow = false;
    int rx0 = 1 << 20, ry0 = 1 << 20, rx1 = -1, ry1 = -1;  // this thread's share of the object's block rect
    int round = 0;
    // vertex indices of the NEXT round's face are fetched one round ahead: the index -> vertex -> projection chain
    // of a round then starts at the vertex gather
    int vi0 = 0, vi1 = 0, vi2 = 0;
    if (tid < nF) {
        const int* pf = pool_faces + (size_t)(fo + tid) * 3;
        vi0 = pf[0]; vi1 = pf[1]; vi2 = pf[2];
    }
    for (int base = 0; base < nF; base += 256, round ^= 1) {
        const int f = base + tid;
        int cnt = 0;
        bool slow = false;
        float w0[3], w1[3], w2[3];  // world-space corners: all that a surviving face carries across the barrier
        const int c0 = vi0, c1 = vi1, c2 = vi2;
        if (f + 256 < nF) {
            const int* pf = pool_faces + (size_t)(fo + f + 256) * 3;
            vi0 = pf[0]; vi1 = pf[1]; vi2 = pf[2];
        }
        if (f < nF) {
            Tri tri;  // fast path: the unclipped face, positions only (recomputed for the survivors below)
            VVert q0, q1, q2;
            world_corner(pool_verts, vo, c0, ox, oy, oz, w0);
            world_corner(pool_verts, vo, c1, ox, oy, oz, w1);
            world_corner(pool_verts, vo, c2, ox, oy, oz, w2);
            view_from_world<false>(C, w0, q0);
            view_from_world<false>(C, w1, q1);
            view_from_world<false>(C, w2, q2);
            slow = (q0.v[2] < kZClip) || (q1.v[2] < kZClip) || (q2.v[2] < kZClip);
            if (!slow) {
                tri.v[0] = project<false>(q0);
                tri.v[1] = project<false>(q1);
                tri.v[2] = project<false>(q2);
                cnt = finish_tri(tri, S) ? 1 : 0;
            }
        }
        if (__ballot(slow)) {
            if (slow) {
                Tri tmp[2];
                int fl[2];
                cnt = clip_face_slow<GRAD>(pool_faces, pool_verts, c, S, vo, fo, f, ox, oy, oz, tmp, fl);
            }
        }
        // ordered compaction: exclusive prefix of cnt in {0,1,2} over the block
        const unsigned long long m1 = __ballot(cnt >= 1), m2 = __ballot(cnt == 2);
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int pre = __popcll(m1 & lt) + __popcll(m2 & lt);
        if (lane == 0) s_wcnt[round][wave] = __popcll(m1) + __popcll(m2);
        __syncthreads();
        int woff = 0, itot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int cw = s_wcnt[round][w];
            if (w < wave) woff += cw;
            itot += cw;
        }
        const int pos = total + woff + pre;
        // Records of a wave are consecutive (ordered compaction): the survivors put theirs into LDS and the wave
        // copies the block out with full-width 16-byte stores (a lane writing its own 128-byte record straight to
        // memory issues eight partial-line stores).  Waves with a z-clipped face, or at the
        // capacity limit, store directly.
        const int wstart = total + woff, nw = __popcll(m1) + __popcll(m2);
        const bool staged = (__ballot(slow) == 0ull) && (wstart + nw <= rec_cap);
        if (cnt >= 1) {
            if (pos + cnt <= rec_cap) {
                int x0, y0, x1, y1;
                const Shade sh = flat_shade(w0, w1, w2, c[C_C], c[C_C + 1], c[C_C + 2]);
                if (!slow) {
                    Tri tri;
                    {
                        VVert q;
                        view_from_world<false>(C, w0, q);
                        tri.v[0] = project<false>(q);
                        view_from_world<false>(C, w1, q);
                        tri.v[1] = project<false>(q);
                        view_from_world<false>(C, w2, q);
                        tri.v[2] = project<false>(q);
                        finish_tri(tri, S);
                    }
                    auto emit = [&](float* __restrict__ r) {
                        write_record<false>(r, bbs + pos, scan + pos, pos, tri, f, 0, sh);
                        if (GRAD) {
                            // tangents only for the faces that survived culling, stored vertex by vertex
                            VVert q;
                            PVert pk;
                            view_from_world<true>(C, w0, q);
                            pk = project<true>(q);
                            reinterpret_cast<float4*>(r)[5] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
                            view_from_world<true>(C, w1, q);
                            pk = project<true>(q);
                            reinterpret_cast<float4*>(r)[6] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
                            view_from_world<true>(C, w2, q);
                            pk = project<true>(q);
                            reinterpret_cast<float4*>(r)[7] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
                        }
                    };
                    if (staged) {
                        emit(reinterpret_cast<float*>(&s_rec[(wave * 64 + pre) * kRecPad]));
                    } else {
                        emit(rec + (size_t)pos * OCC_REC_STRIDE);
                    }
                    x0 = tri.tx0; y0 = tri.ty0; x1 = tri.tx1; y1 = tri.ty1;
                } else {
                    Tri tmp[2];
                    int fl[2];
                    clip_face_slow<GRAD>(pool_faces, pool_verts, c, S, vo, fo, f, ox, oy, oz, tmp, fl);
                    write_record<GRAD>(rec + (size_t)pos * OCC_REC_STRIDE, bbs + pos, scan + pos, pos, tmp[0], f, fl[0], sh);
                    x0 = tmp[0].tx0; y0 = tmp[0].ty0; x1 = tmp[0].tx1; y1 = tmp[0].ty1;
                    if (cnt == 2) {
                        write_record<GRAD>(rec + (size_t)(pos + 1) * OCC_REC_STRIDE, bbs + pos + 1, scan + pos + 1, pos + 1,
                                           tmp[1], f, fl[1], sh);
                        x0 = min(x0, tmp[1].tx0); y0 = min(y0, tmp[1].ty0);
                        x1 = max(x1, tmp[1].tx1); y1 = max(y1, tmp[1].ty1);
                    }
                }
                rx0 = min(rx0, x0); ry0 = min(ry0, y0);
                rx1 = max(rx1, x1); ry1 = max(ry1, y1);
            } else {
                overflow = true;
            }
        }
        if (staged) {  // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int kP = GRAD ? kRecParts : 5;  // parts this variant writes
            float4* __restrict__ dst = reinterpret_cast<float4*>(rec + (size_t)wstart * OCC_REC_STRIDE);
            const float4* src = &s_rec[wave * 64 * kRecPad];
            for (int i = lane; i < nw * kP; i += 64) {
                const int rj = i / kP, part = i - rj * kP;
                dst[rj * kRecParts + part] = src[rj * kRecPad + part];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        total += itot;
    }
    if (overflow) atomicOr(&ws.status[env], OCC_STATUS_REC_OVERFLOW);
    // object block rect: wave reduction, then one LDS atomic per wave
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        rx0 = min(rx0, __shfl_xor(rx0, m, 64));
        ry0 = min(ry0, __shfl_xor(ry0, m, 64));
        rx1 = max(rx1, __shfl_xor(rx1, m, 64));
        ry1 = max(ry1, __shfl_xor(ry1, m, 64));
    }
    if (lane == 0) {
        atomicMin(&s_rect[0], rx0);
        atomicMin(&s_rect[1], ry0);
        atomicMax(&s_rect[2], rx1);
        atomicMax(&s_rect[3], ry1);
    }
    {
        // SCAN ORDER of the raster kernel: (pixel bbox, key of the nearest vertex depth, record index) rows, written
        // with the records in face order (mesh order is spatially coherent, which makes the chunk boxes selective);
        // occ_sort_kernel re-orders dense objects front to back.  The depth keys make the raster kernel's
        // pruning exact in ANY order; the order only decides how early it bites.
        const int nr = min(total, rec_cap);
        __syncthreads();  // scan[] of the whole object written (and s_rect complete)
        chunk_boxes(scan, reinterpret_cast<uint4*>(ws.rec_cbox) + (size_t)eo * ((rec_cap + 63) >> 6), nr, wave, lane);
    }
    if (tid == 0) {
        ws.nrec[eo] = min(total, rec_cap);
        ws.objrect[eo * 4 + 0] = s_rect[0];
        ws.objrect[eo * 4 + 1] = s_rect[1];
        ws.objrect[eo * 4 + 2] = s_rect[2];
        ws.objrect[eo * 4 + 3] = s_rect[3];
    }
}

// ------------------------------------------------------------------------------------------
// tile rasteriser
// ------------------------------------------------------------------------------------------
struct Cand {
    bool cand;    // soft candidate (inside, or within blur)
    bool inside;  // pixel centre strictly inside (hard candidate)
    float z;      // soft depth (clipped barycentrics)
    float zh;     // hard depth (unclipped barycentrics)
    float ad;     // |squared distance|
    int amin;     // closest edge: 0 = (v0,v1), 1 = (v0,v2), 2 = (v1,v2)
    float q;      // 1 - sigmoid(-d/sigma)
    float ge, ga; // p * d(d)/d el, p * d(d)/d az
};

// Evaluate one projected face (wave-uniform record r -> SGPRs) at this lane's pixel centre.
// Restates [P3D] CheckPixelInsideFace and, for GRAD, the dists part of
// RasterizeMeshesBackward (A.5) pushed forward along the two vertex tangents.
// One staged record pulled out of LDS with 16-byte broadcast reads (every lane reads the same address).
// Slot map as in occ_constants.h:
//   a = x0 y0 z0 x1 | b = y1 z1 x2 y2 | c = z2 id flags inv_area | d = bbox | e = il01 il02 il12 - |
//   g, h, i = tangents of v0, v1, v2 (dx/del dy/del dx/daz dy/daz)      -- 8 parts = 128 bytes = one cache line
// The eight float4 parts travel as SSA values (by value, never through a struct in memory: a select between two
// loads of one stack object gets folded into a dynamically indexed load, which pins the object in scratch).
#define OCC_REC_PARAMS float4 ra, float4 rb, float4 rc, float4 rd, float4 re, float4 rg, float4 rh, float4 ri
#define OCC_REC_LOAD(src, PARTS)                                                                   \
    (src)[0], (src)[1], (src)[2], (src)[3], ((PARTS) > 4 ? (src)[4] : make_float4(0, 0, 0, 0)),      \
        ((PARTS) > 5 ? (src)[5] : make_float4(0, 0, 0, 0)), ((PARTS) > 5 ? (src)[6] : make_float4(0, 0, 0, 0)), \
        ((PARTS) > 5 ? (src)[7] : make_float4(0, 0, 0, 0))

// Evaluate one projected face at this lane's pixel centre.
template <bool SOFT, bool GRAD>
__device__ __forceinline__ void eval_face(OCC_REC_PARAMS, float xf, float yf, Cand& c) {
    c.cand = false;
    c.inside = false;
    c.z = c.zh = c.ad = 0.f;
    c.q = 1.f;
    c.ge = c.ga = 0.f;
    c.amin = 0;
    const bool inb = (rd.x <= xf) && (xf <= rd.y) && (rd.z <= yf) && (yf <= rd.w);
    if (!inb) return;
    const float x0 = ra.x, y0 = ra.y, z0 = ra.z;
    const float x1 = ra.w, y1 = rb.x, z1 = rb.y;
    const float x2 = rb.z, y2 = rb.w, z2 = rc.x;
    const float dx0 = xf - x0, dy0 = yf - y0, dx1 = xf - x1, dy1 = yf - y1, dx2 = xf - x2, dy2 = yf - y2;
    const float ex01 = x1 - x0, ey01 = y1 - y0, ex02 = x2 - x0, ey02 = y2 - y0, ex12 = x2 - x1, ey12 = y2 - y1;
    const float inv_area = rc.w;
    // barycentrics: E(p;v1,v2), E(p;v2,v0), E(p;v0,v1) over area
    const float b0 = (dx1 * ey12 - dy1 * ex12) * inv_area;
    const float b1 = (dy2 * ex02 - dx2 * ey02) * inv_area;
    const float b2 = (dx0 * ey01 - dy0 * ex01) * inv_area;
    // perspective correction
    const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
    const float rden = frcp(fmaxf(w0 + w1 + w2, kEpsilon));
    const float p0 = w0 * rden, p1 = w1 * rden, p2 = w2 * rden;
    const bool inside = (p0 > 0.0f) && (p1 > 0.0f) && (p2 > 0.0f);
    c.zh = p0 * z0 + p1 * z1 + p2 * z2;
    c.inside = inside && !(c.zh < 0.0f);
    if (!SOFT) return;
    // clipped barycentrics -> soft depth
    float c0 = fmaxf(p0, 0.f), c1 = fmaxf(p1, 0.f), c2 = fmaxf(p2, 0.f);
    const float rs = frcp(fmaxf(c0 + c1 + c2, kBaryClipMin));
    c0 *= rs;
    c1 *= rs;
    c2 *= rs;
    const float pz = c0 * z0 + c1 * z1 + c2 * z2;
    // squared distance to the three edges (v0,v1), (v0,v2), (v1,v2)
    const float il01 = re.x, il02 = re.y, il12 = re.z;
    const float dot01 = ex01 * dx0 + ey01 * dy0;
    const float dot02 = ex02 * dx0 + ey02 * dy0;
    const float dot12 = ex12 * dx1 + ey12 * dy1;
    const float t01 = il01 < 0.f ? 1.0f : clamp01(dot01 * il01);
    const float t02 = il02 < 0.f ? 1.0f : clamp01(dot02 * il02);
    const float t12 = il12 < 0.f ? 1.0f : clamp01(dot12 * il12);
    const float qx01 = t01 * ex01 - dx0, qy01 = t01 * ey01 - dy0;
    const float qx02 = t02 * ex02 - dx0, qy02 = t02 * ey02 - dy0;
    const float qx12 = t12 * ex12 - dx1, qy12 = t12 * ey12 - dy1;
    const float d01 = qx01 * qx01 + qy01 * qy01;
    const float d02 = qx02 * qx02 + qy02 * qy02;
    const float d12 = qx12 * qx12 + qy12 * qy12;
    const float dist = fmin3(d01, d02, d12);
    // closest edge with [P3D] tie order e01, e02, e12
    const bool s01 = (d01 <= d02) && (d01 <= d12);
    const bool s02 = !s01 && (d02 <= d01) && (d02 <= d12);
    const bool s12 = !s01 && !s02 && (d12 <= d01) && (d12 <= d02);
    c.amin = s01 ? 0 : (s02 ? 1 : 2);
    const bool cand = !(pz < 0.0f) && (inside || dist < kBlurRadius);
    c.cand = cand;
    c.z = pz;
    c.ad = dist;
    const float sd = inside ? -dist : dist;
    // [P3D] sigmoid_alpha_blend: p = sigmoid(-d/sigma) = 1/(1+exp(d/sigma))  
    const float e = __expf(sd * kInvSigma);
    const float p = frcp(1.0f + e);
    c.q = 1.0f - p;
    if (GRAD) {
        // gradient through the closest edge; t recomputed with (l2 + eps) like [P3D]'s backward
        const float bax = s01 ? ex01 : (s02 ? ex02 : ex12);
        const float bay = s01 ? ey01 : (s02 ? ey02 : ey12);
        const float dotv = s01 ? dot01 : (s02 ? dot02 : dot12);
        // 1 / (|b-a|^2 + eps) from the stored 1 / |b-a|^2 (a degenerate edge is flagged -1: its |b-a|^2 <= eps)
        const float il = s01 ? il01 : (s02 ? il02 : il12);
        const float ile = il < 0.f ? 0.5f / kEpsilon : il * frcp(1.0f + kEpsilon * il);
        const float pax = s12 ? dx1 : dx0, pay = s12 ? dy1 : dy0;
        const float tb = clamp01(dotv * ile);
        const float gx = 2.0f * (tb * bax - pax), gy = 2.0f * (tb * bay - pay);  // 2 (proj - p)
        // tangent of the projected point: (1-t) a' + t b'
        const float a_xe = s12 ? rh.x : rg.x, a_ye = s12 ? rh.y : rg.y;
        const float a_xa = s12 ? rh.z : rg.z, a_ya = s12 ? rh.w : rg.w;
        const float b_xe = s01 ? rh.x : ri.x, b_ye = s01 ? rh.y : ri.y;
        const float b_xa = s01 ? rh.z : ri.z, b_ya = s01 ? rh.w : ri.w;
        const float mxe = a_xe + tb * (b_xe - a_xe), mye = a_ye + tb * (b_ye - a_ye);
        const float mxa = a_xa + tb * (b_xa - a_xa), mya = a_ya + tb * (b_ya - a_ya);
        const float any = (s01 || s02 || s12) ? 1.0f : 0.0f;
        const float sp = (inside ? -p : p) * any;
        c.ge = sp * (gx * mxe + gy * mye);
        c.ga = sp * (gx * mxa + gy * mya);
    }
}

__device__ __forceinline__ uint32_t zkey(float z) {
    const uint32_t b = __float_as_uint(z);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// combine a per-lane value over the four lanes that share a pixel (lanes l, l+16, l+32, l+48)
__device__ __forceinline__ int px_sum_i(int v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float px_sum_f(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float px_prod_f(float v) { v *= __shfl_xor(v, 16, 64); v *= __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ uint32_t px_min_u(uint32_t v) { v = min(v, (uint32_t)__shfl_xor((int)v, 16, 64)); v = min(v, (uint32_t)__shfl_xor((int)v, 32, 64)); return v; }
__device__ __forceinline__ uint32_t px_max_u(uint32_t v) { v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64)); v = max(v, (uint32_t)__shfl_xor((int)v, 32, 64)); return v; }
__device__ __forceinline__ bool px_any(bool v) { return px_sum_i(v ? 1 : 0) != 0; }

#ifdef OCC_DBG_BOUNDS  // diagnostic build only: index checks at every memory access of the raster kernel
__device__ int g_dbg_fault[8];
#define OCC_BOUND(cond, code, v0, v1)                                          \
    ((cond) ? true                                                             \
            : ((atomicCAS(&g_dbg_fault[0], 0, (code)) == 0                     \
                    ? (g_dbg_fault[1] = (int)blockIdx.x, g_dbg_fault[2] = (int)threadIdx.x, g_dbg_fault[3] = (int)(v0), \
                       g_dbg_fault[4] = (int)(v1), 0)                          \
                    : 0),                                                      \
               false))
#define OCC_WATCHDOG(code, v0, v1)                                              \
    do {                                                                        \
        if (++wd_iters > 4000000) {                                             \
            (void)OCC_BOUND(false, (code), (v0), (v1));                         \
            return;                                                             \
        }                                                                       \
    } while (0)
#else
#define OCC_BOUND(cond, code, v0, v1) true
#define OCC_WATCHDOG(code, v0, v1) do { } while (0)
#endif

// Exact top-K-by-z for one PIXEL whose candidates sit in the lists of its four lanes (lane g = lane >> 4 holds
// the candidates of faces g, g+4, ... in face order; entry e = (key(z), 1-p, g_el, g_az) at list[e*64 + lane],
// the key being the order-preserving integer image of z).  Keeps the K smallest z like [P3D]'s (pz, face) ordering
// (SURVEY A.4); exact-z ties at the boundary are granted to lane 0 first, then 1, 2, 3, each in face order.
//
// The lists live in HBM/L2 (they do not fit LDS at 11 waves/CU), so the selection touches them as little as
// possible: a most-significant-digit radix select, kHistBits bits per level.  Every lane histograms its OWN key
// rows into a private u16 histogram in LDS (lane stride kHistStride dwords = conflict-free when lanes agree); the
// four histograms of a pixel are summed with two cross-lane adds while they are scanned, so the four lanes
// take identical decisions.  The window [L, L + (1 << kHistBits) << sh) starts at the pixel's own [kmin, kmax] key
// range, so a few levels resolve the bits below the first differing one.  Key sweeps are pipelined 16 rows
// deep (every level is one latency-bound pass over the lists).  The last sweep reads the payload rows once and takes every key below the boundary bucket plus this
// lane's share of the keys inside it.  Pixels with active == false idle.
// COMPACT: also moves the kept entries to the front of each list (stable) and returns the new own count.
// radix-select digit: 4 bits -> 16 u16 buckets = 8 dwords per lane (+1 pad: conflict-free when lanes agree)
constexpr int kHistBits = 4;       // in-loop compaction (the staging buffer is live): own small LDS area
constexpr int kHistBitsFinal = 5;  // final selection: histograms in the idle staging buffer
constexpr int kHistDwords = (1 << kHistBits) / 2;
constexpr int kHistStride = kHistDwords + 1;

template <bool COMPACT, int kBits>
__device__ __forceinline__ void topk_select4(float4* __restrict__ list, uint32_t* __restrict__ hist, int lane,
                                             int cnt, int K, bool active,
                                             uint32_t kmin_own, uint32_t kmax_own, float& pr, float& se, float& sa,
                                             uint32_t& Tmax, int& kept) {
#ifdef OCC_DBG_BOUNDS
    (void)OCC_BOUND(!active || (cnt >= 0 && cnt <= OCC_LIST_CAP), 24, cnt, K);
    const int maxc = min(wave_max_i(active ? cnt : 0), OCC_LIST_CAP);
#else
    const int maxc = wave_max_i(active ? cnt : 0);
#endif
    constexpr int kDwords = (1 << kBits) / 2, kStride = kDwords + 1;  // u16 buckets, one pad dword per lane
    uint32_t* __restrict__ h = hist + lane * kStride;
    // the key is component x of the 16-byte row entry: row e of this lane sits 256 dwords further on
    const uint32_t* __restrict__ keyp = reinterpret_cast<const uint32_t*>(list) + lane * 4;
    const uint32_t kmin = px_min_u(kmin_own), kmax = px_max_u(kmax_own);
    uint32_t L = kmin;
    const uint32_t range = kmax >= kmin ? kmax - kmin : 0u;
    int sh = range ? max(0, (32 - __builtin_clz(range)) - kBits) : 0;
    int need = K;
    int m_own = 0, mstar_px = 0;
    bool done = !active;
    while (__ballot(!done)) {
#pragma unroll
        for (int i = 0; i < kDwords; ++i) h[i] = 0u;
        for (int e0 = 0; e0 < maxc; e0 += 16) {
            uint32_t kk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e0 + i;
                kk[i] = (!done && e < cnt && OCC_BOUND(e < OCC_LIST_CAP, 21, e, cnt)) ? keyp[(size_t)e * 256] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e0 + i;
                if (!done && e < cnt) {
                    const uint32_t k = kk[i];
                    const uint32_t d = (k - L) >> sh;
                    if (k >= L && d < (1u << kBits)) h[d >> 1] += 1u << (16 * (d & 1u));
                }
            }
        }
        int cum = 0, bstar = (1 << kBits) - 1, mstar = 0, cumb = 0, mown = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < kDwords; ++i) {
            const uint32_t wo = h[i];
            uint32_t w = wo;  // joint histogram of the pixel: the four private ones added up
            w += (uint32_t)__shfl_xor((int)w, 16, 64);
            w += (uint32_t)__shfl_xor((int)w, 32, 64);
            const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16);
            if (!found && cum + c0 >= need) { found = true; bstar = 2 * i; mstar = c0; cumb = cum; mown = (int)(wo & 0xFFFFu); }
            cum += c0;
            if (!found && cum + c1 >= need) { found = true; bstar = 2 * i + 1; mstar = c1; cumb = cum; mown = (int)(wo >> 16); }
            cum += c1;
        }
        if (!done) {
            need -= cumb;
            L += (uint32_t)bstar << sh;
            m_own = mown;
            mstar_px = mstar;
            if (mstar == need || sh == 0 || !found) {
                done = true;
            } else {
                sh = max(0, sh - kBits);
            }
        }
    }
    // this lane's share of the boundary bucket [L, L + 2^sh): all of it when the whole bucket is taken, else
    // (exact ties) lanes are served in order 0, 1, 2, 3
    const int g = lane >> 4, base = lane & 15;
    int before = 0;
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) {
        const int mo = __shfl(m_own, base + 16 * gg, 64);
        if (gg < g) before += mo;
    }
    int take = (mstar_px == need) ? m_own : min(m_own, max(0, need - before));
    int w = 0;
    pr = 1.0f;
    se = 0.f;
    sa = 0.f;
    uint32_t tmax = 0;
    // (payload sweep: 4 rows in flight - 8 would set the kernel's register peak and cost a wave per SIMD)
    for (int e0 = 0; e0 < maxc; e0 += 4) {
        uint32_t kk[4];
        bool inc[4];
        float4 vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            kk[i] = (active && e < cnt && OCC_BOUND(e < OCC_LIST_CAP, 22, e, cnt)) ? keyp[(size_t)e * 256] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            const uint32_t k = kk[i];
            bool in_ = false;
            if (active && e < cnt) {
                in_ = k < L;
                if (k >= L && ((k - L) >> sh) == 0u && take > 0) {
                    in_ = true;
                    take -= 1;
                }
            }
            inc[i] = in_;
            vv[i] = make_float4(0.f, 1.f, 0.f, 0.f);
            if (in_) vv[i] = list[(size_t)e * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (inc[i]) {
                pr *= vv[i].y;
                se += vv[i].z;
                sa += vv[i].w;
                tmax = max(tmax, kk[i]);
                if (COMPACT && OCC_BOUND(w < OCC_LIST_CAP, 23, w, cnt)) list[(size_t)w * 64 + lane] = vv[i];
                w += 1;
            }
        }
    }
    Tmax = px_max_u(tmax);
    kept = w;
}

#ifdef OCC_DBG_STATS  // diagnostic build only: loop trip counts of the raster kernel
__device__ unsigned long long g_dbg_stats[8];
#define OCC_STAT(i, v) do { if (lane == 0) atomicAdd(&g_dbg_stats[i], (unsigned long long)(v)); } while (0)
#else
#define OCC_STAT(i, v) do { } while (0)
#endif

struct RasterParams {
    OccScene sc;
    OccWorkspace ws;
    OccRenderOut out;
    const float* cam;
    int K;
    int ntx;  // tiles per image side
};

// XCD-major order of the (env, object) pairs: env e belongs to XCD group e % 8; group g holds MQ = 3*ceil(N/8) slots.
// All blocks of an env are then dequeued by waves of ONE XCD (when placement follows XCC_ID), so the face
// records of an object, staged again by every block they touch, are fetched into one L2 instead of eight.
__host__ __device__ __forceinline__ int xcd_slots(int n_env) { return 3 * ((n_env + 7) / 8); }
__device__ __forceinline__ int perm_to_eo(int p, int mq, int n_env) {
    const int g = p / mq, slot = p - g * mq;
    const int e = (slot / 3) * 8 + g;
    return e < n_env ? e * 3 + slot % 3 : -1;
}

// One block: exclusive prefix sum of the block counts of every (env, object) rect, in XCD-major order
// -> work-item offsets (8*MQ + 1 entries).
__global__ __launch_bounds__(1024) void occ_scan_kernel(const int* __restrict__ objrect, const int* __restrict__ nrec,
                                                        int* __restrict__ offsets, int n_env) {
    __shared__ int s_part[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mq = xcd_slots(n_env), M = 8 * mq;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < M; base += 1024) {
        const int i = base + tid;
        int c = 0;
        const int eo = i < M ? perm_to_eo(i, mq, n_env) : -1;
        if (eo >= 0 && nrec[eo] > 0) {
            const int w = objrect[4 * eo + 2] - objrect[4 * eo] + 1, h = objrect[4 * eo + 3] - objrect[4 * eo + 1] + 1;
            c = (w > 0 && h > 0) ? w * h : 0;
        }
        int incl = c;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int v = s_part[w];
            if (w < wave) woff += v;
            tot += v;
        }
        const int carry = s_carry;
        if (i < M) offsets[i] = carry + woff + incl - c;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) offsets[M] = s_carry;
}

// ------------------------------------------------------------------------------------------
// raster kernel: one persistent wave64 per work item (env, object, 4x4-pixel block inside the object's rect)
// ------------------------------------------------------------------------------------------
// Lane layout: lane = 16 g + l.  l = pixel of the block (x = l & 3, y = l >> 2); g = FACE SLOT: one loop
// iteration evaluates four different faces (slots 4 it + g of the staged hit list) at the 16 pixels of the
// block, so a face whose footprint (~5x5 px with the blur margin) is about the size of the block no longer
// costs a 64-lane pass.  Every pixel's candidates are therefore spread over four lanes (each in face order):
// counts, products, tangent sums and the nearest hard face are folded across the four lanes at the end of the
// item, and the exact top-K selection works on the four lists jointly (topk_select4).
#ifndef OCC_RASTER_WAVES_PER_SIMD
#define OCC_RASTER_WAVES_PER_SIMD 5
#endif
template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(64, OCC_RASTER_WAVES_PER_SIMD) void occ_raster_kernel(RasterParams P) {
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    const int px = l & 3, py = l >> 2;
    const int S = P.sc.img;
    const float fS = (float)S;
    const int cap = P.sc.rec_cap;
    const int K = P.K;
    // per-wave K-buffer: OCC_LIST_CAP rows of 64 lane entries (key(z), 1-p, g_el, g_az), 16 B each
    float4* __restrict__ mylist = reinterpret_cast<float4*>(P.ws.lists) + (size_t)blockIdx.x * OCC_LIST_CAP * 64;
    __shared__ uint32_t s_hist[64 * kHistStride];
    // records of the faces that touch this block, gathered over as many 64-face chunks as fit, staged by
    // cooperative 16-B loads (one memory round trip per <= 64 staged faces)
    constexpr int kParts = GRAD ? kRecParts : (SOFT ? 5 : 4);  // float4 parts of a record that this variant reads
    constexpr int kStage = 40;  // 40 x 128 B = 5 KiB: the wave stays below 8 KiB of LDS -> 20 waves per CU
    __shared__ float4 s_stage[kStage * kRecParts];
    __shared__ int s_hit[kStage];  // record index of every staged face
    ciptr offs = as_const(P.ws.offsets);
    const int mq = xcd_slots(P.sc.n_env), MP = 8 * mq;
    // this wave's XCD (HW_REG_XCC_ID, bits 3:0); only steers WHICH queue is drained first - any value is correct
    const int my_xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    int qround = 0;  // queues visited so far: own XCD's first, then the others (work stealing)
#ifdef OCC_DBG_BOUNDS
    int wd_iters = 0;
#endif

    for (;;) {
        OCC_WATCHDOG(35, qround, 0);
        int item = -1;
        while (qround < 8) {
            const int qq = (my_xcc + qround) & 7;
            const int qbeg = offs[qq * mq], qend = offs[(qq + 1) * mq];
            int t = qend;
            if (lane == 0 && qbeg < qend) t = qbeg + (int)atomicAdd(P.ws.queue + qq * 16, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < qend) {
                item = t;
                break;
            }
            qround += 1;
        }
        if (item < 0) break;
        // (env, object) of this item: largest permuted index p with offsets[p] <= item
        int lo = 0, hi = MP;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= item) lo = mid; else hi = mid;
        }
        const int eo = perm_to_eo(lo, mq, P.sc.n_env);
        if (!OCC_BOUND(eo >= 0 && eo < 3 * P.sc.n_env, 1, eo, item)) continue;
        const int local = item - offs[lo];
        ciptr rect = as_const(P.ws.objrect + eo * 4);
        const int rw = rect[2] - rect[0] + 1;
        const int by = rect[1] + local / rw, bx = rect[0] + local % rw;
        const int x0b = bx * OCC_BLOCK, y0b = by * OCC_BLOCK;
        if (x0b < 0 || y0b < 0 || x0b + OCC_BLOCK > S || y0b + OCC_BLOCK > S) continue;  // never true for a sane rect
        const int xi = x0b + px, yi = y0b + py;
        // [P3D] pixel centre in NDC, +X left, +Y up
        const float xf = -1.0f + (2.0f * (float)(S - 1 - xi) + 1.0f) / fS;
        const float yf = -1.0f + (2.0f * (float)(S - 1 - yi) + 1.0f) / fS;
        const int n = as_const(P.ws.nrec + eo)[0];
        if (!OCC_BOUND(xi >= 0 && xi < S && yi >= 0 && yi < S && n >= 0 && n <= cap, 2, xi | (yi << 16), n)) continue;
        OCC_STAT(0, 1);              // work items
        const float* __restrict__ recs = P.ws.rec + (size_t)eo * cap * OCC_REC_STRIDE;
        const uint4* __restrict__ bbs = reinterpret_cast<const uint4*>(P.ws.rec_bbox) + (size_t)eo * cap;
        const uint4* __restrict__ scan = reinterpret_cast<const uint4*>(P.ws.scan) + (size_t)eo * cap;

        float hz = 3.0e38f;
        int hrec = 0x7FFFFFFF;
        int count = 0;          // candidates in THIS lane's list
        float prod = 1.0f, sge = 0.f, sga = 0.f;
        bool thr_on = false;    // set once the pixel's lists have been compacted to its K nearest
        bool lim_on = false;    // pixel already holds >= K candidates
        // key bound of the pixel (equal in its four lanes): a later candidate needs key < bnd to matter.  Lowered to
        // the largest stored key once the pixel holds >= K candidates (that key bounds the K-th nearest from
        // above), and to the K-th nearest key itself whenever the lists are compacted
        uint32_t bnd = 0xFFFFFFFFu;
        uint32_t thrB = 0xFFFFFFFFu;  // block-wide skip key (wave-uniform): faces whose nearest vertex is not
                                      // nearer than this can change neither a pixel's K nearest nor its hard face
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;  // key range of this lane's stored candidates

        auto touches = [&](uint4 bb) {
            const int rx0 = bb.x & 0xFFFF, ry0 = bb.x >> 16, rx1 = bb.y & 0xFFFF, ry1 = bb.y >> 16;
            return (rx0 <= x0b + OCC_BLOCK - 1) && (rx1 >= x0b) && (ry0 <= y0b + OCC_BLOCK - 1) && (ry1 >= y0b);
        };

        auto commit = [&](bool cnd, float z, float qv, float ge, float ga) {
            const uint32_t key = zkey(z);
            bool acc = cnd && key < bnd;
            if (__ballot(acc && count >= OCC_LIST_CAP)) {
                // rare: a lane's list is full -> keep the pixel's K nearest (over its four lists), go on
                const bool full = px_any(count >= OCC_LIST_CAP);
                float pr, se, sa;
                uint32_t T;
                int kept;
                topk_select4<true, kHistBits>(mylist, s_hist, lane, count, K, full, kmin, kmax, pr, se, sa, T, kept);
                if (full) {
                    count = kept;
                    thr_on = true;
                    bnd = min(bnd, T);
                    kmax = T;
                    acc = cnd && (key < bnd);
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave no load pending across the hot loop
            }
            if (acc && OCC_BOUND(count < OCC_LIST_CAP, 3, count, item)) {
#ifndef OCC_DBG_NO_STORE  // timing experiment only
                // one 16-byte store per candidate; 32-bit offset from the wave-uniform base
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(mylist) + (uint32_t)(count * 1024 + lane * 16)) =
                    make_float4(__uint_as_float(key), qv, ge, ga);
#endif
                kmin = min(kmin, key);
                kmax = max(kmax, key);
                count += 1;
                prod *= qv;
                sge += ge;
                sga += ga;
            }
        };

        int nst = 0;  // staged faces (wave-uniform)
        auto process_staged = [&]() {
            __syncthreads();
#ifndef OCC_DBG_NO_STAGE  // timing experiment only
            for (int idx = lane; idx < nst * kParts; idx += 64) {
                const int k = idx / kParts, part = idx - k * kParts;
                if (OCC_BOUND(s_hit[k] >= 0 && s_hit[k] < n, 4, s_hit[k], n))
                    s_stage[k * kRecParts + part] = reinterpret_cast<const float4*>(recs + (size_t)s_hit[k] * OCC_REC_STRIDE)[part];
            }
#endif
            __syncthreads();
            int niter = (nst + 3) >> 2;
            OCC_STAT(1, 1);      // staging rounds
            OCC_STAT(2, nst);    // staged records = (face, block) pairs
            OCC_STAT(3, niter);  // eval iterations
#ifdef OCC_DBG_NO_EVAL  // timing experiment only
            niter = 0;
#endif
            for (int it = 0; it < niter; ++it) {
                OCC_WATCHDOG(31, nst, n);
                const int slot = 4 * it + g;
                bool active = slot < nst;
                const int sidx = active ? slot : 0;
                const int j = s_hit[sidx];
                const float4* rs = &s_stage[sidx * kRecParts];
                Cand c1;
                eval_face<SOFT, GRAD>(OCC_REC_LOAD(rs, kParts), xf, yf, c1);
                const int flags = active ? __float_as_int(rs[2].z) : 0;
                // Clipped quad split in two: the pair is resolved where its SECOND half is visited.
                // A FIRST half whose partner also touches the block is skipped here; a SECOND half whose partner
                // touches the block evaluates the partner too and keeps one of them.
                if (__ballot(flags & (FLAG_PAIR_FIRST | FLAG_PAIR_SECOND))) {
                    const bool is_first = (flags & FLAG_PAIR_FIRST) != 0, is_second = (flags & FLAG_PAIR_SECOND) != 0;
                    bool partner = false;
                    if (is_first && j + 1 < n && OCC_BOUND(j >= 0, 5, j, n)) partner = touches(bbs[j + 1]);
                    if (is_second && j >= 1 && OCC_BOUND(j < n, 6, j, n)) partner = touches(bbs[j - 1]);
                    if (is_first && partner) active = false;
                    if (__ballot(is_second && partner)) {
                        const float4* r1 = reinterpret_cast<const float4*>(recs + (size_t)(is_second && partner ? j - 1 : j) *
                                                                          OCC_REC_STRIDE);
                        Cand cf;
                        eval_face<SOFT, GRAD>(OCC_REC_LOAD(r1, kParts), xf, yf, cf);
                        if (is_second && partner) {
                            if (HARD) {
                                if (cf.inside && (cf.zh < hz || (cf.zh == hz && j - 1 < hrec))) {
                                    hz = cf.zh;
                                    hrec = j - 1;
                                }
                            }
                            // [P3D]: the second half replaces the first iff its |d| is strictly smaller.  If both
                            // are closest to the diagonal they share (t1: edge (v1,v2), t2: edge (v0,v1)) the
                            // distances are equal in exact arithmetic: keep the first.
                            const bool shared_tie = (cf.amin == 2) && (c1.amin == 0);
                            const bool take2 = c1.cand && (!cf.cand || (!shared_tie && c1.ad < cf.ad));
                            if (!take2) {
                                const bool ins = c1.inside;
                                const float zh1 = c1.zh;
                                c1 = cf;
                                c1.inside = ins;  // the hard pass still sees the second half itself
                                c1.zh = zh1;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave no load pending across the hot loop
                }
                c1.cand = c1.cand && active;
                c1.inside = c1.inside && active;
                if (HARD) {
                    if (c1.inside && (c1.zh < hz || (c1.zh == hz && j < hrec))) {
                        hz = c1.zh;
                        hrec = j;
                    }
                }
                if (SOFT) commit(c1.cand, c1.z, c1.q, c1.ge, c1.ga);
            }
            __syncthreads();
            nst = 0;
            // Front-to-back pruning (the scan order is ascending in the faces' nearest vertex depth and a
            // candidate's depth is never below it).  Once a pixel holds >= K candidates, the largest stored key
            // bounds its K-th nearest from above -> later candidates at or beyond it are dropped unseen; once this
            // holds for all 16 pixels, and every pixel has a hard face, faces starting beyond both bounds are
            // skipped altogether and the item ends at the first such chunk.
            if (SOFT) {
                const int ctot = px_sum_i(count);
                if (!lim_on && ctot >= K) {
                    lim_on = true;
                    bnd = min(bnd, px_max_u(kmax));
                }
            }
            uint32_t bound = 0xFFFFFFFFu;
            if (SOFT) bound = bnd;  // 0xFFFFFFFF until the pixel holds >= K candidates
            if (HARD) {
                const uint32_t hk = hz < 3.0e38f ? zkey(hz) : 0xFFFFFFFFu;  // every lane keeps its own nearest so far
                bound = SOFT ? max(bound, px_min_u(hk)) : px_min_u(hk);
            }
            // wave max over the 16 pixels (each pixel's four lanes agree)
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) bound = max(bound, (uint32_t)__shfl_xor((int)bound, m, 64));
            thrB = (uint32_t)__builtin_amdgcn_readfirstlane((int)bound);
        };

        // two-level scan: chunk boxes (one lane per 64-record chunk) -> candidate chunks -> their record boxes,
        // the next candidate chunk's row of boxes being fetched while the current one is processed
        const int nch = (n + 63) >> 6;
        const uint4* __restrict__ cbx = reinterpret_cast<const uint4*>(P.ws.rec_cbox) + (size_t)eo * ((cap + 63) >> 6);
        const uint4 kEmptyBox = make_uint4(0xFFFFu, 0u, 0xFFFFFFFFu, 0u);  // x0 = 65535 > any pixel: never overlaps
        int cwin = -64;
        unsigned long long cmask = 0;
        auto next_chunk = [&]() -> int {
            while (!cmask) {
#ifdef OCC_DBG_BOUNDS
                if (++wd_iters > 4000000) { (void)OCC_BOUND(false, 34, cwin, nch); return -1; }
#endif
                cwin += 64;
                if (cwin >= nch) return -1;
                uint4 cb = kEmptyBox;
                if (cwin + lane < nch && OCC_BOUND(nch <= ((cap + 63) >> 6), 7, nch, cap)) cb = cbx[cwin + lane];
                cmask = __ballot(touches(cb) && cb.z < thrB);
            }
            const int bit = __builtin_ctzll(cmask);
            cmask &= cmask - 1;
            return cwin + bit;
        };
        int c = next_chunk();
        uint4 bb_cur = kEmptyBox;
        if (c >= 0 && c * 64 + lane < n && OCC_BOUND(c * 64 + lane < cap, 8, c, n)) bb_cur = scan[c * 64 + lane];
        while (c >= 0) {
            OCC_WATCHDOG(33, c, n);
            const int cn = next_chunk();
            uint4 bb_nxt = kEmptyBox;
            if (cn >= 0 && cn * 64 + lane < n && OCC_BOUND(cn * 64 + lane < cap, 9, cn, n)) bb_nxt = scan[cn * 64 + lane];
            const bool hit = touches(bb_cur) && bb_cur.z < thrB;
            const unsigned long long mask = __ballot(hit);
            OCC_STAT(5, 1);  // chunk rows scanned
            unsigned long long m = mask;
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            while (m) {  // a chunk may hold more hits than the staging buffer has room for
                OCC_WATCHDOG(32, nst, n);
                const int room = kStage - nst;
                const int cnt = __popcll(m);
                const int rank = __popcll(m & lt);
                const bool mine = (m >> lane) & 1ull;
                if (mine && rank < room) s_hit[nst + rank] = (int)bb_cur.w;
                if (cnt <= room) {
                    nst += cnt;
                    m = 0;
                } else {
                    nst = kStage;
                    m = __ballot(mine && rank >= room);
                    process_staged();
                }
            }
            c = cn;
            bb_cur = bb_nxt;
        }
        if (nst > 0) process_staged();

        // ---- fold the four lanes of every pixel ---------------------------------------------------------
        const size_t opix = ((size_t)eo * S + yi) * S + xi;
        if (SOFT) {
            const int ctot = px_sum_i(count);
#ifdef OCC_DBG_NO_TOPK  // timing experiment only: skip the exact selection (results wrong where count > K)
            const bool ovf = false;
#else
            const bool ovf = (ctot > K) || px_any(thr_on);
#endif
#ifdef OCC_DBG_STATS
            {
                const int cw = (int)wave_sum((float)count);
                const int co = (int)wave_sum(ovf ? (float)count : 0.f);
                OCC_STAT(4, cw);                                   // candidates stored
                OCC_STAT(6, co);                                   // ... of which in pixels that need selection
                OCC_STAT(7, __ballot(ovf) ? 1 : 0);                // items with at least one such pixel
            }
#endif
            if (__ballot(ovf)) {
                // more than K candidates: keep the K nearest in z, SURVEY A.4
                float pr, se, sa;
                uint32_t T;
                int kept;
                // the staging buffer is idle now: its LDS holds the wider (5-bit digit) histograms of the final
                // selection - one level less on average, each level being a pass over the lists in memory
                static_assert(sizeof(float4) * kStage * kRecParts >= 64 * ((1 << kHistBitsFinal) / 2 + 1) * 4, "hist");
                __syncthreads();
                topk_select4<false, kHistBitsFinal>(mylist, reinterpret_cast<uint32_t*>(s_stage), lane, count, K, ovf, kmin,
                                                    kmax, pr, se, sa, T, kept);
                __syncthreads();
                if (ovf) {
                    prod = pr;
                    sge = se;
                    sga = sa;
                }
            }
            prod = px_prod_f(prod);
            if (GRAD) {
                sge = px_sum_f(sge);
                sga = px_sum_f(sga);
            }
            if (g == 0) {
                P.ws.obj_alpha[opix] = 1.0f - prod;
                if (GRAD) {
                    // d alpha/d theta = -(A/sigma) * sum_k p_k d(d_k)/d theta   
                    const float coef = -prod * kInvSigma;
                    reinterpret_cast<float2*>(P.ws.obj_grad)[opix] = make_float2(coef * sge, coef * sga);
                }
            }
        }
        if (HARD) {
            // nearest face over the four lanes; equal depth -> smaller record (= face) index, like (pz, f) order
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float oz = __shfl_xor(hz, m, 64);
                const int orr = __shfl_xor(hrec, m, 64);
                if (oz < hz || (oz == hz && orr < hrec)) {
                    hz = oz;
                    hrec = orr;
                }
            }
            if (g == 0) {
                P.ws.obj_hz[opix] = hz;
                P.ws.obj_hrec[opix] = (hrec == 0x7FFFFFFF) ? -1 : hrec;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// combine kernel: one thread per pixel - occlusion image, loss / gradient partials, shading, outputs
// ------------------------------------------------------------------------------------------
template <bool SOFT, bool HARD, bool GRAD>
__global__ __launch_bounds__(256) void occ_combine_kernel(RasterParams P, int bpe) {
    __shared__ float s_red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env = blockIdx.x / bpe, blk = blockIdx.x - env * bpe;
    if (P.sc.skip && P.sc.skip[env]) return;  // outputs of a skipped scene row stay untouched
    const int S = P.sc.img;
    const int pix = blk * 256 + tid;
    const bool live = pix < S * S;
    const int yi = live ? pix / S : 0, xi = live ? pix - (pix / S) * S : 0;
    const int tx = xi / OCC_BLOCK, ty = yi / OCC_BLOCK;
    const int cap = P.sc.rec_cap;
    float alpha[3] = {0.f, 0.f, 0.f}, dae[3] = {0.f, 0.f, 0.f}, daa[3] = {0.f, 0.f, 0.f};
    float hz = 3.0e38f;
    int hrec = -1, hobj = 0;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int eo = env * 3 + o;
        ciptr rect = as_const(P.ws.objrect + eo * 4);
        const bool in = live && as_const(P.ws.nrec + eo)[0] > 0 && tx >= rect[0] && ty >= rect[1] && tx <= rect[2] &&
                        ty <= rect[3];
        if (in) {
            const size_t opix = ((size_t)eo * S + yi) * S + xi;
            if (SOFT) {
                alpha[o] = P.ws.obj_alpha[opix];
                if (GRAD) {
                    const float2 g = reinterpret_cast<const float2*>(P.ws.obj_grad)[opix];
                    dae[o] = g.x;
                    daa[o] = g.y;
                }
            }
            if (HARD) {
                const float z = P.ws.obj_hz[opix];
                if (z < hz) {  // strict: on equal depth the earlier object of the joined scene wins
                    hz = z;
                    hrec = P.ws.obj_hrec[opix];
                    hobj = o;
                }
            }
        }
    }
    const size_t gp = (size_t)yi * S + xi;
    if (SOFT) {
        // environment.py:373: image = i1*i2 + i2*i3 + i1*i3 ; RGB of every silhouette is 1
        const float I = alpha[0] * alpha[1] + alpha[1] * alpha[2] + alpha[0] * alpha[2];
        float lsum = live ? I * I : 0.f, ge = 0.f, ga = 0.f;
        if (GRAD && live) {
            const float g0 = alpha[1] + alpha[2], g1 = alpha[0] + alpha[2], g2 = alpha[0] + alpha[1];
            ge = 2.0f * I * (g0 * dae[0] + g1 * dae[1] + g2 * dae[2]);
            ga = 2.0f * I * (g0 * daa[0] + g1 * daa[1] + g2 * daa[2]);
        }
        lsum = wave_sum(lsum);
        if (GRAD) {
            ge = wave_sum(ge);
            ga = wave_sum(ga);
        }
        if (lane == 0) {
            s_red[wave][0] = lsum;
            s_red[wave][1] = ge;
            s_red[wave][2] = ga;
        }
        __syncthreads();
        if (tid == 0) {
            float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                a += s_red[w][0];
                b += s_red[w][1];
                c += s_red[w][2];
            }
            reinterpret_cast<float4*>(P.ws.partials)[blockIdx.x] = make_float4(a, b, c, 0.f);
        }
        if (live) {
            if (P.out.full_state)
                reinterpret_cast<float4*>(P.out.full_state)[(size_t)env * S * S + gp] = make_float4(3.f, 3.f, 3.f, I);
            if (P.out.alphas) {
                float* __restrict__ al = P.out.alphas + (size_t)env * 3 * S * S + gp;
                al[0] = alpha[0];
                al[(size_t)S * S] = alpha[1];
                al[(size_t)2 * S * S] = alpha[2];
            }
        }
    }
    if (HARD && live) {
        // [P3D] HardFlatShader + hard_rgb_blend; depth in channel 3 (environment.py:378)
        float cr = 1.f, cg = 1.f, cb = 1.f, depth = -1.f;
        if (hrec >= 0) {
            const int eo = env * 3 + hobj;
            const float* __restrict__ r = P.ws.rec + ((size_t)eo * cap + hrec) * OCC_REC_STRIDE;
            const int fid = __float_as_int(r[R_ID]);
            const int mesh = P.sc.scene_mesh[eo];
            const int vo = P.sc.mesh_vert_off[mesh], fo = P.sc.mesh_face_off[mesh];
            const float ox = P.sc.scene_offset[eo * 3], oy = P.sc.scene_offset[eo * 3 + 1], oz = P.sc.scene_offset[eo * 3 + 2];
            // per-face shading terms from the setup kernel (flat_shade): one gather instead of face -> 3 vertices
            const float amb_diff = __uint_as_float(reinterpret_cast<const uint4*>(P.ws.rec_bbox)[(size_t)eo * cap + hrec].w);
            const float spec = r[R_SPEC];
            const float* __restrict__ cm = P.cam + (size_t)env * OCC_CAM_STRIDE;
            // texel: white TexturesVertex interpolated with the (unclipped) barycentrics, or the face's atlas
            const float fS = (float)S;
            const float xf = -1.0f + (2.0f * (float)(S - 1 - xi) + 1.0f) / fS;
            const float yf = -1.0f + (2.0f * (float)(S - 1 - yi) + 1.0f) / fS;
            const float x0 = r[R_X0], y0 = r[R_Y0], z0 = r[R_Z0], x1 = r[R_X1], y1 = r[R_Y1], z1 = r[R_Z1];
            const float x2 = r[R_X2], y2 = r[R_Y2], z2 = r[R_Z2];
            const float ia = r[R_INV_AREA];
            const float b0 = ((xf - x1) * (y2 - y1) - (yf - y1) * (x2 - x1)) * ia;
            const float b1 = ((yf - y2) * (x2 - x0) - (xf - x2) * (y2 - y0)) * ia;
            const float b2 = ((xf - x0) * (y1 - y0) - (yf - y0) * (x1 - x0)) * ia;
            const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
            const float den = fmaxf(w0 + w1 + w2, kEpsilon);
            float q0 = w0 / den, q1 = w1 / den, q2 = w2 / den;
            float tr = q0 + q1 + q2, tg = tr, tb = tr;
            const int64_t aoff = P.sc.pool_atlas ? P.sc.mesh_atlas_off[mesh] : -1;
            if (aoff >= 0) {
                if (__float_as_int(r[R_FLAGS]) & FLAG_CLIPPED) {
                    // [P3D] convert_clipped_rasterization_to_original_faces: barycentrics w.r.t. the ORIGINAL face.
                    // Perspective-correct barycentrics are the 3-D ones: beta_i ~ d . (V_j x V_k) with d the pixel ray
                    // and V the face's view-space vertices (valid for vertices behind the clip plane too).
                    float V[3][3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int vi = P.sc.pool_faces[(size_t)(fo + fid) * 3 + k];
                        const float* pv = P.sc.pool_verts + (size_t)(vo + vi) * 3;
                        const float wx_ = pv[0] + ox, wy_ = pv[1] + oy, wz_ = pv[2] + oz;
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            V[k][j] = wx_ * cm[C_R + j] + wy_ * cm[C_R + 3 + j] + wz_ * cm[C_R + 6 + j] + cm[C_T + j];
                    }
                    const float dx = xf / kProjScale, dy = yf / kProjScale, dz = 1.0f;
                    auto tri = [&](const float* a, const float* b) {
                        return dx * (a[1] * b[2] - a[2] * b[1]) + dy * (a[2] * b[0] - a[0] * b[2]) + dz * (a[0] * b[1] - a[1] * b[0]);
                    };
                    const float e0 = tri(V[1], V[2]), e1 = tri(V[2], V[0]), e2 = tri(V[0], V[1]);
                    const float es = e0 + e1 + e2;
                    q0 = e0 / es; q1 = e1 / es; q2 = e2 / es;
                }
                // [P3D] TexturesAtlas.sample_textures: (w0, w1) -> texel of the R x R grid, upper triangle mirrored
                const int Rr = P.sc.atlas_res;
                int wx = min((int)(q0 * (float)Rr), Rr - 1), wy = min((int)(q1 * (float)Rr), Rr - 1);
                const bool below = ((q0 + q1) * (float)Rr - ((float)wx + (float)wy)) <= 1.0f;
                if (!below) { wx = Rr - 1 - wx; wy = Rr - 1 - wy; }
                const float* tx = P.sc.pool_atlas + aoff + (((size_t)fid * Rr + wy) * Rr + wx) * 3;
                tr = tx[0]; tg = tx[1]; tb = tx[2];
            }
            cr = amb_diff * tr + spec;
            cg = amb_diff * tg + spec;
            cb = amb_diff * tb + spec;
            depth = hz;
        }
        float* __restrict__ ob = P.out.obs + (size_t)env * 4 * S * S + gp;
        ob[0] = cr;
        ob[(size_t)S * S] = cg;
        ob[(size_t)2 * S * S] = cb;
        ob[(size_t)3 * S * S] = depth;
    }
}

// ------------------------------------------------------------------------------------------
// per-env fixed-order reduction of the tile partials
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void occ_reduce_kernel(const float* __restrict__ partials, int ntiles,
                                                        float* __restrict__ loss, float* __restrict__ grad_elaz,
                                                        const int* __restrict__ skip) {
    const int env = blockIdx.x, lane = threadIdx.x;
    if (skip && skip[env]) return;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(partials) + (size_t)env * ntiles;
    float l = 0.f, ge = 0.f, ga = 0.f;
    for (int t = lane; t < ntiles; t += 64) {
        const float4 v = p[t];
        l += v.x;
        ge += v.y;
        ga += v.z;
    }
    l = wave_sum(l);
    ge = wave_sum(ge);
    ga = wave_sum(ga);
    if (lane == 0) {
        if (loss) loss[env] = l;
        if (grad_elaz) {
            grad_elaz[2 * env] = ge;
            grad_elaz[2 * env + 1] = ga;
        }
    }
}

// environment.py:381-392 + action Jacobian (:356-361)
__global__ __launch_bounds__(64) void occ_finish_kernel(const float* __restrict__ loss, const float* __restrict__ grad_elaz,
                                                        const float* __restrict__ cam, float* __restrict__ full_reward,
                                                        const float* __restrict__ object_mass, float* __restrict__ reward,
                                                        uint8_t* __restrict__ done, float* __restrict__ grad_action,
                                                        int n_env) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_env) return;
    const float l = loss[n];
    const float om = object_mass[n];
    float rw = full_reward[n] - l;
    full_reward[n] = l;
    const bool fin = l < kDoneThreshold;
    rw = rw / om;
    rw = fin ? rw + kDoneBonus : rw - kStepPenalty;
    reward[n] = rw;
    done[n] = fin ? 1 : 0;
    if (grad_action) {
        float ga0 = 0.f, ga1 = 0.f;
        if (grad_elaz) {
            const float* __restrict__ J = cam + (size_t)n * OCC_CAM_STRIDE + C_J;
            const float gl_e = grad_elaz[2 * n], gl_a = grad_elaz[2 * n + 1];
            // d reward/d action_j = -(1/objectMass) * (dL/del * del/da_j + dL/daz * daz/da_j)
            ga0 = -(gl_e * J[0] + gl_a * J[2]) / om;
            ga1 = -(gl_e * J[1] + gl_a * J[3]) / om;
        }
        grad_action[2 * n] = ga0;
        grad_action[2 * n + 1] = ga1;
    }
}

// One small int32 buffer per step for the host: [0..n) done, [n..n+r) reserve scene passes the reset test
// (loss > 0.1, environment.py:327), [n+r] any kernel status bit set -> ONE device-to-host copy per step.
__global__ __launch_bounds__(256) void occ_flags_kernel(const uint8_t* __restrict__ done, const float* __restrict__ loss_all,
                                                        const int* __restrict__ status, int n, int r,
                                                        int* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flags[n + r] = 0;
    if (i < n) flags[i] = done[i];
    else if (i < n + r) flags[i] = loss_all[i] > kDoneThreshold ? 1 : 0;
}
__global__ __launch_bounds__(256) void occ_status_any_kernel(const int* __restrict__ status, int nt, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt && status[i] != 0) atomicOr(out, 1);
}

// Auto-reset commit: env dst[k] takes over reserve row src[k] (all per-env state + the freshly rendered
// observation) in one launch.  block = one (k, array) pair chunk.
struct CommitArgs {
    const int* pairs;  // (n,2): dst env row, src row
    int n;
    float* el; float* az; float* radius; float* campos; float* cam; float* alphas; float* full_reward; float* object_mass;
    int* scene_mesh; float* scene_offset; float* obs; const float* obs_all; const float* loss_all;
    int img;
};
__global__ __launch_bounds__(256) void occ_commit_kernel(CommitArgs a) {
    const int k = blockIdx.x;
    const int dst = a.pairs[2 * k], src = a.pairs[2 * k + 1];
    const int tid = threadIdx.x;
    const size_t S2 = (size_t)a.img * a.img;
    if (blockIdx.y == 0) {
        if (tid == 0) {
            a.el[dst] = a.el[src];
            a.az[dst] = a.az[src];
            a.radius[dst] = a.radius[src];
            const float l = a.loss_all[src];
            a.full_reward[dst] = l;
            a.object_mass[dst] = l + 1.0f;
        }
        if (tid < 3) {
            a.campos[dst * 3 + tid] = 0.f;
            a.scene_mesh[dst * 3 + tid] = a.scene_mesh[src * 3 + tid];
        }
        if (tid < 9) a.scene_offset[dst * 9 + tid] = a.scene_offset[src * 9 + tid];
        if (tid < OCC_CAM_STRIDE) a.cam[(size_t)dst * OCC_CAM_STRIDE + tid] = a.cam[(size_t)src * OCC_CAM_STRIDE + tid];
    } else if (blockIdx.y == 1) {
        const float4* s4 = reinterpret_cast<const float4*>(a.obs_all + (size_t)src * 4 * S2);
        float4* d4 = reinterpret_cast<float4*>(a.obs + (size_t)dst * 4 * S2);
        for (size_t i = tid; i < S2; i += 256) d4[i] = s4[i];
    } else {
        const float* s1 = a.alphas + (size_t)src * 3 * S2;
        float* d1 = a.alphas + (size_t)dst * 3 * S2;
        for (size_t i = tid; i < 3 * S2; i += 256) d1[i] = s1[i];
    }
}


// [P3D] sigmoid_alpha_blend on K-buffers (operator level; the fused path never materialises them).  One thread per
// pixel, plain IEEE arithmetic in PyTorch3D's order: prob = sigmoid(-d / sigma) * mask, alpha = 1 - prod(1 - prob).
__global__ __launch_bounds__(256) void occ_blend_fwd_kernel(const float* __restrict__ dists, const int64_t* __restrict__ p2f,
                                                            long n_pix, int K, float sigma, float* __restrict__ images) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    float prod = 1.0f;
    for (int k = 0; k < K; ++k) {
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        const float prob = m / (1.0f + expf(dists[p * K + k] / sigma));
        prod *= 1.0f - prob;
    }
    reinterpret_cast<float4*>(images)[p] = make_float4(1.0f, 1.0f, 1.0f, 1.0f - prod);
}

// d alpha / d d_k = (prod_{j != k} (1 - prob_j)) * prob_k (1 - prob_k) / sigma   (masked entries carry no gradient);
// the leave-one-out products come from prefix / suffix passes, so a factor (1 - prob_k) = 0 needs no division
__global__ __launch_bounds__(256) void occ_blend_bwd_kernel(const float* __restrict__ dists, const int64_t* __restrict__ p2f,
                                                            const float* __restrict__ grad_images, long n_pix, int K,
                                                            float sigma, float* __restrict__ grad_dists) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float g = grad_images[p * 4 + 3];
    // suffix products into the output buffer first, then a forward sweep with the running prefix
    float suf = 1.0f;
    for (int k = K - 1; k >= 0; --k) {
        grad_dists[p * K + k] = suf;
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        suf *= 1.0f - m / (1.0f + expf(dists[p * K + k] / sigma));
    }
    float pre = 1.0f;
    for (int k = 0; k < K; ++k) {
        const float m = p2f[p * K + k] >= 0 ? 1.0f : 0.0f;
        const float prob = m / (1.0f + expf(dists[p * K + k] / sigma));
        // alpha = 1 - prod(q): d alpha / d prob_k = prod_{j != k} q_j ; d prob_k / d d_k = -prob_k (1 - prob_k) / sigma
        grad_dists[p * K + k] = -g * (pre * grad_dists[p * K + k]) * prob * (1.0f - prob) / sigma * m;
        pre *= 1.0f - prob;
    }
}

// ------------------------------------------------------------------------------------------
// device-side auto-reset: pairing (one block) + commit (one block group per pair)
// ------------------------------------------------------------------------------------------
struct PairArgs {
    const uint8_t* done; const float* loss_all; const int* status;
    int n_env, n_res;
    int* rs_state; int* rs_tries; int* pairs; int* report; int* skip;
};

// ordered compaction helper: exclusive prefix of flag over a 1024-thread block (16 waves)
__device__ __forceinline__ int block_prefix_1024(bool flag, int* s_w, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(flag);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    __syncthreads();  // s_w reuse
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int v = s_w[w];
        if (w < wave) off += v;
        tot += v;
    }
    total = tot;
    return off + __popcll(m & lt);
}

__global__ __launch_bounds__(1024) void occ_pair_kernel(PairArgs a) {
    __shared__ int s_w[16];
    __shared__ int s_fin[512], s_ready[512];
    __shared__ int s_any;
    const int tid = threadIdx.x;
    const int N = a.n_env, R = a.n_res;
    if (tid == 0) s_any = 0;
    // (1) age the PENDING slots: every slot was rendered by this step's launch with its current scene
    int st = OCC_RS_EMPTY;
    if (tid < R) {
        st = a.rs_state[tid];
        if (st == OCC_RS_PENDING) {
            const int t = a.rs_tries[tid] + 1;
            // accept, or keep the 10th try regardless (environment.py:288,327)
            st = (a.loss_all[N + tid] > kDoneThreshold || t >= 10) ? OCC_RS_READY : OCC_RS_EMPTY;
            a.rs_tries[tid] = t;
        }
        a.report[N + R + tid] = -1;
    }
    int nready;
    const int rpos = block_prefix_1024(tid < R && st == OCC_RS_READY, s_w, tid, nready);
    if (tid < R && st == OCC_RS_READY) s_ready[rpos] = tid;
    // (2) finished envs in index order (the first 512 are kept: no more slots than that exist)
    int nfin = 0, any = 0;
    for (int base = 0; base < N; base += 1024) {
        const int i = base + tid;
        const bool f = i < N && a.done[i] != 0;
        if (i < N) {
            a.report[i] = f ? 1 : 0;
            any |= a.status[i];
        }
        int tot;
        const int pos = nfin + block_prefix_1024(f, s_w, tid, tot);
        if (f && pos < 512) s_fin[pos] = i;
        nfin += tot;
    }
    if (tid < R) any |= a.status[N + tid];
    if (any) atomicOr(&s_any, 1);
    __syncthreads();
    // (3) pair them
    const int npair = min(min(nfin, nready), R);
    if (tid < npair) {
        const int r = s_ready[tid], i = s_fin[tid];
        a.pairs[2 + 2 * tid] = i;
        a.pairs[3 + 2 * tid] = N + r;
        a.report[N + R + r] = i;
    }
    // a READY slot that was taken goes back to EMPTY with a fresh try count
    const bool taken = tid < R && st == OCC_RS_READY && rpos < npair;
    if (tid < R) {
        if (taken) {
            st = OCC_RS_EMPTY;
            a.rs_tries[tid] = 0;
        }
        a.rs_state[tid] = st;
        a.report[N + tid] = st;
        a.skip[N + tid] = (st != OCC_RS_PENDING) ? 1 : 0;  // only slots under test are rendered by the next step
    }
    if (tid == 0) {
        a.pairs[0] = npair;
        a.report[N + 2 * R] = s_any;
        a.report[N + 2 * R + 1] = nfin - npair;
    }
}

// rows this step rendered for PENDING slots -> persistent store (runs BEFORE the pairing changes any state)
struct StashArgs {
    const int* rs_state; const float* obs_all; const float* fs_all; const float* loss_all;
    OccReserveStore store;
    int img, n_env;
};
constexpr int kStashBlocks = 16;
__global__ __launch_bounds__(256) void occ_stash_kernel(StashArgs a) {
    const int r = blockIdx.x;
    if (a.rs_state[r] != OCC_RS_PENDING) return;
    const int tid = threadIdx.x, y = blockIdx.y;
    const size_t S2 = (size_t)a.img * a.img, src = (size_t)(a.n_env + r);
    const float4* o4 = reinterpret_cast<const float4*>(a.obs_all + src * 4 * S2);
    const float4* f4 = reinterpret_cast<const float4*>(a.fs_all + src * 4 * S2);
    float4* od = reinterpret_cast<float4*>(a.store.obs + (size_t)r * 4 * S2);
    float4* fd = reinterpret_cast<float4*>(a.store.full_state + (size_t)r * 4 * S2);
    for (size_t i = (size_t)y * 256 + tid; i < S2; i += (size_t)kStashBlocks * 256) {
        od[i] = o4[i];
        fd[i] = f4[i];
    }
    if (y == 0 && tid == 0) a.store.loss[r] = a.loss_all[src];
}

struct AutoCommitArgs {
    const int* pairs;
    OccEnvState st;
    float* obs_all; float* term_obs; const float* res_obs; const float* res_loss;
    int img, n_env;
};
constexpr int kCommitObsBlocks = 8, kCommitAlphaBlocks = 6;
__global__ __launch_bounds__(256) void occ_auto_commit_kernel(AutoCommitArgs a) {
    const int k = blockIdx.x;
    if (k >= a.pairs[0]) return;
    const int dst = a.pairs[2 + 2 * k], src = a.pairs[3 + 2 * k];
    const int tid = threadIdx.x, y = blockIdx.y;
    const size_t S2 = (size_t)a.img * a.img;
    if (y == 0) {
        if (tid == 0) {
            a.st.el[dst] = a.st.el[src];
            a.st.az[dst] = a.st.az[src];
            a.st.radius[dst] = a.st.radius[src];
            const float l = a.res_loss[src - a.n_env];
            a.st.full_reward[dst] = l;
            a.st.object_mass[dst] = l + 1.0f;
        }
        if (tid < 3) {
            a.st.campos[dst * 3 + tid] = 0.f;
            a.st.scene_mesh[dst * 3 + tid] = a.st.scene_mesh[src * 3 + tid];
        }
        if (tid < 9) a.st.scene_offset[dst * 9 + tid] = a.st.scene_offset[src * 9 + tid];
        if (tid < OCC_CAM_STRIDE) a.st.cam[(size_t)dst * OCC_CAM_STRIDE + tid] = a.st.cam[(size_t)src * OCC_CAM_STRIDE + tid];
    } else if (y <= kCommitObsBlocks) {
        // final observation -> term_obs[slot], stored reset observation -> obs[env] (same element range, same thread)
        const float4* s4 = reinterpret_cast<const float4*>(a.res_obs + (size_t)(src - a.n_env) * 4 * S2);
        float4* d4 = reinterpret_cast<float4*>(a.obs_all + (size_t)dst * 4 * S2);
        float4* t4 = reinterpret_cast<float4*>(a.term_obs + (size_t)(src - a.n_env) * 4 * S2);
        for (size_t i = (size_t)(y - 1) * 256 + tid; i < S2; i += (size_t)kCommitObsBlocks * 256) {
            t4[i] = d4[i];
            d4[i] = s4[i];
        }
    } else {
        const float* s1 = a.st.alphas + (size_t)src * 3 * S2;
        float* d1 = a.st.alphas + (size_t)dst * 3 * S2;
        for (size_t i = (size_t)(y - 1 - kCommitObsBlocks) * 256 + tid; i < 3 * S2; i += (size_t)kCommitAlphaBlocks * 256)
            d1[i] = s1[i];
    }
}

__global__ __launch_bounds__(64) void occ_refill_kernel(const int* __restrict__ packed, int n, int n_env, int n_res,
                                                        int* __restrict__ scene_mesh, float* __restrict__ scene_offset,
                                                        int* __restrict__ rs_state, int* __restrict__ skip) {
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= n) return;
    const int* row = packed + 13 * k;
    const int slot = row[0];
    if (slot < 0 || slot >= n_res) return;
    const int e = n_env + slot;
    if (tid < 3) scene_mesh[e * 3 + tid] = row[1 + tid];
    if (tid < 9) scene_offset[e * 9 + tid] = __int_as_float(row[4 + tid]);
    if (tid == 0) {
        rs_state[slot] = OCC_RS_PENDING;
        skip[e] = 0;  // rendered from the next launch on
    }
}


// ------------------------------------------------------------------------------------------
// Operator-level replacement of PyTorch3D's _C.rasterize_meshes / _C.rasterize_meshes_backward
// (naive path, bin_size = 0): K-buffer outputs in PyTorch3D's layout.  The fused step() above never
// materialises these; this pair exists for callers of the rasteriser itself (SURVEY.md §8b lower surface)
// and is written for exactness, not speed: one thread per pixel, all faces of its mesh, replace-the-farthest
// K list kept directly in the output arrays, bubble sort at the end - the structure of upstream's naive
// CUDA kernel.  No FMA contraction / reciprocal shortcuts: the arithmetic order is the one of SURVEY A.4.
// ------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float k_edge(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
__device__ __forceinline__ float k_seg(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= kEpsilon) return (px - bx) * (px - bx) + (py - by) * (py - by);
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = ax + t * bax - px, qy = ay + t * bay - py;
    return qx * qx + qy * qy;
}

struct KbufArgs {
    const float* face_verts;      // (F,3,3)
    const int64_t* first_idx;     // (N)
    const int64_t* num_faces;     // (N)
    const int64_t* neighbor;      // (F) or null
    int N, H, W, K;
    float blur;
    int persp, clipb, cull;
    int64_t* p2f;  // (N,H,W,K)
    float* zbuf;   // (N,H,W,K)
    float* bary;   // (N,H,W,K,3)
    float* dists;  // (N,H,W,K)
};

__global__ __launch_bounds__(64) void occ_rast_naive_fwd_kernel(KbufArgs a) {
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long npix = (long)a.N * a.H * a.W;
    if (pix >= npix) return;
    const int n = (int)(pix / ((long)a.H * a.W));
    const int rem = (int)(pix - (long)n * a.H * a.W);
    const int yi = rem / a.W, xi = rem - yi * a.W;
    const float yf = -1.0f + (2.0f * (float)(a.H - 1 - yi) + 1.0f) / (float)a.H;
    const float xf = -1.0f + (2.0f * (float)(a.W - 1 - xi) + 1.0f) / (float)a.W;
    const float sqb = sqrtf(a.blur);
    const int K = a.K;
    int64_t* qf = a.p2f + pix * K;
    float* qz = a.zbuf + pix * K;
    float* qd = a.dists + pix * K;
    float* qb = a.bary + pix * K * 3;
    int qn = 0;
    const int64_t f0 = a.first_idx[n], f1 = f0 + a.num_faces[n];
    for (int64_t f = f0; f < f1; ++f) {
        const float* v = a.face_verts + f * 9;
        const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
        const float area = k_edge(x0, y0, x1, y1, x2, y2);
        if (a.cull && area < 0.0f) continue;
        if (area <= kEpsilon && area >= -kEpsilon) continue;
        if (fmaxf(fmaxf(z0, z1), z2) < 0.0f) continue;
        const float xmin = fminf(fminf(x0, x1), x2) - sqb, xmax = fmaxf(fmaxf(x0, x1), x2) + sqb;
        const float ymin = fminf(fminf(y0, y1), y2) - sqb, ymax = fmaxf(fmaxf(y0, y1), y2) + sqb;
        if (!((xmin <= xf && xf <= xmax) && (ymin <= yf && yf <= ymax))) continue;
        const float ar = k_edge(x2, y2, x0, y0, x1, y1) + kEpsilon;
        const float b0 = k_edge(xf, yf, x1, y1, x2, y2) / ar;
        const float b1 = k_edge(xf, yf, x2, y2, x0, y0) / ar;
        const float b2 = k_edge(xf, yf, x0, y0, x1, y1) / ar;
        float p0 = b0, p1 = b1, p2 = b2;
        if (a.persp) {
            const float w0 = b0 * z1 * z2, w1 = z0 * b1 * z2, w2 = z0 * z1 * b2;
            const float den = fmaxf(w0 + w1 + w2, kEpsilon);
            p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
        }
        float c0 = p0, c1 = p1, c2 = p2;
        if (a.clipb) {
            c0 = fmaxf(p0, 0.0f); c1 = fmaxf(p1, 0.0f); c2 = fmaxf(p2, 0.0f);
            const float sm = fmaxf(c0 + c1 + c2, kBaryClipMin);
            c0 /= sm; c1 /= sm; c2 /= sm;
        }
        const float pz = c0 * z0 + c1 * z1 + c2 * z2;
        if (pz < 0.0f) continue;
        const float e01 = k_seg(xf, yf, x0, y0, x1, y1), e02 = k_seg(xf, yf, x0, y0, x2, y2), e12 = k_seg(xf, yf, x1, y1, x2, y2);
        const float dist = fminf(fminf(e01, e02), e12);
        const int amin = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : 2);
        const bool inside = p0 > 0.0f && p1 > 0.0f && p2 > 0.0f;
        if (!inside && dist >= a.blur) continue;
        const float sd = inside ? -dist : dist;
        // clipped-pair rule, incl. the shared-diagonal tie definition of DESIGN.md §2
        int itop = -1;
        const int64_t nb = a.neighbor ? a.neighbor[f] : -1;
        if (nb != -1) {
            for (int i = 0; i < qn; ++i)
                if (qf[i] == nb) { itop = i; break; }
        }
        int slot = -1;
        if (itop != -1) {
            // closest edge of the entry already in the list: recompute from its face
            const float* u = a.face_verts + nb * 9;
            const float g01 = k_seg(xf, yf, u[0], u[1], u[3], u[4]), g02 = k_seg(xf, yf, u[0], u[1], u[6], u[7]),
                        g12 = k_seg(xf, yf, u[3], u[4], u[6], u[7]);
            const int amin_nb = (g01 <= g02 && g01 <= g12) ? 0 : ((g02 <= g01 && g02 <= g12) ? 1 : 2);
            const bool shared_tie = (nb == f - 1 && amin_nb == 2 && amin == 0) || (nb == f + 1 && amin_nb == 0 && amin == 2);
            if (!shared_tie && dist < fabsf(qd[itop])) slot = itop;
        } else if (qn < K) {
            slot = qn++;
        } else {
            // full: the candidate displaces the largest (z, f) entry if it is smaller
            int im = 0;
            for (int i = 1; i < K; ++i)
                if (qz[i] > qz[im] || (qz[i] == qz[im] && qf[i] > qf[im])) im = i;
            if (pz < qz[im] || (pz == qz[im] && f < qf[im])) slot = im;
        }
        if (slot >= 0) {
            qf[slot] = f; qz[slot] = pz; qd[slot] = sd;
            qb[slot * 3] = c0; qb[slot * 3 + 1] = c1; qb[slot * 3 + 2] = c2;
        }
    }
    // ascending (z, f); empty slots = -1
    for (int i = 0; i < qn - 1; ++i)
        for (int j = 0; j < qn - 1 - i; ++j)
            if (qz[j] > qz[j + 1] || (qz[j] == qz[j + 1] && qf[j] > qf[j + 1])) {
                const int64_t tf = qf[j]; qf[j] = qf[j + 1]; qf[j + 1] = tf;
                float t = qz[j]; qz[j] = qz[j + 1]; qz[j + 1] = t;
                t = qd[j]; qd[j] = qd[j + 1]; qd[j + 1] = t;
#pragma unroll
                for (int c = 0; c < 3; ++c) { t = qb[j * 3 + c]; qb[j * 3 + c] = qb[(j + 1) * 3 + c]; qb[(j + 1) * 3 + c] = t; }
            }
    for (int i = qn; i < K; ++i) {
        qf[i] = -1; qz[i] = -1.0f; qd[i] = -1.0f;
        qb[i * 3] = qb[i * 3 + 1] = qb[i * 3 + 2] = -1.0f;
    }
}

// dists part of RasterizeMeshesBackward: one thread per (pixel, k), atomicAdd into grad_face_verts
__global__ __launch_bounds__(256) void occ_rast_naive_bwd_kernel(const float* __restrict__ face_verts,
                                                                 const int64_t* __restrict__ p2f,
                                                                 const float* __restrict__ grad_dists, int N, int H, int W,
                                                                 int K, int persp, int clipb,
                                                                 float* __restrict__ grad_face_verts) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long tot = (long)N * H * W * K;
    if (t >= tot) return;
    const int64_t f = p2f[t];
    if (f < 0) return;
    const long pix = t / K;
    const int rem = (int)(pix % ((long)H * W));
    const int yi = rem / W, xi = rem - yi * W;
    const float yf = -1.0f + (2.0f * (float)(H - 1 - yi) + 1.0f) / (float)H;
    const float xf = -1.0f + (2.0f * (float)(W - 1 - xi) + 1.0f) / (float)W;
    const float* v = face_verts + f * 9;
    const float x0 = v[0], y0 = v[1], z0 = v[2], x1 = v[3], y1 = v[4], z1 = v[5], x2 = v[6], y2 = v[7], z2 = v[8];
    const float ar = k_edge(x2, y2, x0, y0, x1, y1) + kEpsilon;
    float p0 = k_edge(xf, yf, x1, y1, x2, y2) / ar, p1 = k_edge(xf, yf, x2, y2, x0, y0) / ar, p2 = k_edge(xf, yf, x0, y0, x1, y1) / ar;
    if (persp) {
        const float w0 = p0 * z1 * z2, w1 = z0 * p1 * z2, w2 = z0 * z1 * p2;
        const float den = fmaxf(w0 + w1 + w2, kEpsilon);
        p0 = w0 / den; p1 = w1 / den; p2 = w2 / den;
    }
    if (clipb) { p0 = fmaxf(p0, 0.f); p1 = fmaxf(p1, 0.f); p2 = fmaxf(p2, 0.f); }  // the sign test below is all that matters
    const bool inside = p0 > 0.0f && p1 > 0.0f && p2 > 0.0f;
    const float g = (inside ? -1.0f : 1.0f) * grad_dists[t];
    const float e01 = k_seg(xf, yf, x0, y0, x1, y1), e02 = k_seg(xf, yf, x0, y0, x2, y2), e12 = k_seg(xf, yf, x1, y1, x2, y2);
    int ia, ib;
    if (e01 <= e02 && e01 <= e12) { ia = 0; ib = 1; }
    else if (e02 <= e01 && e02 <= e12) { ia = 0; ib = 2; }
    else if (e12 <= e01 && e12 <= e02) { ia = 1; ib = 2; }
    else return;
    const float ax = v[ia * 3], ay = v[ia * 3 + 1], bx = v[ib * 3], by = v[ib * 3 + 1];
    const float bax = bx - ax, bay = by - ay;
    float tt = (bax * (xf - ax) + bay * (yf - ay)) / (bax * bax + bay * bay + kEpsilon);
    tt = fminf(fmaxf(tt, 0.0f), 1.0f);
    const float dx = (1.0f - tt) * ax + tt * bx - xf, dy = (1.0f - tt) * ay + tt * by - yf;
    float* gf = grad_face_verts + f * 9;
    atomicAdd(gf + ia * 3, g * (1.0f - tt) * 2.0f * dx);
    atomicAdd(gf + ia * 3 + 1, g * (1.0f - tt) * 2.0f * dy);
    atomicAdd(gf + ib * 3, g * tt * 2.0f * dx);
    atomicAdd(gf + ib * 3 + 1, g * tt * 2.0f * dy);
}
#pragma clang fp contract(fast)

}  // namespace occ

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace occ;

extern "C" int occ_abi_version(void) { return OCC_ABI_VERSION; }

// ---- measurement hooks ---------------------------------------------------------------------
namespace {
constexpr int kProfMax = 4096;
bool g_prof_on = false;
int g_prof_n = 0;
hipEvent_t g_prof_ev[2 * kProfMax];
int g_prof_nenv[kProfMax];
bool g_prof_created = false;
}  // namespace

extern "C" int occ_profile_enable(int on) {
    if (on && !g_prof_created) {
        for (int i = 0; i < 2 * kProfMax; ++i)
            if (hipEventCreate(&g_prof_ev[i]) != hipSuccess) return OCC_ERR_LAUNCH;
        g_prof_created = true;
    }
    g_prof_on = on != 0;
    g_prof_n = 0;
    return OCC_OK;
}

extern "C" int occ_profile_read(double* ms_sum, int* launches) {
    if (!ms_sum || !launches) return OCC_ERR_ARG;
    double tot = 0.0;
    int big = 0, cnt = 0;
    for (int i = 0; i < g_prof_n; ++i) big = g_prof_nenv[i] > big ? g_prof_nenv[i] : big;
    for (int i = 0; i < g_prof_n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof_ev[2 * i + 1]) != hipSuccess) return OCC_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]) != hipSuccess) return OCC_ERR_LAUNCH;
        if (g_prof_nenv[i] != big) continue;  // full-batch launches only (auto-resets render tiny batches)
        tot += ms;
        cnt += 1;
    }
    *ms_sum = tot;
    *launches = cnt;
    g_prof_n = 0;
    return OCC_OK;
}

#ifdef OCC_DBG_STATS
extern "C" int occ_debug_stats(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(occ::g_dbg_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(occ::g_dbg_stats), z, sizeof(z)) == hipSuccess ? 0 : 2;
}
#endif

#ifdef OCC_DBG_BOUNDS
extern "C" int occ_debug_fault(int* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(occ::g_dbg_fault), 8 * sizeof(int)) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int occ_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}

static bool scene_ok(const OccScene* s) {
    return s && s->pool_verts && s->pool_faces && s->mesh_vert_off && s->mesh_face_off && s->scene_mesh &&
           s->scene_offset && s->n_env > 0 && s->img >= OCC_TILE && s->img % OCC_TILE == 0 && s->img <= 2048 &&
           s->rec_cap > 0;
}

extern "C" int occ_workspace_query(const OccScene* scene, int n_slots, OccWorkspaceSizes* out) {
    if (!scene || !out || scene->n_env <= 0 || scene->img < OCC_TILE || scene->img % OCC_TILE || scene->rec_cap <= 0)
        return OCC_ERR_ARG;
    if (n_slots <= 0) {
        const int cus = occ_device_cu_count();
        n_slots = (cus > 0 ? cus : 256) * 8;
    }
    const size_t N = (size_t)scene->n_env, cap = (size_t)scene->rec_cap;
    out->rec_bytes = N * 3 * cap * OCC_REC_STRIDE * sizeof(float);
    out->rec_bbox_bytes = N * 3 * cap * 4 * sizeof(uint32_t);
    out->scan_bytes = N * 3 * cap * 4 * sizeof(uint32_t);
    out->rec_cbox_bytes = N * 3 * ((cap + 63) / 64) * 4 * sizeof(uint32_t);
    out->nrec_bytes = N * 3 * sizeof(int32_t);
    out->objrect_bytes = N * 3 * 4 * sizeof(int32_t);
    out->queue_bytes = 8 * 16 * sizeof(uint32_t);  // eight queue heads, one 64-B line each
    out->lists_bytes = (size_t)n_slots * OCC_LIST_CAP * 64 * 4 * sizeof(float);
    const size_t S2 = (size_t)scene->img * scene->img;
    out->partials_bytes = N * ((S2 + 255) / 256) * 4 * sizeof(float);
    out->offsets_bytes = (size_t)(8 * xcd_slots(scene->n_env) + 1) * sizeof(int32_t);
    out->obj_alpha_bytes = N * 3 * S2 * sizeof(float);
    out->obj_grad_bytes = N * 3 * S2 * 2 * sizeof(float);
    out->obj_hz_bytes = N * 3 * S2 * sizeof(float);
    out->obj_hrec_bytes = N * 3 * S2 * sizeof(int32_t);
    out->status_bytes = N * sizeof(int32_t);
    out->n_slots = n_slots;
    return OCC_OK;
}

extern "C" int occ_camera(int mode, const float* action, float* el, float* az, const float* radius, float* cam,
                          float* cam_pos_out, int n_env, void* stream) {
    if (!cam || n_env <= 0) return OCC_ERR_ARG;
    if (mode == OCC_CAM_STEP && (!action || !el || !az || !radius)) return OCC_ERR_ARG;
    if (mode == OCC_CAM_LOOKAT && (!el || !az || !radius)) return OCC_ERR_ARG;
    if (mode == OCC_CAM_POSITION && !action) return OCC_ERR_ARG;
    if (mode < 0 || mode > 2) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_camera_kernel, dim3((n_env + 63) / 64), dim3(64), 0, (hipStream_t)stream, mode, action, el, az,
                       radius, cam, cam_pos_out, n_env);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

// OCC_DEBUG_SYNC=1 in the environment: every launch of occ_render is announced on stderr and waited for, so that a
// faulting kernel is the last one named (diagnostics only; serialises the stream).
static bool dbg_sync_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("OCC_DEBUG_SYNC");
        on = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return on == 1;
}
#define OCC_DBG_SYNC(name)                                                          \
    do {                                                                            \
        if (dbg_sync_on()) {                                                        \
            fprintf(stderr, "[occ] %s launched (n_env %d) ...", name, scene->n_env); \
            fflush(stderr);                                                         \
            const hipError_t e_ = hipStreamSynchronize(st);                         \
            fprintf(stderr, " %s\n", e_ == hipSuccess ? "ok" : hipGetErrorString(e_)); \
            fflush(stderr);                                                         \
        }                                                                           \
    } while (0)

extern "C" int occ_render(const OccScene* scene, const float* cam, const OccWorkspace* ws, const OccRenderOut* out,
                          int flags, int faces_per_pixel, void* stream) {
    if (!scene_ok(scene) || !cam || !ws || !out) return OCC_ERR_ARG;
    if (!ws->rec || !ws->rec_bbox || !ws->nrec || !ws->objrect || !ws->queue || !ws->lists || !ws->partials ||
        !ws->status || !ws->rec_cbox || !ws->scan || !ws->offsets || !ws->obj_alpha || !ws->obj_grad || !ws->obj_hz || !ws->obj_hrec ||
        ws->n_slots <= 0)
        return OCC_ERR_ARG;
    const bool soft = flags & OCC_RENDER_SOFT, hard = flags & OCC_RENDER_HARD, grad = flags & OCC_RENDER_GRAD;
    if (!soft && !hard) return OCC_ERR_ARG;
    if (grad && !soft) return OCC_ERR_ARG;
    if (hard && !out->obs) return OCC_ERR_ARG;
    if (soft && (faces_per_pixel <= 0 || faces_per_pixel > OCC_MAX_K)) return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws->queue, 0, 8 * 16 * sizeof(uint32_t), st) != hipSuccess) return OCC_ERR_LAUNCH;
    const int N = scene->n_env;
    if (grad)
        hipLaunchKernelGGL(occ_setup_kernel<true>, dim3(N * 3), dim3(256), 0, st, *scene, cam, *ws);
    else
        hipLaunchKernelGGL(occ_setup_kernel<false>, dim3(N * 3), dim3(256), 0, st, *scene, cam, *ws);
    OCC_DBG_SYNC("setup");
    if (scene->rec_cap >= kSortMin) {
        // dense objects only: front-to-back scan order (LDS sort buffer: 8192 keys = 64 KiB)
        const int sort_cap = 8192;
        hipLaunchKernelGGL(occ_sort_kernel, dim3(N * 3), dim3(256), (size_t)sort_cap * sizeof(unsigned long long), st,
                           *scene, *ws, sort_cap);
        OCC_DBG_SYNC("sort");
    }
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    RasterParams P;
    P.sc = *scene;
    P.ws = *ws;
    P.out = *out;
    P.cam = cam;
    P.K = faces_per_pixel;
    P.ntx = scene->img / OCC_TILE;
    hipLaunchKernelGGL(occ_scan_kernel, dim3(1), dim3(1024), 0, st, ws->objrect, ws->nrec, ws->offsets, N);
    OCC_DBG_SYNC("scan");
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    const dim3 grid(ws->n_slots), block(64);
    const bool prof = g_prof_on && g_prof_n < kProfMax;
    if (prof) (void)hipEventRecord(g_prof_ev[2 * g_prof_n], st);
    const int bpe = (scene->img * scene->img + 255) / 256;
    const dim3 cgrid(N * bpe), cblock(256);
#define OCC_LAUNCH(SOFT_, HARD_, GRAD_)                                                             \
    do {                                                                                            \
        hipLaunchKernelGGL((occ_raster_kernel<SOFT_, HARD_, GRAD_>), grid, block, 0, st, P);        \
        OCC_DBG_SYNC("raster");                                                                     \
        if (prof) {                                                                                 \
            (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st);                                  \
            g_prof_nenv[g_prof_n] = N;                                                              \
            g_prof_n += 1;                                                                          \
        }                                                                                           \
        hipLaunchKernelGGL((occ_combine_kernel<SOFT_, HARD_, GRAD_>), cgrid, cblock, 0, st, P, bpe); \
        OCC_DBG_SYNC("combine");                                                                    \
    } while (0)
    if (soft && hard && grad)
        OCC_LAUNCH(true, true, true);
    else if (soft && hard)
        OCC_LAUNCH(true, true, false);
    else if (soft && grad)
        OCC_LAUNCH(true, false, true);
    else if (soft)
        OCC_LAUNCH(true, false, false);
    else
        OCC_LAUNCH(false, true, false);
#undef OCC_LAUNCH
    if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    if (soft && (out->loss || out->grad_elaz)) {
        hipLaunchKernelGGL(occ_reduce_kernel, dim3(N), dim3(64), 0, st, ws->partials, bpe, out->loss,
                           grad ? out->grad_elaz : nullptr, scene->skip);
        if (hipGetLastError() != hipSuccess) return OCC_ERR_LAUNCH;
    }
    return OCC_OK;
}

extern "C" int occ_rasterize_meshes_naive(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                          const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx,
                                          int n_meshes, int H, int W, float blur_radius, int faces_per_pixel,
                                          int perspective_correct, int clip_barycentric_coords, int cull_backfaces,
                                          int64_t* pix_to_face, float* zbuf, float* bary, float* dists, void* stream) {
    if (!face_verts || !mesh_to_face_first_idx || !num_faces_per_mesh || !pix_to_face || !zbuf || !bary || !dists ||
        n_meshes <= 0 || H <= 0 || W <= 0 || faces_per_pixel <= 0 || blur_radius < 0.f)
        return OCC_ERR_ARG;
    KbufArgs a{face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, n_meshes, H, W,
               faces_per_pixel, blur_radius, perspective_correct, clip_barycentric_coords, cull_backfaces, pix_to_face,
               zbuf, bary, dists};
    const long npix = (long)n_meshes * H * W;
    hipLaunchKernelGGL(occ_rast_naive_fwd_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_rasterize_meshes_backward_dists(const float* face_verts, const int64_t* pix_to_face,
                                                   const float* grad_dists, int64_t n_faces, int n_meshes, int H, int W,
                                                   int faces_per_pixel, int perspective_correct,
                                                   int clip_barycentric_coords, float* grad_face_verts, void* stream) {
    if (!face_verts || !pix_to_face || !grad_dists || !grad_face_verts || n_faces <= 0 || n_meshes <= 0 || H <= 0 ||
        W <= 0 || faces_per_pixel <= 0)
        return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_face_verts, 0, (size_t)n_faces * 9 * sizeof(float), st) != hipSuccess) return OCC_ERR_LAUNCH;
    const long tot = (long)n_meshes * H * W * faces_per_pixel;
    hipLaunchKernelGGL(occ_rast_naive_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, face_verts,
                       pix_to_face, grad_dists, n_meshes, H, W, faces_per_pixel, perspective_correct,
                       clip_barycentric_coords, grad_face_verts);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_sigmoid_alpha_blend_fwd(const float* dists, const int64_t* pix_to_face, int64_t n_pix, int faces_per_pixel,
                                           float sigma, float* images, void* stream) {
    if (!dists || !pix_to_face || !images || n_pix <= 0 || faces_per_pixel <= 0 || !(sigma > 0.f)) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_blend_fwd_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dists,
                       pix_to_face, (long)n_pix, faces_per_pixel, sigma, images);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_sigmoid_alpha_blend_bwd(const float* dists, const int64_t* pix_to_face, const float* grad_images,
                                           int64_t n_pix, int faces_per_pixel, float sigma, float* grad_dists, void* stream) {
    if (!dists || !pix_to_face || !grad_images || !grad_dists || n_pix <= 0 || faces_per_pixel <= 0 || !(sigma > 0.f))
        return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_blend_bwd_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dists,
                       pix_to_face, grad_images, (long)n_pix, faces_per_pixel, sigma, grad_dists);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_step_flags(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                              int32_t* flags, void* stream) {
    if (!done || !status || !flags || n_env <= 0 || n_reserve < 0 || (n_reserve > 0 && !loss_all)) return OCC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int tot = n_env + n_reserve;
    hipLaunchKernelGGL(occ_flags_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, done, loss_all, status, n_env, n_reserve,
                       flags);
    hipLaunchKernelGGL(occ_status_any_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, status, tot, flags + tot);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_reset_commit(const int32_t* pairs, int n, float* el, float* az, float* radius, float* campos, float* cam,
                                float* alphas, float* full_reward, float* object_mass, int32_t* scene_mesh,
                                float* scene_offset, float* obs, const float* obs_all, const float* loss_all, int img,
                                void* stream) {
    if (n == 0) return OCC_OK;
    if (!pairs || n < 0 || !el || !az || !radius || !campos || !cam || !alphas || !full_reward || !object_mass ||
        !scene_mesh || !scene_offset || !obs || !obs_all || !loss_all || img <= 0)
        return OCC_ERR_ARG;
    CommitArgs a{pairs, n, el, az, radius, campos, cam, alphas, full_reward, object_mass, scene_mesh, scene_offset,
                 obs, obs_all, loss_all, img};
    hipLaunchKernelGGL(occ_commit_kernel, dim3(n, 3), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_auto_reset(const uint8_t* done, const float* loss_all, const int32_t* status, int n_env, int n_reserve,
                              int32_t* rs_state, int32_t* rs_tries, const OccEnvState* st, float* obs_all,
                              const float* full_state_all, const OccReserveStore* store, float* term_obs, int img,
                              int32_t* pairs, int32_t* report, void* stream) {
    if (!done || !loss_all || !status || n_env <= 0 || n_reserve <= 0 || n_reserve > 512 || !rs_state || !rs_tries || !st ||
        !obs_all || !full_state_all || !store || !term_obs || img <= 0 || !pairs || !report)
        return OCC_ERR_ARG;
    if (!st->el || !st->az || !st->radius || !st->campos || !st->cam || !st->alphas || !st->full_reward || !st->object_mass ||
        !st->scene_mesh || !st->scene_offset || !store->obs || !store->full_state || !store->loss || !store->skip)
        return OCC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    StashArgs sa{rs_state, obs_all, full_state_all, loss_all, *store, img, n_env};
    hipLaunchKernelGGL(occ_stash_kernel, dim3(n_reserve, kStashBlocks), dim3(256), 0, s, sa);
    PairArgs pa{done, loss_all, status, n_env, n_reserve, rs_state, rs_tries, pairs, report, store->skip};
    hipLaunchKernelGGL(occ_pair_kernel, dim3(1), dim3(1024), 0, s, pa);
    AutoCommitArgs ca{pairs, *st, obs_all, term_obs, store->obs, store->loss, img, n_env};
    hipLaunchKernelGGL(occ_auto_commit_kernel, dim3(n_reserve, 1 + kCommitObsBlocks + kCommitAlphaBlocks), dim3(256), 0, s, ca);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_reserve_refill(const int32_t* packed, int n, int n_env, int n_reserve, int32_t* scene_mesh,
                                  float* scene_offset, int32_t* rs_state, int32_t* skip, void* stream) {
    if (n == 0) return OCC_OK;
    if (!packed || n < 0 || n_env <= 0 || n_reserve <= 0 || !scene_mesh || !scene_offset || !rs_state || !skip) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_refill_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, packed, n, n_env, n_reserve, scene_mesh,
                       scene_offset, rs_state, skip);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}

extern "C" int occ_step_finish(const float* loss, const float* grad_elaz, const float* cam, float* full_reward,
                               const float* object_mass, float* reward, uint8_t* done, float* grad_action, int n_env,
                               void* stream) {
    if (!loss || !cam || !full_reward || !object_mass || !reward || !done || n_env <= 0) return OCC_ERR_ARG;
    hipLaunchKernelGGL(occ_finish_kernel, dim3((n_env + 63) / 64), dim3(64), 0, (hipStream_t)stream, loss, grad_elaz, cam,
                       full_reward, object_mass, reward, done, grad_action, n_env);
    return hipGetLastError() == hipSuccess ? OCC_OK : OCC_ERR_LAUNCH;
}
