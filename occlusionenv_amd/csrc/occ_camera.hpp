// occ_camera.hpp -- camera kernel: action -> (el, az) -> look_at R, T with forward-mode tangents.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// camera: dual numbers with two tangent directions (d/d el, d/d az), in double precision
// ------------------------------------------------------------------------------------------
// The camera is evaluated in DOUBLE precision and rounded to f32 once, at the store (round 5).  One thread per env
// runs this, so the cost is nil - and it matters: R and T move every vertex of the scene TOGETHER, so an error in them does
// not average out over faces the way per-vertex rounding does.  With f32 duals T_z came out one ulp low (2.49999976 for
// a radius of 2.5, where -R^T C = (0, 0, |C|) exactly); at radius 2.5 with the camera inside the scene that one ulp moved
// d reward / d action by 2.5e-4 of its norm (scripts/dbg/emul_engine_records.py: the f64 oracle's gradient evaluated on
// the engine's own records reproduced the engine's excess, its tangents alone none of it).  The STATE (el, az) stays
// f32 like the reference's tensors (environment.py:360-361).
template <class F>
struct D2T {
    F v, e, a;
};
using D2 = D2T<double>;
__device__ __forceinline__ D2 dconst(double v) { return {v, 0.0, 0.0}; }
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return {x.v + y.v, x.e + y.e, x.a + y.a}; }
__device__ __forceinline__ D2 operator-(D2 x, D2 y) { return {x.v - y.v, x.e - y.e, x.a - y.a}; }
__device__ __forceinline__ D2 operator-(D2 x) { return {-x.v, -x.e, -x.a}; }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) {
    return {x.v * y.v, x.e * y.v + x.v * y.e, x.a * y.v + x.v * y.a};
}
__device__ __forceinline__ D2 operator/(D2 x, D2 y) {
    const double q = x.v / y.v;
    return {q, (x.e - q * y.e) / y.v, (x.a - q * y.a) / y.v};
}
__device__ __forceinline__ D2 dsin(D2 x) {
    const double s = sin(x.v), c = cos(x.v);
    return {s, c * x.e, c * x.a};
}
__device__ __forceinline__ D2 dcos(D2 x) {
    const double s = sin(x.v), c = cos(x.v);
    return {c, -s * x.e, -s * x.a};
}
// F.normalize(v, eps): v / max(||v||, eps)    [P3D look_at_rotation, SURVEY A.1]
__device__ __forceinline__ void dnormalize3(D2* v, double eps) {
    const D2 n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double n = sqrt(n2.v);
    D2 nn;
    if (n > eps) {
        const double h = 0.5 / n;
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

// camera of env n (one thread); pos2: optional second copy of C (the step's position snapshot)
__device__ __forceinline__ void camera_one(int mode, const float* __restrict__ action, float* __restrict__ el_io,
                                           float* __restrict__ az_io, const float* __restrict__ radius,
                                           float* __restrict__ cam, float* __restrict__ cam_pos_out,
                                           float* __restrict__ pos2, int n) {
    float* c = cam + (size_t)n * OCC_CAM_STRIDE;
    D2 C[3];
    double J[4] = {0.0, 0.0, 0.0, 0.0};
    float el_new = 0.f, az_new = 0.f;
    if (mode == OCC_CAM_STEP) {
        // environment.py:356-365
        const float a0 = action[2 * n], a1 = action[2 * n + 1];
        // the state update itself in f32, like the reference's tensors: torch.norm, a / norm, el += n0 * 0.05
        const float nrm = sqrtf(a0 * a0 + a1 * a1);
        float n0 = a0, n1 = a1;
        double j00 = 1.0, j01 = 0.0, j10 = 0.0, j11 = 1.0;  // d n_i / d a_j
        if (nrm != 0.0f) {
            n0 = a0 / nrm;
            n1 = a1 / nrm;
            const double dn = sqrt((double)a0 * a0 + (double)a1 * a1), d0 = a0 / dn, d1 = a1 / dn;
            j00 = (1.0 - d0 * d0) / dn;
            j01 = (-d0 * d1) / dn;
            j10 = j01;
            j11 = (1.0 - d1 * d1) / dn;
        }
        el_new = el_io[n] + n0 * kStepSize;
        az_new = az_io[n] + n1 * kStepSize;
        el_io[n] = el_new;
        az_io[n] = az_new;
        J[0] = (double)kStepSize * j00;
        J[1] = (double)kStepSize * j01;
        J[2] = (double)kStepSize * j10;
        J[3] = (double)kStepSize * j11;
        const D2 el = {(double)el_new, 1.0, 0.0}, az = {(double)az_new, 0.0, 1.0};
        const D2 r = dconst((double)radius[n]);
        const D2 rs = r * dsin(az);
        C[0] = rs * dcos(el);
        C[1] = rs * dsin(el);
        C[2] = r * dcos(az);
    } else if (mode == OCC_CAM_LOOKAT) {
        // environment.py:308 -> [P3D] camera_position_from_spherical_angles(degrees=False)
        el_new = el_io[n];
        az_new = az_io[n];
        const double r = radius[n], de = el_new, da = az_new;
        C[0] = dconst(r * cos(de) * sin(da));
        C[1] = dconst(r * sin(de));
        C[2] = dconst(r * cos(de) * cos(da));
    } else {
        C[0] = dconst((double)action[3 * n]);
        C[1] = dconst((double)action[3 * n + 1]);
        C[2] = dconst((double)action[3 * n + 2]);
    }
    // [P3D] look_at_rotation(C, at=0, up=+Y)
    D2 z[3] = {-C[0], -C[1], -C[2]};
    dnormalize3(z, (double)kLookAtEps);
    const D2 up[3] = {dconst(0.0), dconst(1.0), dconst(0.0)};
    D2 x[3], y[3];
    dcross(up, z, x);
    dnormalize3(x, (double)kLookAtEps);
    dcross(z, x, y);
    dnormalize3(y, (double)kLookAtEps);
    if (fabs(x[0].v) <= (double)kLookAtClose && fabs(x[1].v) <= (double)kLookAtClose && fabs(x[2].v) <= (double)kLookAtClose) {
        dcross(y, z, x);
        dnormalize3(x, (double)kLookAtEps);
    }
    // R[i][j]: columns are x, y, z ; T = -R^T C
    const D2* ax[3] = {x, y, z};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const D2 t = -(ax[j][0] * C[0] + ax[j][1] * C[1] + ax[j][2] * C[2]);
        c[C_T + j] = (float)t.v;
        c[C_DT_EL + j] = (float)t.e;
        c[C_DT_AZ + j] = (float)t.a;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c[C_R + i * 3 + j] = (float)ax[j][i].v;
            c[C_DR_EL + i * 3 + j] = (float)ax[j][i].e;
            c[C_DR_AZ + i * 3 + j] = (float)ax[j][i].a;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) c[C_C + i] = (float)C[i].v;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[C_J + i] = (float)J[i];
    c[C_EL] = el_new;
    c[C_AZ] = az_new;
    c[45] = c[46] = c[47] = 0.f;
    if (cam_pos_out) {
        cam_pos_out[3 * n] = (float)C[0].v;
        cam_pos_out[3 * n + 1] = (float)C[1].v;
        cam_pos_out[3 * n + 2] = (float)C[2].v;
    }
    if (pos2) {
        pos2[3 * n] = (float)C[0].v;
        pos2[3 * n + 1] = (float)C[1].v;
        pos2[3 * n + 2] = (float)C[2].v;
    }
}

__global__ __launch_bounds__(64) void occ_camera_kernel(int mode, const float* __restrict__ action,
                                                        float* __restrict__ el_io, float* __restrict__ az_io,
                                                        const float* __restrict__ radius, float* __restrict__ cam,
                                                        float* __restrict__ cam_pos_out, int n_env) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_env) return;
    camera_one(mode, action, el_io, az_io, radius, cam, cam_pos_out, nullptr, n);
}
