// occ_camera.hpp -- camera kernel: action -> (el, az) -> look_at R, T with forward-mode tangents.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

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

// camera of env n (one thread); pos2: optional second copy of C (the step's position snapshot)
__device__ __forceinline__ void camera_one(int mode, const float* __restrict__ action, float* __restrict__ el_io,
                                           float* __restrict__ az_io, const float* __restrict__ radius,
                                           float* __restrict__ cam, float* __restrict__ cam_pos_out,
                                           float* __restrict__ pos2, int n) {
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
    if (pos2) {
        pos2[3 * n] = C[0].v;
        pos2[3 * n + 1] = C[1].v;
        pos2[3 * n + 2] = C[2].v;
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
