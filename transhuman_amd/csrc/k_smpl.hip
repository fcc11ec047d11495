// K10: SMPL linear blend skinning on device.
//
// SMPL._call, lib/utils/SMPL.py:114-186: shape blend (v_shaped = shapedirs beta + template, :122), joint
// regression (:127), axis-angle -> rotation (cv2.Rodrigues, :135-139), pose blend (posedirs (R - I), :147-149),
// the 24-joint kinematic chain (:151-170) and the per-vertex blend matrices T = weights G with the skinned
// vertices v = T [v_posed, 1] (:175-186).  The reference runs it in numpy float64 inside the DataLoader for
// every frame (can_smpl.py:262,:300); `T` is the path's `blend_mtx` input and `v` its `*_smplcoord` vertices.
// Everything here is float64 like the reference (model arrays are float64, R is float32 promoted at first use).
// 4 launches, all latency-bound (6890 vertices): bound = HBM read of posedirs (34 MB, L2/MALL resident after
// the first frame).
#include "th_internal.h"

#define SM_J 24

// rotation of one joint: OpenCV's Rodrigues formula (documented: theta = |r|; R = c I + (1 - c) n n^T + s [n]x),
// float64, rounded to float32 like `np.array([...], dtype='float32')` (:137-139)
__device__ __forceinline__ void sm_rodrigues(const float* __restrict__ r, float* __restrict__ Rout) {
    double x = r[0], y = r[1], z = r[2];
    double t = sqrt(x * x + y * y + z * z);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (t >= 2.220446049250313e-16) {
        double nx = x / t, ny = y / t, nz = z / t, c = cos(t), s = sin(t), c1 = 1.0 - c;
        R[0] = c + c1 * nx * nx;      R[1] = c1 * nx * ny - s * nz; R[2] = c1 * nx * nz + s * ny;
        R[3] = c1 * ny * nx + s * nz; R[4] = c + c1 * ny * ny;      R[5] = c1 * ny * nz - s * nx;
        R[6] = c1 * nz * nx - s * ny; R[7] = c1 * nz * ny + s * nx; R[8] = c + c1 * nz * nz;
    }
    for (int i = 0; i < 9; ++i) Rout[i] = (float)R[i];
}

// (1) v_shaped [nv,3]; one thread per vertex
__global__ void smpl_shape_kernel(const double* __restrict__ tmpl, const double* __restrict__ shapedirs,
                                  const double* __restrict__ beta, int nv, double* __restrict__ v_shaped) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a = 0.0;
        for (int b = 0; b < 10; ++b) a += shapedirs[((long long)v * 3 + c) * 10 + b] * beta[b];
        v_shaped[3 * v + c] = a + tmpl[3 * v + c];
    }
}

// (2) J [24,3] = J_regressor v_shaped; one 256-thread block per joint (tree reduction: order differs from BLAS,
// float64 keeps the difference at 1e-15 relative)
__global__ __launch_bounds__(256) void smpl_joint_kernel(const double* __restrict__ jreg,
                                                         const double* __restrict__ v_shaped, int nv,
                                                         double* __restrict__ J) {
    __shared__ double red[3][256];
    const int j = blockIdx.x;
    double a[3] = {0, 0, 0};
    for (int v = threadIdx.x; v < nv; v += 256) {
        double w = jreg[(long long)j * nv + v];
        a[0] += w * v_shaped[3 * v]; a[1] += w * v_shaped[3 * v + 1]; a[2] += w * v_shaped[3 * v + 2];
    }
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = a[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) J[3 * j + threadIdx.x] = red[threadIdx.x][0];
}

// (3) rotations, kinematic chain, G - G [0 | J;0]  -> Gout [24,16], joints [24,3], lrot [207]; one block
__global__ __launch_bounds__(64) void smpl_chain_kernel(const float* __restrict__ pose_aa, const float* __restrict__ Rin,
                                                        const int32_t* __restrict__ parent, const double* __restrict__ J,
                                                        float* __restrict__ Rf, double* __restrict__ Gout,
                                                        double* __restrict__ joints, double* __restrict__ lrot) {
    __shared__ float R[SM_J][9];
    __shared__ double G[SM_J][16];
    const int t = threadIdx.x;
    if (t < SM_J) {
        if (pose_aa) sm_rodrigues(pose_aa + 3 * t, R[t]);
        else for (int i = 0; i < 9; ++i) R[t][i] = Rin[9 * t + i];
        for (int i = 0; i < 9; ++i) Rf[9 * t + i] = R[t][i];
    }
    __syncthreads();
    if (t == 0) {
        for (int j = 0; j < SM_J; ++j) {
            // local transform [R | J_rel; 0 0 0 1] (:152-157)
            double L[16];
            const int p = parent[j];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) L[4 * r + c] = (double)R[j][3 * r + c];
                L[4 * r + 3] = (j == 0) ? J[3 * j + r] : J[3 * j + r] - J[3 * p + r];
            }
            L[12] = 0; L[13] = 0; L[14] = 0; L[15] = 1;
            if (j == 0) for (int i = 0; i < 16; ++i) G[0][i] = L[i];
            else
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 4; ++c) {
                        double a = 0.0;
                        for (int k = 0; k < 4; ++k) a += G[p][4 * r + k] * L[4 * k + c];     // G[parent] . G_[j] (:161)
                        G[j][4 * r + c] = a;
                    }
        }
    }
    __syncthreads();
    if (t < SM_J) {
        for (int c = 0; c < 3; ++c) joints[3 * t + c] = G[t][4 * c + 3];
        // G - G . [0 | (J, 0)] : only the last column changes (:166-170)
        for (int r = 0; r < 4; ++r) {
            double a = 0.0;
            for (int k = 0; k < 3; ++k) a += G[t][4 * r + k] * J[3 * t + k];
            for (int c = 0; c < 3; ++c) Gout[16 * t + 4 * r + c] = G[t][4 * r + c];
            Gout[16 * t + 4 * r + 3] = G[t][4 * r + 3] - a;
        }
        if (t >= 1)
            for (int i = 0; i < 9; ++i) lrot[9 * (t - 1) + i] = (double)(R[t][i] - ((i % 4 == 0) ? 1.0f : 0.0f));   // float32 subtract (:147)
    }
}

// (4) pose blend + skinning; one thread per vertex
__global__ __launch_bounds__(128) void smpl_skin_kernel(const double* __restrict__ v_shaped,
                                                        const double* __restrict__ posedirs,
                                                        const double* __restrict__ weights,
                                                        const double* __restrict__ G, const double* __restrict__ lrot,
                                                        int nv, double* __restrict__ verts, double* __restrict__ T) {
    __shared__ double Gs[SM_J * 16];
    __shared__ double ls[207];
    for (int i = threadIdx.x; i < SM_J * 16; i += blockDim.x) Gs[i] = G[i];
    for (int i = threadIdx.x; i < 207; i += blockDim.x) ls[i] = lrot[i];
    __syncthreads();
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    double vp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double* pd = posedirs + ((long long)v * 3 + c) * 207;
        double a = 0.0;
        for (int k = 0; k < 207; ++k) a += pd[k] * ls[k];
        vp[c] = v_shaped[3 * v + c] + a;
    }
    double Tm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Tm[i] = 0.0;
    for (int j = 0; j < SM_J; ++j) {
        double w = weights[(long long)v * SM_J + j];
#pragma unroll
        for (int i = 0; i < 16; ++i) Tm[i] += w * Gs[16 * j + i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) T[(long long)v * 16 + i] = Tm[i];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        verts[3 * v + r] = ((Tm[4 * r] * vp[0] + Tm[4 * r + 1] * vp[1]) + Tm[4 * r + 2] * vp[2]) + Tm[4 * r + 3] * 1.0;
}

size_t th_smpl_ws(int nv) {
    return th_align((size_t)nv * 3 * 8) + th_align(SM_J * 3 * 8) + th_align(SM_J * 16 * 8) + th_align(207 * 8) +
           th_align(SM_J * 9 * 4);
}

int th_smpl_launch(const th_smpl_model& m, const float* pose_aa, const float* R, const double* beta, double* verts,
                   double* joints, double* T, void* ws, size_t ws_bytes, hipStream_t s) {
    TH_REQUIRE(ws_bytes >= th_smpl_ws(m.n_verts), "workspace too small");
    ThArena ar(ws, ws_bytes);
    const int nv = m.n_verts;
    double* v_shaped = ar.take<double>((size_t)nv * 3);
    double* J = ar.take<double>(SM_J * 3);
    double* G = ar.take<double>(SM_J * 16);
    double* lrot = ar.take<double>(207);
    float* Rf = ar.take<float>(SM_J * 9);
    TH_REQUIRE(Rf != nullptr, "workspace carve failed");
    hipLaunchKernelGGL(smpl_shape_kernel, dim3(th_cdiv(nv, 256)), dim3(256), 0, s, m.v_template, m.shapedirs, beta, nv,
                       v_shaped);
    hipLaunchKernelGGL(smpl_joint_kernel, dim3(SM_J), dim3(256), 0, s, m.J_regressor, v_shaped, nv, J);
    hipLaunchKernelGGL(smpl_chain_kernel, dim3(1), dim3(64), 0, s, pose_aa, R, m.parent, J, Rf, G, joints, lrot);
    hipLaunchKernelGGL(smpl_skin_kernel, dim3(th_cdiv(nv, 128)), dim3(128), 0, s, v_shaped, m.posedirs, m.weights, G, lrot,
                       nv, verts, T);
    TH_LAUNCH_CHECK();
    return 0;
}
