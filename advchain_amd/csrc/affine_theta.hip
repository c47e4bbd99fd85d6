// AdvAffine parameter -> matrix map (and its inverse) for gfx950.
//
//   advchain_affine_theta_fwd <- gen_batch_affine_matrix adv_affine.py:210-273 + get_inverse_matrix :316-324
//   advchain_affine_theta_bwd <- autograd through the above (Hardtanh, sin/cos, 4x4 matmuls, linalg_inv)
//
// O(N) work (N = batch): one tiny launch replaces ~40 ATen launches.  The backward uses forward-mode
// dual numbers over the same templated evaluation, so forward and gradient cannot drift apart.
#include "common.h"

namespace advchain {

struct AffineCfg { float c[9]; };

struct Dual {
  float v, d;
  __device__ Dual() : v(0.f), d(0.f) {}
  __device__ Dual(float a) : v(a), d(0.f) {}
  __device__ Dual(float a, float b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual dsin(Dual a) { return Dual(sinf(a.v), cosf(a.v) * a.d); }
__device__ __forceinline__ Dual dcos(Dual a) { return Dual(cosf(a.v), -sinf(a.v) * a.d); }
__device__ __forceinline__ float dsin(float a) { return sinf(a); }
__device__ __forceinline__ float dcos(float a) { return cosf(a); }

constexpr float kPi = 3.14159265358979323846f;

// theta rows [r][c], r < DIM, c <= DIM, from already-bounded parameters p
template <int DIM, class T>
__device__ __forceinline__ void theta_eval(const T* p, const AffineCfg& k, T (&th)[3][4]) {
  if constexpr (DIM == 2) {
    // cfg = {rot, scale_x, scale_y, shift_x, shift_y}; p = {rot, sx, sy, tx, ty}   (adv_affine.py:220-226)
    const T a = (p[0] * T(k.c[0])) * T(kPi);
    const T sx = T(1.f) + p[1] * T(k.c[1]);
    const T sy = T(1.f) + p[2] * T(k.c[2]);
    th[0][0] = sx * dcos(a);
    th[0][1] = sy * (-dsin(a));
    th[0][2] = p[3] * T(k.c[3]);
    th[1][0] = sx * dsin(a);
    th[1][1] = sy * dcos(a);
    th[1][2] = p[4] * T(k.c[4]);
  } else {
    // cfg = {rot_x,rot_y,rot_z, scale_x,scale_y,scale_z, shift_x,shift_y,shift_z}  (adv_affine.py:229-269)
    const T ph = (p[0] * T(k.c[0])) * T(kPi);
    const T t = (p[1] * T(k.c[1])) * T(kPi);
    const T ps = (p[2] * T(k.c[2])) * T(kPi);
    const T s0 = T(1.f) + p[3] * T(k.c[3]);
    const T s1 = T(1.f) + p[4] * T(k.c[4]);
    const T s2 = T(1.f) + p[5] * T(k.c[5]);
    const T cph = dcos(ph), sph = dsin(ph), ct = dcos(t), st = dsin(t), cps = dcos(ps), sps = dsin(ps);
    // R = Euler z-y'-x'' ; theta = (T . R . S)[:3,:4] = [R diag(s) | t]
    th[0][0] = (ct * cps) * s0;
    th[0][1] = (-(cph * sps) + (sph * st) * cps) * s1;
    th[0][2] = (sph * sps + (cph * st) * cps) * s2;
    th[0][3] = p[6] * T(k.c[6]);
    th[1][0] = (ct * sps) * s0;
    th[1][1] = (cph * cps + (sph * st) * sps) * s1;
    th[1][2] = (-(sph * cps) + (cph * st) * sps) * s2;
    th[1][3] = p[7] * T(k.c[7]);
    th[2][0] = (-st) * s0;
    th[2][1] = (sph * ct) * s1;
    th[2][2] = (cph * ct) * s2;
    th[2][3] = p[8] * T(k.c[8]);
  }
}

// inverse of the homogeneous matrix [A t; 0 1]: [A^-1 | -A^-1 t]
template <int DIM>
__device__ __forceinline__ void affine_invert(const float (&th)[3][4], float (&inv)[3][4]) {
  if constexpr (DIM == 2) {
    const float a = th[0][0], b = th[0][1], c = th[1][0], d = th[1][1];
    const float det = a * d - b * c;
    const float r = 1.f / det;
    inv[0][0] = d * r; inv[0][1] = -b * r; inv[1][0] = -c * r; inv[1][1] = a * r;
    inv[0][2] = -(inv[0][0] * th[0][2] + inv[0][1] * th[1][2]);
    inv[1][2] = -(inv[1][0] * th[0][2] + inv[1][1] * th[1][2]);
  } else {
    const float a = th[0][0], b = th[0][1], c = th[0][2];
    const float d = th[1][0], e = th[1][1], f = th[1][2];
    const float g = th[2][0], h = th[2][1], i = th[2][2];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    const float r = 1.f / det;
    inv[0][0] = A * r; inv[0][1] = -(b * i - c * h) * r; inv[0][2] = (b * f - c * e) * r;
    inv[1][0] = B * r; inv[1][1] = (a * i - c * g) * r;  inv[1][2] = -(a * f - c * d) * r;
    inv[2][0] = C * r; inv[2][1] = -(a * h - b * g) * r; inv[2][2] = (a * e - b * d) * r;
    for (int q = 0; q < 3; ++q)
      inv[q][3] = -(inv[q][0] * th[0][3] + inv[q][1] * th[1][3] + inv[q][2] * th[2][3]);
  }
}

template <int DIM>
__global__ void k_affine_theta_fwd(const float* __restrict__ param, AffineCfg k, float pscale,
                                   float* __restrict__ theta, float* __restrict__ theta_inv, int N) {
  constexpr int NP = DIM == 2 ? 5 : 9;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float p[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) p[j] = fminf(fmaxf(pscale * param[n * NP + j], -1.f), 1.f);  // Hardtanh
  float th[3][4], inv[3][4];
  theta_eval<DIM, float>(p, k, th);
  affine_invert<DIM>(th, inv);
  for (int r = 0; r < DIM; ++r)
    for (int c = 0; c <= DIM; ++c) {
      theta[(n * DIM + r) * (DIM + 1) + c] = th[r][c];
      if (theta_inv) theta_inv[(n * DIM + r) * (DIM + 1) + c] = inv[r][c];
    }
}

// one thread per (sample, parameter)
template <int DIM>
__global__ void k_affine_theta_bwd(const float* __restrict__ param, AffineCfg k, float pscale,
                                   const float* __restrict__ gtheta, const float* __restrict__ gtheta_inv,
                                   float* __restrict__ gparam, int N) {
  constexpr int NP = DIM == 2 ? 5 : 9;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * NP) return;
  const int n = t / NP, j = t % NP;
  float raw[NP];
  Dual p[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    raw[q] = pscale * param[n * NP + q];
    p[q] = Dual(fminf(fmaxf(raw[q], -1.f), 1.f), 0.f);
  }
  // hardtanh_backward: gradient strictly inside (-1, 1)
  const bool interior = raw[j] > -1.f && raw[j] < 1.f;
  if (!interior) { gparam[t] = 0.f; return; }
  p[j].d = 1.f;
  Dual thd[3][4];
  theta_eval<DIM, Dual>(p, k, thd);
  // G = gtheta - (Minv^T [ginv;0] Minv^T)[:DIM]   with Minv the homogeneous inverse
  float G[3][4];
  for (int r = 0; r < DIM; ++r)
    for (int c = 0; c <= DIM; ++c) G[r][c] = gtheta ? gtheta[(n * DIM + r) * (DIM + 1) + c] : 0.f;
  if (gtheta_inv) {
    float th[3][4], inv[3][4];
    for (int r = 0; r < DIM; ++r)
      for (int c = 0; c <= DIM; ++c) th[r][c] = thd[r][c].v;
    affine_invert<DIM>(th, inv);
    // homogeneous Minv (DIM+1 x DIM+1): rows < DIM from inv, last row e_{DIM}
    float Mi[4][4], Gi[4][4], tmp[4][4];
    for (int r = 0; r <= DIM; ++r)
      for (int c = 0; c <= DIM; ++c) {
        Mi[r][c] = r < DIM ? inv[r][c] : (c == DIM ? 1.f : 0.f);
        Gi[r][c] = r < DIM ? gtheta_inv[(n * DIM + r) * (DIM + 1) + c] : 0.f;
      }
    // tmp = Mi^T Gi ; out = tmp Mi^T
    for (int r = 0; r <= DIM; ++r)
      for (int c = 0; c <= DIM; ++c) {
        float s = 0.f;
        for (int q = 0; q <= DIM; ++q) s += Mi[q][r] * Gi[q][c];
        tmp[r][c] = s;
      }
    for (int r = 0; r < DIM; ++r)
      for (int c = 0; c <= DIM; ++c) {
        float s = 0.f;
        for (int q = 0; q <= DIM; ++q) s += tmp[r][q] * Mi[c][q];
        G[r][c] -= s;
      }
  }
  float acc = 0.f;
  for (int r = 0; r < DIM; ++r)
    for (int c = 0; c <= DIM; ++c) acc += G[r][c] * thd[r][c].d;
  gparam[t] = acc * pscale;
}

}  // namespace advchain

using namespace advchain;

extern "C" {

int advchain_affine_theta_fwd(const float* param, const float* cfg_host, float param_scale, float* theta,
                              float* theta_inv, int64_t N, int ndim, void* stream) {
  ADVCHAIN_CHECK_ARG(param && cfg_host && theta, "affine_theta_fwd: null pointer");
  ADVCHAIN_CHECK_ARG((ndim == 2 || ndim == 3) && N >= 0 && N < (1 << 24), "affine_theta_fwd: bad ndim/N");
  if (N == 0) return ADVCHAIN_OK;
  AffineCfg k;
  const int nc = ndim == 2 ? 5 : 9;
  for (int i = 0; i < 9; ++i) k.c[i] = i < nc ? cfg_host[i] : 0.f;
  dim3 g(advchain_blocks(N, 64)), b(64);
  if (ndim == 2) hipLaunchKernelGGL(k_affine_theta_fwd<2>, g, b, 0, (hipStream_t)stream, param, k, param_scale, theta, theta_inv, (int)N);
  else hipLaunchKernelGGL(k_affine_theta_fwd<3>, g, b, 0, (hipStream_t)stream, param, k, param_scale, theta, theta_inv, (int)N);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_affine_theta_bwd(const float* param, const float* cfg_host, float param_scale, const float* grad_theta,
                              const float* grad_theta_inv, float* grad_param, int64_t N, int ndim, void* stream) {
  ADVCHAIN_CHECK_ARG(param && cfg_host && grad_param && (grad_theta || grad_theta_inv), "affine_theta_bwd: null pointer");
  ADVCHAIN_CHECK_ARG((ndim == 2 || ndim == 3) && N >= 0 && N < (1 << 24), "affine_theta_bwd: bad ndim/N");
  if (N == 0) return ADVCHAIN_OK;
  AffineCfg k;
  const int nc = ndim == 2 ? 5 : 9;
  for (int i = 0; i < 9; ++i) k.c[i] = i < nc ? cfg_host[i] : 0.f;
  dim3 g(advchain_blocks(N * nc, 64)), b(64);
  if (ndim == 2) hipLaunchKernelGGL(k_affine_theta_bwd<2>, g, b, 0, (hipStream_t)stream, param, k, param_scale, grad_theta, grad_theta_inv, grad_param, (int)N);
  else hipLaunchKernelGGL(k_affine_theta_bwd<3>, g, b, 0, (hipStream_t)stream, param, k, param_scale, grad_theta, grad_theta_inv, grad_param, (int)N);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

}  // extern "C"
