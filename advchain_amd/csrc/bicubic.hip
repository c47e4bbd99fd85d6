// Bicubic sampling for 2D warps (gfx950): F.grid_sample(..., mode='bicubic', align_corners=True), which the reference
// reaches through the `forward_interp` / `backward_interp` config keys of AdvMorph and AdvAffine (adv_morph.py:255-258,
// 546-557; adv_affine.py:297-313; 2D only -- ATen has no 5-D bicubic).  An optional, non-default mode: plain direct
// gathers, one output pixel per thread, 16 taps -- semantics of ATen's GridSampler (cubic convolution with A = -0.75, the
// coordinate of EVERY tap padded on its own: zeros / border clip / reflection, GridSampler.h `get_value_bounded`).
//   advchain_grid_sample_bicubic2d_fwd / _bwd : dense planar grid (N, 2, H, W)
//   advchain_affine_grid2d_fwd / _bwd         : theta (N, 2, 3) -> planar grid and its adjoint (F.affine_grid), so that an
//                                               affine bicubic warp is the composition of the two
#include <stdlib.h>
#include "common.h"

namespace advchain {

constexpr float kCubicA = -0.75f;

__device__ __forceinline__ float cubic1(float x) { return ((kCubicA + 2.f) * x - (kCubicA + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { return ((kCubicA * x - 5.f * kCubicA) * x + 8.f * kCubicA) * x - 4.f * kCubicA; }

__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  c[0] = cubic2(t + 1.f);
  c[1] = cubic1(t);
  c[2] = cubic1(1.f - t);
  c[3] = cubic2(2.f - t);
}
__device__ __forceinline__ void cubic_coeffs_grad(float t, float (&c)[4]) {
  float x = -1.f - t;
  c[0] = (-3.f * kCubicA * x - 10.f * kCubicA) * x - 8.f * kCubicA;
  x = -t;
  c[1] = (-3.f * (kCubicA + 2.f) * x - 2.f * (kCubicA + 3.f)) * x;
  x = 1.f - t;
  c[2] = (3.f * (kCubicA + 2.f) * x - 2.f * (kCubicA + 3.f)) * x;
  x = 2.f - t;
  c[3] = (3.f * kCubicA * x - 10.f * kCubicA) * x + 8.f * kCubicA;
}

// padded integer coordinate of one tap (compute_coordinates on an integer-valued float), -1 = outside (zeros padding)
template <int PAD>
__device__ __forceinline__ int tap_index(int i, int S) {
  float x = (float)i;
  if (PAD == PAD_BORDER) {
    x = fminf(fmaxf(x, 0.f), (float)(S - 1));
  } else if (PAD == PAD_REFLECTION) {
    float g;
    x = reflect_coord(x, 0, 2 * (S - 1), g);
    x = fminf(fmaxf(x, 0.f), (float)(S - 1));
  }
  const int k = (int)x;
  return (k >= 0 && k < S) ? k : -1;
}

struct CubicTaps {
  int xi[4], yi[4];       // padded tap indices (-1: outside)
  float cx[4], cy[4];
  float tx, ty;
};

template <int PAD>
__device__ __forceinline__ CubicTaps cubic_taps(float gx, float gy, int W, int H) {
  CubicTaps t;
  float x = ((gx + 1.f) * 0.5f) * (float)(W - 1), y = ((gy + 1.f) * 0.5f) * (float)(H - 1);
  if (!(x > -1.0e9f && x < 1.0e9f)) x = -1.0e6f;      // NaN / huge: every tap outside (zeros) or at the clipped border
  if (!(y > -1.0e9f && y < 1.0e9f)) y = -1.0e6f;
  const float fx = floorf(x), fy = floorf(y);
  t.tx = x - fx;
  t.ty = y - fy;
  cubic_coeffs(t.tx, t.cx);
  cubic_coeffs(t.ty, t.cy);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t.xi[i] = tap_index<PAD>((int)fx - 1 + i, W);
    t.yi[i] = tap_index<PAD>((int)fy - 1 + i, H);
  }
  return t;
}

template <int PAD>
__global__ void __launch_bounds__(kBlock)
k_bicubic2d_fwd(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out, int C, int H, int W,
                int OH, int OW) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int OV = OH * OW, V = H * W;
  if (o >= OV) return;
  const float* gn = grid + (int64_t)n * 2 * OV;
  const CubicTaps t = cubic_taps<PAD>(gn[o], gn[OV + o], W, H);
  for (int c = 0; c < C; ++c) {
    const float* p = in + ((int64_t)n * C + c) * V;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float row = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = t.xi[i] >= 0 && t.yi[j] >= 0;
        const float v = p[max(t.yi[j], 0) * W + max(t.xi[i], 0)];     // unconditional load, value discarded by a select
        row += (ok ? v : 0.f) * t.cx[i];
      }
      acc += row * t.cy[j];
    }
    out[((int64_t)n * C + c) * OV + o] = acc;
  }
}

// grad_in must be zero-filled by the caller (atomics); grad_grid overwritten.  Either may be NULL.
template <int PAD>
__global__ void __launch_bounds__(kBlock)
k_bicubic2d_bwd(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                float* __restrict__ gin, float* __restrict__ ggrid, int C, int H, int W, int OH, int OW) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * kBlock + threadIdx.x;
  const int OV = OH * OW, V = H * W;
  if (o >= OV) return;
  const float* gn = grid + (int64_t)n * 2 * OV;
  const CubicTaps t = cubic_taps<PAD>(gn[o], gn[OV + o], W, H);
  float dcx[4], dcy[4];
  cubic_coeffs_grad(t.tx, dcx);
  cubic_coeffs_grad(t.ty, dcy);
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float go = gout[((int64_t)n * C + c) * OV + o];
    const float* p = in + ((int64_t)n * C + c) * V;
    float* q = gin ? gin + ((int64_t)n * C + c) * V : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = t.xi[i] >= 0 && t.yi[j] >= 0;
        const int off = max(t.yi[j], 0) * W + max(t.xi[i], 0);
        if (ggrid) {
          const float v = ok ? p[off] : 0.f;
          gix -= v * dcx[i] * t.cy[j] * go;
          giy -= v * dcy[j] * t.cx[i] * go;
        }
        if (q && ok) atomic_add_f32(q + off, go * t.cx[i] * t.cy[j]);
      }
  }
  if (ggrid) {
    float* gg = ggrid + (int64_t)n * 2 * OV;
    gg[o] = (0.5f * (float)(W - 1)) * gix;
    gg[OV + o] = (0.5f * (float)(H - 1)) * giy;
  }
}

// F.affine_grid(theta, (N, C, H, W), align_corners=True), planar: grid[n][r][y][x] = th[r][0] bx + th[r][1] by + th[r][2]
__global__ void __launch_bounds__(kBlock)
k_affine_grid2d_fwd(const float* __restrict__ theta, float* __restrict__ grid, int H, int W) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= H * W) return;
  const float* th = theta + (int64_t)n * 6;
  const float bx = affine_base_coord(o % W, W), by = affine_base_coord(o / W, H);
  float* g = grid + (int64_t)n * 2 * H * W;
  g[o] = th[0] * bx + th[1] * by + th[2];
  g[H * W + o] = th[3] * bx + th[4] * by + th[5];
}

// grad_theta[n][r][c] = sum_pixels ggrid[n][r] * base_c : block partial sums -> partial[(n * nb + b) * 6 + k]
__global__ void __launch_bounds__(kBlock)
k_affine_grid2d_bwd(const float* __restrict__ ggrid, float* __restrict__ partial, int H, int W) {
  __shared__ float smem[4 * 6];
  const int n = blockIdx.y;
  const int o = blockIdx.x * kBlock + threadIdx.x;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (o < H * W) {
    const float bx = affine_base_coord(o % W, W), by = affine_base_coord(o / W, H);
    const float* g = ggrid + (int64_t)n * 2 * H * W;
    const float a = g[o], b = g[H * W + o];
    acc[0] = a * bx; acc[1] = a * by; acc[2] = a;
    acc[3] = b * bx; acc[4] = b * by; acc[5] = b;
  }
  block_sum<6>(acc, smem);
  if (threadIdx.x == 0) {
    float* dst = partial + ((int64_t)n * gridDim.x + blockIdx.x) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k] = acc[k];
  }
}

__global__ void k_reduce_partials6(const float* __restrict__ partial, float* __restrict__ out, int nb) {
  const int n = blockIdx.x, k = blockIdx.y;
  __shared__ float smem[4];
  float s[1] = {0.f};
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s[0] += partial[((int64_t)n * nb + b) * 6 + k];
  block_sum<1>(s, smem);
  if (threadIdx.x == 0) out[(int64_t)n * 6 + k] = s[0];
}

}  // namespace advchain

using namespace advchain;

#define BICUBIC_PAD(PADV, ...)                                             \
  switch (PADV) {                                                          \
    case PAD_ZEROS: { constexpr int PAD = PAD_ZEROS; __VA_ARGS__; } break; \
    case PAD_BORDER: { constexpr int PAD = PAD_BORDER; __VA_ARGS__; } break; \
    default: { constexpr int PAD = PAD_REFLECTION; __VA_ARGS__; } break;   \
  }

static inline bool bdims_ok(const int64_t* s) { return s && s[0] >= 1 && s[1] >= 1 && s[0] * s[1] < (1ll << 30); }

extern "C" {

int advchain_grid_sample_bicubic2d_fwd(const float* in, const float* grid, float* out, int64_t N, int64_t C,
                                       const int64_t* in_dims, const int64_t* out_dims, int padding, void* stream) {
  ADVCHAIN_CHECK_ARG(in && grid && out, "grid_sample_bicubic2d_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(bdims_ok(in_dims) && bdims_ok(out_dims), "grid_sample_bicubic2d_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1 && padding >= 0 && padding <= 2, "grid_sample_bicubic2d_fwd: bad N/C/padding");
  if (N == 0) return ADVCHAIN_OK;
  const int H = (int)in_dims[0], W = (int)in_dims[1], OH = (int)out_dims[0], OW = (int)out_dims[1];
  dim3 g(advchain_blocks((int64_t)OH * OW, kBlock), (unsigned)N), b(kBlock);
  BICUBIC_PAD(padding, { hipLaunchKernelGGL(k_bicubic2d_fwd<PAD>, g, b, 0, (hipStream_t)stream, in, grid, out, (int)C, H, W, OH, OW); });
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_grid_sample_bicubic2d_bwd(const float* grad_out, const float* in, const float* grid, float* grad_in,
                                       float* grad_grid, int64_t N, int64_t C, const int64_t* in_dims,
                                       const int64_t* out_dims, int padding, void* stream) {
  ADVCHAIN_CHECK_ARG(grad_out && in && grid && (grad_in || grad_grid), "grid_sample_bicubic2d_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(bdims_ok(in_dims) && bdims_ok(out_dims), "grid_sample_bicubic2d_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1 && padding >= 0 && padding <= 2, "grid_sample_bicubic2d_bwd: bad N/C/padding");
  if (N == 0) return ADVCHAIN_OK;
  const int H = (int)in_dims[0], W = (int)in_dims[1], OH = (int)out_dims[0], OW = (int)out_dims[1];
  hipStream_t st = (hipStream_t)stream;
  if (grad_in) advchain_zero_async(grad_in, sizeof(float) * N * C * H * W, st);
  dim3 g(advchain_blocks((int64_t)OH * OW, kBlock), (unsigned)N), b(kBlock);
  BICUBIC_PAD(padding, { hipLaunchKernelGGL(k_bicubic2d_bwd<PAD>, g, b, 0, st, grad_out, in, grid, grad_in, grad_grid, (int)C, H, W, OH, OW); });
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_affine_grid2d_fwd(const float* theta, float* grid, int64_t N, const int64_t* dims, void* stream) {
  ADVCHAIN_CHECK_ARG(theta && grid && bdims_ok(dims), "affine_grid2d_fwd: null pointer / bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "affine_grid2d_fwd: bad N");
  if (N == 0) return ADVCHAIN_OK;
  const int H = (int)dims[0], W = (int)dims[1];
  hipLaunchKernelGGL(k_affine_grid2d_fwd, dim3(advchain_blocks((int64_t)H * W, kBlock), (unsigned)N), dim3(kBlock), 0,
                     (hipStream_t)stream, theta, grid, H, W);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int64_t advchain_affine_grid2d_bwd_workspace(int64_t N, const int64_t* dims) {
  if (!bdims_ok(dims)) return -1;
  return N * (int64_t)advchain_blocks(dims[0] * dims[1], kBlock) * 6;
}

int advchain_affine_grid2d_bwd(const float* grad_grid, float* grad_theta, float* workspace, int64_t N, const int64_t* dims,
                               void* stream) {
  ADVCHAIN_CHECK_ARG(grad_grid && grad_theta && workspace && bdims_ok(dims), "affine_grid2d_bwd: null pointer / bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "affine_grid2d_bwd: bad N");
  if (N == 0) return ADVCHAIN_OK;
  const int H = (int)dims[0], W = (int)dims[1];
  const int nb = advchain_blocks((int64_t)H * W, kBlock);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_affine_grid2d_bwd, dim3(nb, (unsigned)N), dim3(kBlock), 0, st, grad_grid, workspace, H, W);
  hipLaunchKernelGGL(k_reduce_partials6, dim3((unsigned)N, 6), dim3(kBlock), 0, st, workspace, grad_theta, nb);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

}  // extern "C"
