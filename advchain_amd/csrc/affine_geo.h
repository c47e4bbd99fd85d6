// Per-sample geometry of an affine warp for the grad_in kernels (k_affine_gather_bwd, k_affine_box_gin): the voxel-space matrix
// M and offset t of theta, the inverse Mi, the extent of a unit voxel's pre-image, and mode = 1 for a sample the lattice gather
// cannot take (degenerate / strongly minifying theta: it goes through the atomic kernel).  One function for the stand-alone
// launch (k_affine_geometry) and for the theta-gradient kernel, which computes it on the side when both gradients are asked
// for (round 5: one launch less per affine backward).
#ifndef ADVCHAIN_AFFINE_GEO_H_
#define ADVCHAIN_AFFINE_GEO_H_
#include "common.h"

namespace advchain {

constexpr int kGeoFloats = 24;
constexpr float kGatherMaxExt = 6.f;

template <int DIM>
__device__ __forceinline__ void affine_geometry_one(const float* __restrict__ theta, float* __restrict__ geo, int* __restrict__ mode,
                                                    int n, const Dims& d) {
  const int S[3] = {d.s2, d.s1, d.s0};  // x, y, z
  float M[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
  bool ok = true;
  for (int r = 0; r < DIM; ++r) {
    float sum = 0.f;
    for (int a = 0; a < DIM; ++a) {
      const float th = theta[(n * DIM + r) * (DIM + 1) + a];
      if (S[a] > 1) { M[r][a] = th * (float)(S[r] - 1) / (float)(S[a] - 1); sum += th; }
      else { M[r][a] = 0.f; ok = false; }
    }
    t[r] = ((theta[(n * DIM + r) * (DIM + 1) + DIM] - sum) + 1.f) * 0.5f * (float)(S[r] - 1);
  }
  const float a = M[0][0], b = M[0][1], c = M[0][2], e = M[1][0], f = M[1][1], g = M[1][2], h = M[2][0], i = M[2][1], j = M[2][2];
  const float A = f * j - g * i, B = -(e * j - g * h), Cc = e * i - f * h;
  const float det = a * A + b * B + c * Cc;
  float Mi[3][3];
  const float rdet = 1.f / det;
  Mi[0][0] = A * rdet; Mi[0][1] = -(b * j - c * i) * rdet; Mi[0][2] = (b * g - c * f) * rdet;
  Mi[1][0] = B * rdet; Mi[1][1] = (a * j - c * h) * rdet;  Mi[1][2] = -(a * g - c * e) * rdet;
  Mi[2][0] = Cc * rdet; Mi[2][1] = -(a * i - b * h) * rdet; Mi[2][2] = (a * f - b * e) * rdet;
  float* gn = geo + (int64_t)n * kGeoFloats;
  float ext[3];
  for (int q = 0; q < 3; ++q) {
    ext[q] = fabsf(Mi[q][0]) + fabsf(Mi[q][1]) + fabsf(Mi[q][2]) + 0.01f;
    if (q < DIM && !(ext[q] < kGatherMaxExt)) ok = false;   // also catches NaN / inf
  }
  if (!(fabsf(det) > 1e-6f)) ok = false;
  for (int r = 0; r < 3; ++r)
    for (int q = 0; q < 3; ++q) { gn[r * 3 + q] = M[r][q]; gn[12 + r * 3 + q] = Mi[r][q]; }
  for (int r = 0; r < 3; ++r) { gn[9 + r] = t[r]; gn[21 + r] = ext[r]; }
  mode[n] = ok ? 0 : 1;
}

}  // namespace advchain
#endif
