// Fused backward of the early squarings of the 2D scaling-and-squaring chain (gfx950): grad(phi_k) -> grad(phi_0) in ONE
// launch, for squarings whose inputs move every sample by less than a pixel (exact bound, measured in the forward).
//
//   replaces k launches of advchain_compose_self_bwd (gather form, adjoint_gather.hip: k_adjoint_gather<2, 2, 1, SELF>) at
//   the END of the chain's backward -- the autograd of `for i in range(nb_steps): phi = applyComposition2D(phi, phi)`
//   (adv_morph.py:116-146, 179-190) for its first k iterations.
//
// With |p_s - s| < 1 the scatter-add of grid_sample's backward is a gather over the 3 x 3 neighbouring samples
// (adjoint_gather.hip).  One level turns grad(phi_L) on a tile + 1 halo into grad(phi_{L-1}) on the tile, so a workgroup that
// stages grad(phi_k) on its tile + k halo walks k levels without leaving LDS: the k - 1 intermediate gradients (33 MB each
// at cfg-2, written and read back by the per-squaring launches) never exist in memory, and the k stage -> barrier -> gather
// -> store round trips of k launches become one.  Per level only the field phi_{L-1} comes from memory (requested one level
// ahead, into registers).
//
//   * geometry of k_adjoint_gather<2, 2, 1, ...>: lane <-> x (64 lanes, 56 owned + 4 either side when the row is wider
//     than a wave), TY owned rows + k halo rows either side; LDS [4][ROWS][64]: offsets o = unnormalize(phi) - s (2 channels)
//     and the running gradient (2 channels);
//   * a level is that kernel's phase A (coordinate path of the own sample, corners from LDS) and phase B (deposits of the
//     3 sample rows around an output row, 3 partial sums per lane folded by two whole-wave DPP shifts) with the same
//     arithmetic in the same order: the result is BIT-IDENTICAL to the k separate launches (tests/test_fused2d_gpu.py);
//   * the region shrinks by one row / one lane per level; what lies outside it is computed anyway (garbage in, garbage out,
//     never read by a valid output) with every LDS index clamped;
//   * exact bounds only: the caller guarantees the displacement of phi_0..phi_{k-1} (read back from the forward's
//     measurement) -- no overflow list, no device-side check needed here.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

constexpr int kFXPad = 4;          // x halo of a 64-lane tile (>= the number of fused levels, multiple of 4)
constexpr int kFuseBwdMax = 4;

__device__ __forceinline__ float unnormalize_f(float g, int S) { return ((g + 1.f) * 0.5f) * (float)(S - 1); }

template <int E>
__device__ __forceinline__ float tent_f(float f) { return fmaxf(0.f, 1.f - fabsf(f - (float)E)); }

template <int TY, int NT>
__global__ void __launch_bounds__(NT)
k_adjoint_fused2d(const float* __restrict__ gk, const float* __restrict__ phi0, const float* __restrict__ fields, int64_t F,
                  float* __restrict__ g0, Dims d, int k, int n1, int n2, int wide, int32_t* __restrict__ untracked) {
  constexpr int NW = NT / 64;
  constexpr int RMAX = TY + 2 * kFuseBwdMax;             // window rows at most
  constexpr int RPW = (TY + 2 * (kFuseBwdMax - 1) + NW - 1) / NW;   // output rows of a level per wave, at most
  constexpr int QPT = (RMAX * 16 + NT - 1) / NT;         // staging quads per thread, at most
  if (untracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) untracked[3] = -1;   // (see k_adjoint_gather)
  extern __shared__ float lds[];                         // [4][ROWS][64]: o_x, o_y, g_x, g_y
  const int ROWS = TY + 2 * k;
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b = blockIdx.x;
  const int tx = b % n2;
  const int ty = b / n2;
  const int xown = wide ? 64 - 2 * kFXPad : 64;
  const int x0 = tx * xown, y0 = ty * TY;
  const int rx0 = wide ? x0 - kFXPad : 0;                // x of lane 0 (multiple of 4)
  const int ry0 = y0 - k;                                // image row of window row 0
  const int S[2] = {d.s2, d.s1};

  // one item = 4 consecutive x of one window row: the offsets of a field (rows / quads outside the image: zeros)
  auto load_field = [&](const float* __restrict__ fn, int e, float (&o)[2][4]) {
    const int q = e & 15, r = e >> 4;
    const int sy = ry0 + r, x = rx0 + 4 * q;
    const bool inside = e < ROWS * 16 && sy >= 0 && sy < d.s1 && x >= 0 && x < d.s2;
    const int s = inside ? sy * d.s2 + x : 0;            // (unconditional loads from a clamped address)
#pragma unroll
    for (int a = 0; a < 2; ++a) load_vec<4>(fn + (int64_t)a * V + s, o[a]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float xs0 = unnormalize_f(o[0][kk], S[0]), xs1 = unnormalize_f(o[1][kk], S[1]);
      o[0][kk] = (inside && xs0 > -1.0e9f && xs0 < 1.0e9f) ? xs0 - (float)(x + kk) : 0.f;
      o[1][kk] = (inside && xs1 > -1.0e9f && xs1 < 1.0e9f) ? xs1 - (float)sy : 0.f;
    }
  };
  auto field_of = [&](int m) { return (m == 0 ? phi0 : fields + (int64_t)(m - 1) * F) + (int64_t)n * 2 * V; };

  // ---- stage level k: offsets of phi_{k-1}, gradient w.r.t. phi_k
  {
    const float* fn = field_of(k - 1);
    const float* gn = gk + (int64_t)n * 2 * V;
    for (int e = threadIdx.x; e < ROWS * 16; e += NT) {
      const int q = e & 15, r = e >> 4;
      const int sy = ry0 + r, x = rx0 + 4 * q;
      const bool inside = sy >= 0 && sy < d.s1 && x >= 0 && x < d.s2;
      float o[2][4], g[2][4];
      load_field(fn, e, o);
      const int s = inside ? sy * d.s2 + x : 0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        load_vec<4>(gn + (int64_t)c * V + s, g[c]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) g[c][kk] = inside ? g[c][kk] : 0.f;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) store_vec<4>(lds + (a * ROWS + r) * 64 + 4 * q, o[a]);
#pragma unroll
      for (int c = 0; c < 2; ++c) store_vec<4>(lds + ((2 + c) * ROWS + r) * 64 + 4 * q, g[c]);
    }
  }
  __syncthreads();

  const int sx = rx0 + lane;
  const bool in_x = sx >= 0 && sx < d.s2;
  const bool xowned = wide ? (lane >= kFXPad && lane < 64 - kFXPad && sx < d.s2) : (sx < d.s2);
  const float xlo = -(float)sx, xhi = (float)(d.s2 - 1 - sx);   // clip bounds of a sample in this lane, as offsets
  float* g0n = g0 + (int64_t)n * 2 * V;

  for (int L = k; L >= 1; --L) {
    // ---- request the offsets of phi_{L-2} for the next level (in flight under this level's arithmetic)
    float nxt[QPT][2][4];
    if (L >= 2) {
      const float* fn = field_of(L - 2);
#pragma unroll
      for (int i = 0; i < QPT; ++i) load_field(fn, threadIdx.x + i * NT, nxt[i]);
    }
    const int rfirst = k - L + 1, count = TY + 2 * (L - 1);     // window rows this level produces
    float res[RPW][2];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      res[j][0] = res[j][1] = 0.f;
      const int lr = wave + j * NW;                       // wave-uniform
      const int rc = rfirst + lr;                         // window row of the output
      const int uy = ry0 + rc;
      if (lr >= count || uy < 0 || uy >= d.s1) continue;
      // ---- phase A: coordinate-path gradient of the sample at this output position (k_adjoint_gather, SELF, 2D)
      float gg[2] = {0.f, 0.f};
      {
        const int sc[2] = {sx, uy};
        float w1[2], mult[2];
        int i0[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float xs = lds[(a * ROWS + rc) * 64 + lane] + (float)sc[a];
          const float top = (float)(S[a] - 1);
          mult[a] = 0.5f * top;
          if (xs <= 0.f) mult[a] = 0.f;
          if (xs >= top) mult[a] = 0.f;
          xs = fminf(fmaxf(xs, 0.f), top);
          const float fl = floorf(xs);
          i0[a] = (int)fl;
          w1[a] = xs - fl;
        }
        float go[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) go[c] = lds[((2 + c) * ROWS + rc) * 64 + lane];
        const bool okx0 = i0[0] >= 0, okx1 = i0[0] + 1 < d.s2;
        // (indices clamped into the tile: lanes / rows outside the level's region may hold anything)
        const int lx0 = min(max(max(i0[0], 0) - rx0, 0), 63), lx1 = min(max(min(i0[0] + 1, d.s2 - 1) - rx0, 0), 63);
        const int r00 = min(max(i0[1] - ry0, 0), ROWS - 2);
        const float wx1 = w1[0], wx0 = 1.f - wx1, wy1 = w1[1], wy0 = 1.f - wy1;
        float acc2[2] = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float* p = lds + c * ROWS * 64;
          float v[2][2];
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const int r = r00 + cy;
            v[cy][0] = okx0 ? p[r * 64 + lx0] : 0.f;
            v[cy][1] = okx1 ? p[r * 64 + lx1] : 0.f;
          }
          const float ux = c == 0 ? 1.f : 0.f, uyy = c == 1 ? 1.f : 0.f;
          const float dx = (v[0][1] - v[0][0] + ux) * wy0 + (v[1][1] - v[1][0] + ux) * wy1;
          const float dy = (v[1][0] - v[0][0] + uyy) * wx0 + (v[1][1] - v[0][1] + uyy) * wx1;
          const float kc = go[c] * (2.f / (float)(S[c] - 1));
          acc2[0] = fmaf(dx, kc, acc2[0]); acc2[1] = fmaf(dy, kc, acc2[1]);
        }
        // (a product that is NOT contracted into the final `(up + dn) + gg`: in k_adjoint_gather it is formed inside the
        // `regular` branch, another basic block than the sum)
        gg[0] = mul_nc(mult[0], acc2[0]);
        gg[1] = mul_nc(mult[1], acc2[1]);
      }
      // ---- phase B: gather the deposits of the 3 x 3 neighbouring samples
      float acc[2][3];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) acc[c][kk] = 0.f;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int r = rc + dy;                            // sample row uy + dy: this output sits at offset -dy from it
        float fx = lds[(0 * ROWS + r) * 64 + lane];
        float fy = lds[(1 * ROWS + r) * 64 + lane];
        fx = __builtin_amdgcn_fmed3f(fx, xlo, xhi);
        fy = __builtin_amdgcn_fmed3f(fy, -(float)(uy + dy), (float)(d.s1 - 1 - uy - dy));     // wave-uniform bounds
        const float w = fmaxf(0.f, 1.f - fabsf(fy + (float)dy));
        float a[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) a[c] = lds[((2 + c) * ROWS + r) * 64 + lane] * w;
        const float t0 = tent_f<-1>(fx), t1 = tent_f<0>(fx), t2 = tent_f<1>(fx);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[c][0] = fmaf(a[c], t0, acc[c][0]);
          acc[c][1] = fmaf(a[c], t1, acc[c][1]);
          acc[c][2] = fmaf(a[c], t2, acc[c][2]);
        }
      }
      // fold the x partial sums into the owning lanes: out(x) = sum_k acc[k](lane x - k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float up = acc[c][2];                             // deposits on x + 1
        up = lane_prev_f(up) + acc[c][1];
        float dn = lane_next_f(acc[c][0]);                // deposits on x - 1
        res[j][c] = (up + dn) + gg[c];
      }
    }
    if (L == 1) {
      // ---- the last level: grad(phi_0) of the owned rows and lanes
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const int lr = wave + j * NW;
        const int uy = y0 + lr;
        if (lr >= TY || uy >= d.s1) continue;
        if (xowned) {
          g0n[uy * d.s2 + sx] = res[j][0];
          g0n[V + uy * d.s2 + sx] = res[j][1];
        }
      }
      break;
    }
    __syncthreads();                                      // every read of this level is done
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int lr = wave + j * NW;
      const int rc = rfirst + lr;
      const int uy = ry0 + rc;
      if (lr >= count) continue;
      const bool ok = in_x && uy >= 0 && uy < d.s1;       // samples outside the image do not exist: zero gradient
      lds[(2 * ROWS + rc) * 64 + lane] = ok ? res[j][0] : 0.f;
      lds[(3 * ROWS + rc) * 64 + lane] = ok ? res[j][1] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e >= ROWS * 16) continue;
      const int q = e & 15, r = e >> 4;
      store_vec<4>(lds + (0 * ROWS + r) * 64 + 4 * q, nxt[i][0]);
      store_vec<4>(lds + (1 * ROWS + r) * 64 + 4 * q, nxt[i][1]);
    }
    __syncthreads();
  }
}

}  // namespace advchain

using namespace advchain;

// grad(phi_k) -> grad(phi_0) through the squarings 0..k-1 of a 2D chain (2 <= k <= 4), every one with an EXACT displacement
// bound below one pixel.  ADVCHAIN_ERR_UNSUPPORTED when the shape does not fit (rows of 4j >= 8 pixels, 16-byte aligned
// pointers); gk, g0 and the fields must not alias.
int advchain_adjoint_fused2d_launch(const float* gk, const float* phi0, const float* fields, float* g0, int64_t N, Dims d, int k,
                                    int32_t* workspace, hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_FUSE2D_BWD") != nullptr;   // A/B knob
  if (off || d.s0 != 1 || k < 2 || k > kFuseBwdMax || !workspace) return ADVCHAIN_ERR_UNSUPPORTED;
  if ((d.s2 & 3) != 0 || d.s2 < 8) return ADVCHAIN_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(gk) | reinterpret_cast<uintptr_t>(phi0) | reinterpret_cast<uintptr_t>(fields) |
                       reinterpret_cast<uintptr_t>(g0);
  if (al & 15) return ADVCHAIN_ERR_UNSUPPORTED;
  constexpr int TY = 16, NT = 256;
  const int wide = d.s2 > 64;
  const int n2 = wide ? (d.s2 + (64 - 2 * kFXPad) - 1) / (64 - 2 * kFXPad) : 1;
  const int n1 = (d.s1 + TY - 1) / TY;
  const int64_t F = N * 2 * d.voxels();
  const size_t lds = (size_t)4 * (TY + 2 * k) * 64 * sizeof(float);
  hipLaunchKernelGGL((k_adjoint_fused2d<TY, NT>), dim3((unsigned)(n1 * n2), (unsigned)N), dim3(NT), lds, st, gk, phi0, fields, F,
                     g0, d, k, n1, n2, wide, workspace);
  return ADVCHAIN_OK;
}
