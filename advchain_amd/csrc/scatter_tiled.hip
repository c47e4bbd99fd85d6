// Owner-computes, LDS-tiled scatter for the sampler backward passes (gfx950).
//
// grid_sampler backward is a scatter-add: sample s deposits w*grad_out[s] on the 2^d corners of its
// sampling position.  Doing that with global fp32 atomics caps the kernel at ~40 G atomics/s (measured:
// 180 GB/s on grid_sample3d bwd, 60 GB/s on the 3-channel self-composition).  The warps on this path are
// near-identity (|displacement| ~ 0.1-7 voxels), so instead:
//
//   * a workgroup OWNS a tile of the gradient tensor and keeps it in LDS (C x tile floats);
//   * it walks every sample s of the tile plus a halo of H voxels, recomputes the taps of s, and
//     accumulates (ds_add_f32) only the corners that fall inside its own tile -- halo samples are processed
//     redundantly by the neighbouring owners (their reads hit L2), nothing is communicated;
//   * the tile is written back with plain coalesced stores: no global atomics, no zero-fill pass.
//
// A deposit (s -> u) whose sample lies outside the halo box of u's tile cannot be seen by u's owner; the
// owner of s detects that with the same box test and appends s to an overflow list, which a second (usually
// empty) launch drains with global atomics after the tiles have been stored.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

struct TileCfg {
  int t0, t1, t2;   // tile extent (z, y, x)
  int h0, h1, h2;   // halo
  int n0, n1, n2;   // number of tiles per axis
};

__device__ __forceinline__ bool in_tile_box(int s, int u, int T, int H) {
  const int lo = (u / T) * T;
  return (s >= lo - H) && (s < lo + T + H);
}

// true when the owner of corner u processes sample s (so the deposit is handled in LDS)
__device__ __forceinline__ bool deposit_handled(int sz, int sy, int sx, int uz, int uy, int ux, const TileCfg& tc) {
  return in_tile_box(sx, ux, tc.t2, tc.h2) && in_tile_box(sy, uy, tc.t1, tc.h1) && in_tile_box(sz, uz, tc.t0, tc.h0);
}

__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// SELF: input == grid == phi with C == DIM channels, and the coordinate-path gradient is added to the same
//       output (advchain_compose_self_bwd).  Otherwise grad_in -> gin tile, grad_grid -> ggrid (plain stores).
template <int DIM, int PAD, bool SELF, bool NEED_GGRID>
__global__ void __launch_bounds__(kBlock)
k_scatter_tiled(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                float* __restrict__ gin, float* __restrict__ ggrid, int C, Dims d, TileCfg tc, int clamp_grid,
                int* __restrict__ ovf_count, int2* __restrict__ ovf_list, int ovf_cap, int dbg) {
  extern __shared__ float lds[];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  // tile coordinates
  int b = blockIdx.x;
  const int tx = b % tc.n2; b /= tc.n2;
  const int ty = b % tc.n1;
  const int tz = b / tc.n1;
  const int x0 = tx * tc.t2, y0 = ty * tc.t1, z0 = tz * tc.t0;
  const int tvox = tc.t0 * tc.t1 * tc.t2;
  for (int i = threadIdx.x; i < C * tvox; i += kBlock) lds[i] = 0.f;
  __syncthreads();
  // source region = tile + halo, clipped to the volume
  const int rx0 = max(x0 - tc.h2, 0), rx1 = min(x0 + tc.t2 + tc.h2, d.s2);
  const int ry0 = max(y0 - tc.h1, 0), ry1 = min(y0 + tc.t1 + tc.h1, d.s1);
  const int rz0 = max(z0 - tc.h0, 0), rz1 = min(z0 + tc.t0 + tc.h0, d.s0);
  const int rw = rx1 - rx0, rh = ry1 - ry0, rd = rz1 - rz0;
  const int rvox = rw * rh * rd;
  const float* gn = grid + (int64_t)n * DIM * V;
  const float* inn = in + (int64_t)n * C * V;
  const float* gon = gout + (int64_t)n * C * V;
  for (int r = threadIdx.x; r < rvox; r += kBlock) {
    const int lx = r % rw;
    const int q = r / rw;
    const int ly = q % rh;
    const int lz = q / rh;
    const int sx = rx0 + lx, sy = ry0 + ly, sz = rz0 + lz;
    const int s = (sz * d.s1 + sy) * d.s2 + sx;
    float gx = gn[s], gy = gn[V + s], gz = DIM == 3 ? gn[2 * V + s] : 0.f;
    bool px = true, py = true, pz = true;
    if (clamp_grid) {
      px = gx >= -1.f && gx <= 1.f; py = gy >= -1.f && gy <= 1.f; pz = gz >= -1.f && gz <= 1.f;
      gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz);
    }
    Taps<DIM, PAD> t;
    t.build(gx, gy, gz, d);
    const bool owned = (sx >= x0) && (sx < x0 + tc.t2) && (sy >= y0) && (sy < y0 + tc.t1) && (sz >= z0) && (sz < z0 + tc.t0);
    bool overflow = false;
    // corner bookkeeping (shared by all channels)
    int loff[8];
    float w[8];
#pragma unroll
    for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const int k = (cz * 2 + cy) * 2 + cx;
          loff[k] = -1;
          w[k] = 0.f;
          if (t.ok(cz, cy, cx)) {
            const int ux = t.x.i0 + cx, uy = t.y.i0 + cy, uz = t.z.i0 + cz;
            const bool mine = (ux >= x0) && (ux < x0 + tc.t2) && (uy >= y0) && (uy < y0 + tc.t1) && (uz >= z0) && (uz < z0 + tc.t0);
            if (mine) {
              loff[k] = ((uz - z0) * tc.t1 + (uy - y0)) * tc.t2 + (ux - x0);
              w[k] = t.w(cz, cy, cx);
            } else if (owned && !deposit_handled(sz, sy, sx, uz, uy, ux, tc)) {
              overflow = true;
            }
          }
        }
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int c = 0; c < C; ++c) {
      const float go = gon[(int64_t)c * V + s];
      float* tile = lds + c * tvox;
#pragma unroll
      for (int k = 0; k < (DIM == 3 ? 8 : 4); ++k)
        if (loff[k] >= 0 && !(dbg & 1)) lds_add(tile + loff[k], w[k] * go);
      if (owned && (SELF || NEED_GGRID) && !(dbg & 2)) {
        float dummy = 0.f;
        sample_linear_bwd<DIM, PAD, false, true>(inn + (int64_t)c * V, nullptr, go, t, d, ax, ay, DIM == 3 ? az : dummy);
      }
    }
    if (owned) {
      const float ggx = px ? t.x.mult * ax : 0.f;
      const float ggy = py ? t.y.mult * ay : 0.f;
      const float ggz = (DIM == 3 && pz) ? t.z.mult * az : 0.f;
      if (SELF) {
        const int lo = ((sz - z0) * tc.t1 + (sy - y0)) * tc.t2 + (sx - x0);
        if (ggx != 0.f) lds_add(lds + lo, ggx);
        if (ggy != 0.f) lds_add(lds + tvox + lo, ggy);
        if (DIM == 3 && ggz != 0.f) lds_add(lds + 2 * tvox + lo, ggz);
      } else if (NEED_GGRID) {
        float* gg = ggrid + (int64_t)n * DIM * V + s;
        gg[0] = ggx;
        gg[V] = ggy;
        if (DIM == 3) gg[2 * V] = ggz;
      }
      if (overflow) {
        const int slot = atomicAdd(ovf_count, 1);
        if (slot < ovf_cap) ovf_list[slot] = make_int2(n, s);
      }
    }
  }
  __syncthreads();
  // flush the tile (plain, coalesced along x)
  float* ginn = gin + (int64_t)n * C * V;
  for (int i = threadIdx.x; i < C * tvox; i += kBlock) {
    const int c = i / tvox;
    const int l = i - c * tvox;
    const int lx = l % tc.t2;
    const int q = l / tc.t2;
    const int ly = q % tc.t1;
    const int lz = q / tc.t1;
    const int ux = x0 + lx, uy = y0 + ly, uz = z0 + lz;
    if (ux < d.s2 && uy < d.s1 && uz < d.s0) ginn[(int64_t)c * V + (uz * d.s1 + uy) * d.s2 + ux] = lds[i];
  }
}

// Drains the overflow list with global atomics (runs after the tiles were stored).
template <int DIM, int PAD>
__global__ void __launch_bounds__(kBlock)
k_scatter_overflow(const float* __restrict__ gout, const float* __restrict__ grid, float* __restrict__ gin, int C,
                   Dims d, TileCfg tc, int clamp_grid, const int* __restrict__ ovf_count,
                   const int2* __restrict__ ovf_list, int ovf_cap) {
  const int V = (int)d.voxels();
  const int count = min(*ovf_count, ovf_cap);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    const int2 e = ovf_list[i];
    const int n = e.x, s = e.y;
    const int sx = s % d.s2;
    const int q = s / d.s2;
    const int sy = q % d.s1;
    const int sz = q / d.s1;
    const float* gn = grid + (int64_t)n * DIM * V;
    float gx = gn[s], gy = gn[V + s], gz = DIM == 3 ? gn[2 * V + s] : 0.f;
    if (clamp_grid) { gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz); }
    Taps<DIM, PAD> t;
    t.build(gx, gy, gz, d);
#pragma unroll
    for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          if (!t.ok(cz, cy, cx)) continue;
          const int ux = t.x.i0 + cx, uy = t.y.i0 + cy, uz = t.z.i0 + cz;
          if (deposit_handled(sz, sy, sx, uz, uy, ux, tc)) continue;
          const int o = (uz * d.s1 + uy) * d.s2 + ux;
          const float wgt = t.w(cz, cy, cx);
          for (int c = 0; c < C; ++c)
            atomic_add_f32(gin + ((int64_t)n * C + c) * V + o, wgt * gout[((int64_t)n * C + c) * V + s]);
        }
  }
}

}  // namespace advchain

using namespace advchain;

// Tile geometry.  Near-identity warps: 3D displacements are <~ 2 voxels, 2D <~ 8 pixels; anything
// larger goes through the overflow list.  LDS budget: C * tile * 4 B <= 64 KiB (2 workgroups per CU).
static TileCfg choose_tiles(int ndim, const Dims& d, int C) {
  TileCfg tc;
  if (ndim == 3) {
    tc.t2 = d.s2 < 64 ? d.s2 : 64;
    tc.t1 = 8;
    tc.t0 = 8;
    tc.h0 = tc.h1 = tc.h2 = 2;
  } else {
    tc.t2 = d.s2 < 64 ? d.s2 : 64;
    tc.t1 = 32;
    tc.t0 = 1;
    tc.h0 = 0;
    tc.h1 = tc.h2 = 8;
  }
  if (tc.t1 > d.s1) tc.t1 = d.s1;
  if (tc.t0 > d.s0) tc.t0 = d.s0;
  while ((int64_t)C * tc.t0 * tc.t1 * tc.t2 * 4 > 65536) {
    if (tc.t0 > 1) tc.t0 = (tc.t0 + 1) / 2;
    else if (tc.t1 > 1) tc.t1 = (tc.t1 + 1) / 2;
    else tc.t2 = (tc.t2 + 1) / 2;
  }
  tc.n2 = (d.s2 + tc.t2 - 1) / tc.t2;
  tc.n1 = (d.s1 + tc.t1 - 1) / tc.t1;
  tc.n0 = (d.s0 + tc.t0 - 1) / tc.t0;
  return tc;
}

template <int DIM, int PAD>
static void launch_tiled(bool self, bool need_ggrid, dim3 g, size_t lds, hipStream_t st, const float* gout,
                         const float* in, const float* grid, float* gin, float* ggrid, int C, Dims d, TileCfg tc,
                         int clamp_grid, int* cnt, int2* list, int cap) {
  static const int dbg = getenv("ADVCHAIN_DBG") ? atoi(getenv("ADVCHAIN_DBG")) : 0;  // tuning knob
  if (self) hipLaunchKernelGGL((k_scatter_tiled<DIM, PAD, true, false>), g, dim3(kBlock), lds, st, gout, in, grid, gin, ggrid, C, d, tc, clamp_grid, cnt, list, cap, dbg);
  else if (need_ggrid) hipLaunchKernelGGL((k_scatter_tiled<DIM, PAD, false, true>), g, dim3(kBlock), lds, st, gout, in, grid, gin, ggrid, C, d, tc, clamp_grid, cnt, list, cap, dbg);
  else hipLaunchKernelGGL((k_scatter_tiled<DIM, PAD, false, false>), g, dim3(kBlock), lds, st, gout, in, grid, gin, ggrid, C, d, tc, clamp_grid, cnt, list, cap, dbg);
  hipLaunchKernelGGL((k_scatter_overflow<DIM, PAD>), dim3(64), dim3(kBlock), 0, st, gout, grid, gin, C, d, tc, clamp_grid, cnt, list, cap);
}

// Shared entry used by advchain_grid_sample_bwd_tiled / advchain_compose_self_bwd_tiled.
// workspace: int32[2 + 2*N*V]: [0] = overflow counter (zeroed here), [2..] = (n, s) pairs.
int advchain_scatter_tiled_launch(bool self, const float* gout, const float* in, const float* grid, float* gin,
                                  float* ggrid, int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                  int32_t* workspace, hipStream_t st) {
  const TileCfg tc = choose_tiles(ndim, d, (int)C);
  const int64_t V = d.voxels();
  int* cnt = workspace;
  int2* list = reinterpret_cast<int2*>(workspace + 2);
  const int64_t cap64 = N * V;
  const int cap = cap64 > 0x7fffffff ? 0x7fffffff : (int)cap64;
  (void)hipMemsetAsync(cnt, 0, 2 * sizeof(int32_t), st);
  dim3 g((unsigned)(tc.n0 * tc.n1 * tc.n2), (unsigned)N);
  const size_t lds = (size_t)C * tc.t0 * tc.t1 * tc.t2 * sizeof(float);
  const bool need_ggrid = ggrid != nullptr;
  if (ndim == 3) {
    switch (padding) {
      case PAD_ZEROS: launch_tiled<3, PAD_ZEROS>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
      case PAD_BORDER: launch_tiled<3, PAD_BORDER>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
      default: launch_tiled<3, PAD_REFLECTION>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
    }
  } else {
    switch (padding) {
      case PAD_ZEROS: launch_tiled<2, PAD_ZEROS>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
      case PAD_BORDER: launch_tiled<2, PAD_BORDER>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
      default: launch_tiled<2, PAD_REFLECTION>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, (int)C, d, tc, clamp_grid, cnt, list, cap); break;
    }
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

extern "C" int64_t advchain_scatter_workspace(int64_t N, int ndim, const int64_t* dims) {
  int64_t V = 1;
  for (int i = 0; i < ndim; ++i) V *= dims[i];
  return 2 + 2 * N * V;  // int32 elements
}
