// Owner-computes, LDS-tiled scatter for the sampler backward passes (gfx950).
//
// grid_sampler backward is a scatter-add: sample s deposits w*grad_out[s] on the 2^d corners of its sampling
// position.  Measured on MI355X: global fp32 atomics cap it at ~40 G atomics/s (180 GB/s on grid_sample3d
// bwd, 60 GB/s on the 3-channel self-composition); LDS *float* atomics are no better (ds_add_f32: ~194 clk
// per wave instruction) -- but LDS *integer* atomics run at full LDS rate (ds_add_u64: <= 16 clk).  The warps
// on this path are near-identity (|displacement| ~ 0.1-7 voxels), so:
//
//   * a workgroup OWNS a tile of the gradient tensor and accumulates it in LDS as 64-bit fixed point
//     (value * 2^30 / max|grad_out|; exact integer adds => the result does not depend on the order of the
//     deposits, i.e. the kernel is deterministic, unlike any float-atomic formulation);
//   * wave w walks the rows w, w+NW, ... of the tile plus a halo of H voxels (lane = x), prefetching the next
//     row while it processes the current one; it recomputes the taps of every sample and accumulates only
//     the corners that fall inside its own tile -- halo samples are processed redundantly by the neighbouring
//     owners (their reads hit L2), nothing is communicated;
//   * the tile is converted back and written with plain coalesced stores: no global atomics, no zero-fill.
//
// A deposit (s -> u) whose sample lies outside the halo box of u's tile cannot be seen by u's owner; the
// owner of s detects that with the same box test and appends s to an overflow list, which a second (usually
// empty) launch drains with global atomics after the tiles have been stored.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

struct TileCfg {
  int t0, t1, t2;   // owned tile extent (z, y, x)
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

constexpr float kFixedOne = 1073741824.f;  // 2^30

__device__ __forceinline__ void lds_add_fixed(long long* p, float v) {
  const long long q = (long long)__float2int_rn(v);
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);  // ds_add_u64
}

// same for values that may exceed max|grad_out| by a large factor (coordinate-path gradient): two-part conversion
__device__ __forceinline__ void lds_add_fixed_wide(long long* p, float v) {
  const float hi = rintf(v * (1.f / 65536.f));
  const long long q = ((long long)__float2int_rn(hi) << 16) + (long long)__float2int_rn(v - hi * 65536.f);
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}

__device__ __forceinline__ void atomic_max_abs(float* addr, float v) {
  // non-negative floats order like their bit patterns.  Same-address atomics serialise at ~10 ns each on
  // MI355X (measured: 8192 of them = 80 us), so skip the atomic when the plain read already dominates.
  v = fabsf(v);
  if (v > __builtin_nontemporal_load(addr)) atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// max over the workgroup (result valid in thread 0); smem >= NT/64 floats
template <int NT>
__device__ __forceinline__ float block_max(float m, float* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, smem[w]);
  return m;
}

// |x|_max of a buffer -> *out (caller zeroes *out).  `needed` (optional): skip the pass when *needed == 0.
__global__ void __launch_bounds__(kBlock) k_absmax(const float* __restrict__ x, int64_t n, float* __restrict__ out, int vec,
                                                   const int* __restrict__ needed) {
  if (needed && *needed == 0) return;
  float m = 0.f;
  const int64_t n4 = vec ? (n >> 2) : 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 q = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) m = fmaxf(m, fabsf(x[i]));
  __shared__ float smem[kBlock / 64];
  m = block_max<kBlock>(m, smem);
  if (threadIdx.x == 0 && m > 0.f) atomic_max_abs(out, m);
}

// workspace header: [0] overflow counter, [1] "max|grad_out| still to be computed", [2] max|grad_out|, [3] max|result|
// ([3] == -1: the previous launch of the chain was one that does not track it -- the window scatter)
__global__ void k_scatter_prepare(int32_t* ws, int chain) {
  const bool have = chain && ws[3] != -1;
  ws[2] = have ? ws[3] : 0;
  ws[1] = have ? 0 : 1;
  ws[0] = 0;
  ws[3] = 0;
}

template <int DIM, int C>
struct RowRegs {
  float g[DIM];
  float go[C];
};

// SELF: input == grid == phi with C == DIM channels, and the coordinate-path gradient is added to the same
//       output (advchain_compose_self_bwd).  Otherwise grad_in -> gin tile, grad_grid -> ggrid (plain stores).
template <int DIM, int PAD, int C, bool SELF, bool NEED_GGRID, int NT>
__global__ void __launch_bounds__(NT)
k_scatter_rows(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
               float* __restrict__ gin, float* __restrict__ ggrid, Dims d, TileCfg tc, int clamp_grid,
               const float* __restrict__ absmax_in, float* __restrict__ absmax_out, int* __restrict__ ovf_count,
               int2* __restrict__ ovf_list, int ovf_cap) {
  extern __shared__ long long acc[];
  constexpr int NW = NT / 64;
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = blockIdx.x;
  const int tx = b % tc.n2; b /= tc.n2;
  const int ty = b % tc.n1;
  const int tz = b / tc.n1;
  const int x0 = tx * tc.t2, y0 = ty * tc.t1, z0 = tz * tc.t0;
  const int tvox = tc.t0 * tc.t1 * tc.t2;
  for (int i = threadIdx.x; i < C * tvox; i += NT) acc[i] = 0;
  const float amax = absmax_in[0];
  const float scale = amax > 0.f ? kFixedOne / amax : 0.f;
  __syncthreads();
  // source region = tile + halo, clipped to the volume; lane <-> x
  const int rx0 = max(x0 - tc.h2, 0), rx1 = min(x0 + tc.t2 + tc.h2, d.s2);
  const int ry0 = max(y0 - tc.h1, 0), ry1 = min(y0 + tc.t1 + tc.h1, d.s1);
  const int rz0 = max(z0 - tc.h0, 0), rz1 = min(z0 + tc.t0 + tc.h0, d.s0);
  const int rh = ry1 - ry0, rd = rz1 - rz0;
  const int nrows = rh * rd;
  const int sx = rx0 + lane;
  const bool xvalid = sx < rx1;
  const bool xowned = xvalid && (sx >= x0) && (sx < x0 + tc.t2);
  const float* gn = grid + (int64_t)n * DIM * V;
  const float* inn = in + (int64_t)n * C * V;
  const float* gon = gout + (int64_t)n * C * V;

  auto load_row = [&](int row, RowRegs<DIM, C>& r) {
    const int sy = ry0 + row % rh, sz = rz0 + row / rh;
    const int s = (sz * d.s1 + sy) * d.s2 + sx;
#pragma unroll
    for (int a = 0; a < DIM; ++a) r.g[a] = xvalid ? gn[(int64_t)a * V + s] : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) r.go[c] = xvalid ? gon[(int64_t)c * V + s] : 0.f;
  };

  RowRegs<DIM, C> cur, nxt;
  int row = wave;
  if (row < nrows) load_row(row, cur);
  while (row < nrows) {
    const int nrow = row + NW;
    if (nrow < nrows) load_row(nrow, nxt);   // prefetch: in flight while this row is processed
    const int sy = ry0 + row % rh, sz = rz0 + row / rh;
    const int s = (sz * d.s1 + sy) * d.s2 + sx;
    float gx = cur.g[0], gy = cur.g[1], gz = DIM == 3 ? cur.g[DIM - 1] : 0.f;
    bool px = true, py = true, pz = true;
    if (clamp_grid) {
      px = gx >= -1.f && gx <= 1.f; py = gy >= -1.f && gy <= 1.f; pz = gz >= -1.f && gz <= 1.f;
      gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz);
    }
    const bool rowowned = (sy >= y0) && (sy < y0 + tc.t1) && (sz >= z0) && (sz < z0 + tc.t0);  // wave-uniform
    if (!rowowned && PAD != PAD_REFLECTION) {
      // halo row: skip it when no lane's sampling position comes within one voxel of the tile (cheap test first)
      float qx = ((gx + 1.f) * 0.5f) * (float)(d.s2 - 1), qy = ((gy + 1.f) * 0.5f) * (float)(d.s1 - 1);
      float qz = DIM == 3 ? ((gz + 1.f) * 0.5f) * (float)(d.s0 - 1) : 0.f;
      if (PAD == PAD_BORDER) {
        qx = fminf(fmaxf(qx, 0.f), (float)(d.s2 - 1)); qy = fminf(fmaxf(qy, 0.f), (float)(d.s1 - 1));
        qz = fminf(fmaxf(qz, 0.f), (float)(d.s0 - 1));
      }
      const bool hit = xvalid && qx > (float)(x0 - 1) && qx < (float)(x0 + tc.t2) && qy > (float)(y0 - 1) &&
                       qy < (float)(y0 + tc.t1) && (DIM < 3 || (qz > (float)(z0 - 1) && qz < (float)(z0 + tc.t0)));
      if (__builtin_amdgcn_ballot_w64(hit) == 0) {
        cur = nxt;
        row = nrow;
        continue;
      }
    }
    Taps<DIM, PAD> t;
    t.build(gx, gy, gz, d);
    const bool owned = xowned && rowowned;
    bool overflow = false;
    if (xvalid) {
#pragma unroll
      for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            if (t.ok(cz, cy, cx)) {
              const int ux = t.x.i0 + cx, uy = t.y.i0 + cy, uz = t.z.i0 + cz;
              const bool mine = (ux >= x0) && (ux < x0 + tc.t2) && (uy >= y0) && (uy < y0 + tc.t1) && (uz >= z0) && (uz < z0 + tc.t0);
              if (mine) {
                const int lo = ((uz - z0) * tc.t1 + (uy - y0)) * tc.t2 + (ux - x0);
                const float w = t.w(cz, cy, cx) * scale;
#pragma unroll
                for (int c = 0; c < C; ++c) lds_add_fixed(acc + c * tvox + lo, w * cur.go[c]);
              } else if (owned && !deposit_handled(sz, sy, sx, uz, uy, ux, tc)) {
                overflow = true;
              }
            }
          }
    }
    if (owned) {
      if (SELF || NEED_GGRID) {
        float ax = 0.f, ay = 0.f, az = 0.f, dummy = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
          sample_linear_bwd<DIM, PAD, false, true>(inn + (int64_t)c * V, nullptr, cur.go[c], t, d, ax, ay, DIM == 3 ? az : dummy);
        const float ggx = px ? t.x.mult * ax : 0.f;
        const float ggy = py ? t.y.mult * ay : 0.f;
        const float ggz = (DIM == 3 && pz) ? t.z.mult * az : 0.f;
        if (SELF) {
          const int lo = ((sz - z0) * tc.t1 + (sy - y0)) * tc.t2 + (sx - x0);
          if (ggx != 0.f) lds_add_fixed_wide(acc + lo, ggx * scale);
          if (ggy != 0.f) lds_add_fixed_wide(acc + tvox + lo, ggy * scale);
          if (DIM == 3 && ggz != 0.f) lds_add_fixed_wide(acc + 2 * tvox + lo, ggz * scale);
        } else {
          float* gg = ggrid + (int64_t)n * DIM * V + s;
          gg[0] = ggx;
          gg[V] = ggy;
          if (DIM == 3) gg[2 * V] = ggz;
        }
      }
      if (overflow) {
        const int slot = atomicAdd(ovf_count, 1);
        if (slot < ovf_cap) ovf_list[slot] = make_int2(n, s);
      }
    }
    cur = nxt;
    row = nrow;
  }
  __syncthreads();
  // flush the tile (plain, coalesced along x) and track max|value| for the next launch of a chain
  const float inv = amax * (1.f / kFixedOne);
  float* ginn = gin + (int64_t)n * C * V;
  float m = 0.f;
  for (int i = threadIdx.x; i < C * tvox; i += NT) {
    const int c = i / tvox;
    const int l = i - c * tvox;
    const int lx = l % tc.t2;
    const int q = l / tc.t2;
    const int ly = q % tc.t1;
    const int lz = q / tc.t1;
    const int ux = x0 + lx, uy = y0 + ly, uz = z0 + lz;
    if (ux < d.s2 && uy < d.s1 && uz < d.s0) {
      const float v = (float)acc[i] * inv;
      ginn[(int64_t)c * V + (uz * d.s1 + uy) * d.s2 + ux] = v;
      m = fmaxf(m, fabsf(v));
    }
  }
  if (absmax_out) {
    __shared__ float smem[NT / 64];
    m = block_max<NT>(m, smem);
    if (threadIdx.x == 0 && m > 0.f) atomic_max_abs(absmax_out, m);
  }
}

// Drains the overflow list with global atomics (runs after the tiles were stored).
template <int DIM, int PAD>
__global__ void __launch_bounds__(kBlock)
k_scatter_overflow(const float* __restrict__ gout, const float* __restrict__ grid, float* __restrict__ gin, int C,
                   Dims d, TileCfg tc, int clamp_grid, const int* __restrict__ ovf_count,
                   const int2* __restrict__ ovf_list, int ovf_cap, float* __restrict__ absmax_out) {
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
          for (int c = 0; c < C; ++c) {
            const float v = wgt * gout[((int64_t)n * C + c) * V + s];
            const float old = atomicAdd(gin + ((int64_t)n * C + c) * V + o, v);
            if (absmax_out) atomic_max_abs(absmax_out, old + v);
          }
        }
  }
}

}  // namespace advchain

using namespace advchain;

// Tile geometry.  Near-identity warps: 3D displacements are <~ 2 voxels, 2D <~ 8 pixels at initialisation and up to
// ~16 after a few un-normalised ascent steps; anything larger goes through the overflow list.  lane <-> x: a row of the region (owned x-range + halo) is one wave.
static TileCfg choose_tiles(int ndim, const Dims& d, int C, int halo_hint) {
  TileCfg tc;
  static const int h3d = 2;   // measured optimum (was a tuning knob until round 4)
  static const int h2d = 16;  // measured: 16 beats 8/12 at cfg-2
  if (ndim == 3) {
    tc.h0 = tc.h1 = tc.h2 = h3d;
    tc.t1 = 8;
    tc.t0 = 8;
  } else {
    tc.h0 = 0;
    tc.h1 = tc.h2 = h2d;
    tc.t1 = 32;
    tc.t0 = 1;
  }
  if (halo_hint > 0) {
    tc.h1 = tc.h2 = halo_hint;
    if (ndim == 3) tc.h0 = tc.h1 = tc.h2 = halo_hint > 4 ? 4 : halo_hint;   // wider 3D halos cost more than the overflow list
  }
  if (d.s2 <= 64) {
    tc.t2 = d.s2;  // the whole row: no x-halo needed
  } else {
    if (tc.h2 > 24) tc.h2 = 24;
    tc.t2 = 64 - 2 * tc.h2;
  }
  if (tc.t1 > d.s1) tc.t1 = d.s1;
  if (tc.t0 > d.s0) tc.t0 = d.s0;
  static const int lds_cap = 65536;  // measured optimum (was a tuning knob until round 4)
  while ((int64_t)C * tc.t0 * tc.t1 * tc.t2 * 8 > lds_cap) {   // int64 accumulators in dynamic LDS
    if (tc.t0 > 1) tc.t0 = (tc.t0 + 1) / 2;
    else tc.t1 = (tc.t1 + 1) / 2;
  }
  tc.n2 = (d.s2 + tc.t2 - 1) / tc.t2;
  tc.n1 = (d.s1 + tc.t1 - 1) / tc.t1;
  tc.n0 = (d.s0 + tc.t0 - 1) / tc.t0;
  return tc;
}

template <int DIM, int PAD, int C>
static void launch_rows(bool self, bool need_ggrid, dim3 g, size_t lds, hipStream_t st, const float* gout,
                        const float* in, const float* grid, float* gin, float* ggrid, Dims d, TileCfg tc,
                        int clamp_grid, const float* amax_in, float* amax_out, int* cnt, int2* list, int cap) {
  // > 48 KiB of LDS leaves <= 3 workgroups per CU: use 8 waves per workgroup to keep the CU busy
  const bool big = lds > 40960, huge = lds > 65536;
#define LAUNCH(SELF_, GG_, NT_)                                                                                      \
  do {                                                                                                               \
    auto kern = k_scatter_rows<DIM, PAD, C, SELF_, GG_, NT_>;                                                        \
    if (huge) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kern, g, dim3(NT_), lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax_in, amax_out, \
                       cnt, list, cap);                                                                              \
  } while (0)
  if (self) {
    if constexpr (C == DIM) { if (huge) LAUNCH(true, false, 1024); else if (big) LAUNCH(true, false, 512); else LAUNCH(true, false, 256); }
  } else if (need_ggrid) {
    if (huge) LAUNCH(false, true, 1024); else if (big) LAUNCH(false, true, 512); else LAUNCH(false, true, 256);
  } else {
    if (huge) LAUNCH(false, false, 1024); else if (big) LAUNCH(false, false, 512); else LAUNCH(false, false, 256);
  }
#undef LAUNCH
  hipLaunchKernelGGL((k_scatter_overflow<DIM, PAD>), dim3(64), dim3(kBlock), 0, st, gout, grid, gin, C, d, tc,
                     clamp_grid, cnt, list, cap, amax_out);
}

template <int DIM, int PAD>
static bool launch_rows_c(int C, bool self, bool need_ggrid, dim3 g, size_t lds, hipStream_t st, const float* gout,
                          const float* in, const float* grid, float* gin, float* ggrid, Dims d, TileCfg tc,
                          int clamp_grid, const float* amax_in, float* amax_out, int* cnt, int2* list, int cap) {
  switch (C) {
    case 1: launch_rows<DIM, PAD, 1>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax_in, amax_out, cnt, list, cap); return true;
    case 2: launch_rows<DIM, PAD, 2>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax_in, amax_out, cnt, list, cap); return true;
    case 3: launch_rows<DIM, PAD, 3>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax_in, amax_out, cnt, list, cap); return true;
    case 4: launch_rows<DIM, PAD, 4>(self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax_in, amax_out, cnt, list, cap); return true;
    default: return false;
  }
}

// Shared entry used by advchain_grid_sample_bwd / advchain_compose_self_bwd.
// workspace (int32): [0] overflow counter, [1] flag "max|grad_out| still to be computed", [2] max|grad_out| (float),
//                    [3] max|result| (float; the int -1 when the producing launch did not track it),
//                    [4..] (n, s) overflow pairs.
// chain = 0: max|grad_out| is computed here (one streaming pass over grad_out);
// chain = 1: the previous launch on this workspace produced grad_out and left max|grad_out| in [3]
//            (or -1 there if it was a kernel that does not track it: then the pass runs after all).
// Returns ADVCHAIN_ERR_UNSUPPORTED when C is outside 1..4 (caller falls back to the global-atomic kernels).
int advchain_scatter_tiled_launch(bool self, const float* gout, const float* in, const float* grid, float* gin,
                                  float* ggrid, int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                  int32_t* workspace, int chain, int halo, hipStream_t st) {
  if (C < 1 || C > 4) return ADVCHAIN_ERR_UNSUPPORTED;
  const TileCfg tc = choose_tiles(ndim, d, (int)C, halo < 0 ? -halo : halo);   // sign = exactness, used by the gather form only
  const int64_t V = d.voxels();
  int* cnt = workspace;
  float* amax = reinterpret_cast<float*>(workspace + 2);  // [0] = in, [1] = out
  int2* list = reinterpret_cast<int2*>(workspace + 4);
  const int64_t cap64 = N * V;
  const int cap = cap64 > 0x7fffffff ? 0x7fffffff : (int)cap64;
  hipLaunchKernelGGL(k_scatter_prepare, dim3(1), dim3(1), 0, st, workspace, chain);   // header reset in one launch
  {
    // chain: the pass runs only if the previous launch left no max behind (decided on the device: header [1])
    const int64_t total = N * C * V;
    int blocks = (int)((total / 4 + kBlock - 1) / kBlock);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(kBlock), 0, st, gout, total, amax,
                       (int)((reinterpret_cast<uintptr_t>(gout) & 15) == 0), chain ? workspace + 1 : (const int*)nullptr);
  }
  dim3 g((unsigned)(tc.n0 * tc.n1 * tc.n2), (unsigned)N);
  const size_t lds = (size_t)C * tc.t0 * tc.t1 * tc.t2 * sizeof(long long);
  const bool need_ggrid = ggrid != nullptr;
  bool ok;
#define GO(DIM_, PAD_) \
  ok = launch_rows_c<DIM_, PAD_>((int)C, self, need_ggrid, g, lds, st, gout, in, grid, gin, ggrid, d, tc, clamp_grid, amax, amax + 1, cnt, list, cap)
  if (ndim == 3) {
    switch (padding) {
      case PAD_ZEROS: GO(3, PAD_ZEROS); break;
      case PAD_BORDER: GO(3, PAD_BORDER); break;
      default: GO(3, PAD_REFLECTION); break;
    }
  } else {
    switch (padding) {
      case PAD_ZEROS: GO(2, PAD_ZEROS); break;
      case PAD_BORDER: GO(2, PAD_BORDER); break;
      default: GO(2, PAD_REFLECTION); break;
    }
  }
#undef GO
  if (!ok) return ADVCHAIN_ERR_UNSUPPORTED;
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

extern "C" int advchain_get_deterministic(void);
// int32 elements: header [4] + overflow list / row maxima [2 N V]; in deterministic mode followed by the int64 image of
// grad_in the window scatter accumulates into (up to 4 channels: 8 N V int32) and one max |grad_out| per batch entry
extern "C" int64_t advchain_scatter_workspace(int64_t N, int ndim, const int64_t* dims) {
  int64_t V = 1;
  for (int i = 0; i < ndim; ++i) V *= dims[i];
  const int64_t base = 4 + 2 * N * V;
  return advchain_get_deterministic() ? base + 8 * N * V + ((N + 3) & ~(int64_t)3) : base;
}
