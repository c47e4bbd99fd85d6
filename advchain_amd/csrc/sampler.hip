// (bi/tri)linear + nearest samplers, forward and backward, for gfx950.
//
//   advchain_grid_sample_{fwd,bwd}   <- F.grid_sample(data, grid, mode, padding, align_corners=True)
//                                       reference adv_morph.py:546-557 (dense-field image / prediction warp)
//   advchain_compose_self_{fwd,bwd}  <- F.grid_sample(phi, phi^T, 'border', align_corners=True)
//                                       reference adv_morph.py:179-202 (scaling-and-squaring step, input == grid)
//   advchain_affine_warp_{fwd,bwd}   <- F.affine_grid + F.grid_sample, reference adv_affine.py:289-314
//                                       (grid evaluated in registers, never materialised)
//
// All are HBM-bound gather/scatter kernels.  One thread owns UNR output voxels spaced one workgroup apart
// (voxel k of thread t = base + k*256 + t): every load/gather/store instruction of a wave then touches 64
// consecutive voxels (256 contiguous bytes for near-identity warps -- measured 1.4-1.9x faster than 16-byte
// per-lane ownership, whose gathers stride 16 B across lanes), while the UNR independent chains per thread
// supply the memory-level parallelism a dependent grid -> gather -> store sequence lacks.  Backward scatters
// go through the LDS-tiled owner-computes kernel (scatter_tiled.hip); the global-atomic kernels here are the
// fallback for resampling / nearest.  No MFMA: there is no contraction here.
#include <stdlib.h>
#include "sampler_common.h"
#include "affine_geo.h"

namespace advchain {

// strided ownership: element k of a thread lives at p[k * kBlock]; `n` = number of valid elements
template <int UNR>
__device__ __forceinline__ void load_str(const float* __restrict__ p, int n, float (&r)[UNR]) {
  // every load unconditional (a tail thread re-reads its own first element): conditional loads each get their own
  // branch and `s_waitcnt vmcnt(0)`, i.e. one serial memory round trip per element
#pragma unroll
  for (int k = 0; k < UNR; ++k) r[k] = p[(k < n ? k : 0) * kBlock];
#pragma unroll
  for (int k = 0; k < UNR; ++k) r[k] = k < n ? r[k] : 0.f;
}
template <int UNR>
__device__ __forceinline__ void store_str(float* __restrict__ p, int n, const float (&r)[UNR]) {
#pragma unroll
  for (int k = 0; k < UNR; ++k)
    if (k < n) p[k * kBlock] = r[k];
}
__device__ __forceinline__ int active_count(int64_t v0, int64_t total, int unr) {
  if (v0 >= total) return 0;
  const int64_t left = (total - v0 + kBlock - 1) / kBlock;
  return left < unr ? (int)left : unr;
}

// =============================================================================================
// generic grid_sample with a planar grid
// =============================================================================================
template <int DIM, int INTERP, int PAD, int VEC>
__global__ void __launch_bounds__(kBlock)
k_grid_sample_fwd(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out, int C,
                  Dims id, Dims od, int clamp_grid, const float* __restrict__ ride_in, float* __restrict__ ride_out,
                  int ride_nonzero) {
  const int64_t IV = id.voxels(), OV = od.voxels();
  const int n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * (kBlock * VEC) + threadIdx.x;
  const int na = active_count(v, OV, VEC);
  if (na == 0) return;
  const float* g = grid + (int64_t)n * DIM * OV + v;
  float gx[VEC], gy[VEC], gz[VEC];
  load_str<VEC>(g, na, gx);
  load_str<VEC>(g + OV, na, gy);
  if (DIM == 3) load_str<VEC>(g + 2 * OV, na, gz);
  const float* inn = in + (int64_t)n * C * IV;
  float* outn = out + (int64_t)n * C * OV + v;
  if (INTERP == INTERP_LINEAR) {
    Taps<DIM, PAD> t[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (clamp_grid) { gx[k] = clamp_unit(gx[k]); gy[k] = clamp_unit(gy[k]); if (DIM == 3) gz[k] = clamp_unit(gz[k]); }
      t[k].build(gx[k], gy[k], DIM == 3 ? gz[k] : 0.f, id);
    }
    for (int c = 0; c < C; ++c) {
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = sample_linear<DIM, PAD>(inn + (int64_t)c * IV, t[k], id);
      store_str<VEC>(outn + (int64_t)c * OV, na, r);
    }
    if (ride_out) {     // the rider: one more channel of another tensor through the same taps (advchain_grid_sample_fwd_ride)
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        r[k] = sample_linear<DIM, PAD>(ride_in + (int64_t)n * IV, t[k], id);
        if (ride_nonzero) r[k] = r[k] != 0.f ? 1.f : 0.f;
      }
      store_str<VEC>(ride_out + (int64_t)n * OV + v, na, r);
    }
  } else {
    int off[VEC];
    bool ok[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (clamp_grid) { gx[k] = clamp_unit(gx[k]); gy[k] = clamp_unit(gy[k]); if (DIM == 3) gz[k] = clamp_unit(gz[k]); }
      bool vx, vy, vz = true;
      const int ix = nearest_index<PAD>(gx[k], id.s2, vx);
      const int iy = nearest_index<PAD>(gy[k], id.s1, vy);
      const int iz = DIM == 3 ? nearest_index<PAD>(gz[k], id.s0, vz) : 0;
      ok[k] = vx && vy && vz;
      off[k] = (iz * id.s1 + iy) * id.s2 + ix;
    }
    for (int c = 0; c < C; ++c) {
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = ok[k] ? inn[(int64_t)c * IV + off[k]] : 0.f;
      store_str<VEC>(outn + (int64_t)c * OV, na, r);
    }
    if (ride_out) {
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        r[k] = ok[k] ? ride_in[(int64_t)n * IV + off[k]] : 0.f;
        if (ride_nonzero) r[k] = r[k] != 0.f ? 1.f : 0.f;
      }
      store_str<VEC>(ride_out + (int64_t)n * OV + v, na, r);
    }
  }
}

template <int DIM, int INTERP, int PAD, int VEC, bool NEED_GIN, bool NEED_GGRID>
__global__ void __launch_bounds__(kBlock)
k_grid_sample_bwd(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                  float* __restrict__ gin, float* __restrict__ ggrid, int C, Dims id, Dims od, int clamp_grid) {
  const int64_t IV = id.voxels(), OV = od.voxels();
  const int n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * (kBlock * VEC) + threadIdx.x;
  const int na = active_count(v, OV, VEC);
  if (na == 0) return;
  const float* g = grid + (int64_t)n * DIM * OV + v;
  float gx[VEC], gy[VEC], gz[VEC];
  load_str<VEC>(g, na, gx);
  load_str<VEC>(g + OV, na, gy);
  if (DIM == 3) load_str<VEC>(g + 2 * OV, na, gz);
  const float* inn = in + (int64_t)n * C * IV;
  float* ginn = NEED_GIN ? gin + (int64_t)n * C * IV : nullptr;
  const float* gon = gout + (int64_t)n * C * OV + v;
  if (INTERP == INTERP_LINEAR) {
    Taps<DIM, PAD> t[VEC];
    float ax[VEC], ay[VEC], az[VEC];
    bool pass_x[VEC], pass_y[VEC], pass_z[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      pass_x[k] = pass_y[k] = pass_z[k] = true;
      if (clamp_grid) {  // torch.clamp passes gradient on the closed interval [-1, 1]
        pass_x[k] = (gx[k] >= -1.f) && (gx[k] <= 1.f);
        pass_y[k] = (gy[k] >= -1.f) && (gy[k] <= 1.f);
        gx[k] = clamp_unit(gx[k]); gy[k] = clamp_unit(gy[k]);
        if (DIM == 3) { pass_z[k] = (gz[k] >= -1.f) && (gz[k] <= 1.f); gz[k] = clamp_unit(gz[k]); }
      }
      t[k].build(gx[k], gy[k], DIM == 3 ? gz[k] : 0.f, id);
      ax[k] = ay[k] = az[k] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
      float go[VEC];
      load_str<VEC>(gon + (int64_t)c * OV, na, go);
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        sample_linear_bwd<DIM, PAD, NEED_GIN, NEED_GGRID>(inn + (int64_t)c * IV, NEED_GIN ? ginn + (int64_t)c * IV : nullptr,
                                                          go[k], t[k], id, ax[k], ay[k], az[k]);
    }
    if (NEED_GGRID) {
      float* gg = ggrid + (int64_t)n * DIM * OV + v;
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = pass_x[k] ? t[k].x.mult * ax[k] : 0.f;
      store_str<VEC>(gg, na, r);
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = pass_y[k] ? t[k].y.mult * ay[k] : 0.f;
      store_str<VEC>(gg + OV, na, r);
      if (DIM == 3) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] = pass_z[k] ? t[k].z.mult * az[k] : 0.f;
        store_str<VEC>(gg + 2 * OV, na, r);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (clamp_grid) { gx[k] = clamp_unit(gx[k]); gy[k] = clamp_unit(gy[k]); if (DIM == 3) gz[k] = clamp_unit(gz[k]); }
      bool vx, vy, vz = true;
      const int ix = nearest_index<PAD>(gx[k], id.s2, vx);
      const int iy = nearest_index<PAD>(gy[k], id.s1, vy);
      const int iz = DIM == 3 ? nearest_index<PAD>(gz[k], id.s0, vz) : 0;
      if (NEED_GIN && k < na && vx && vy && vz) {
        const int off = (iz * id.s1 + iy) * id.s2 + ix;
        for (int c = 0; c < C; ++c) atomic_add_f32(ginn + (int64_t)c * IV + off, gon[(int64_t)c * OV + k * kBlock]);
      }
    }
    if (NEED_GGRID) {  // nearest has zero gradient w.r.t. the grid (ATen does the same)
      float* gg = ggrid + (int64_t)n * DIM * OV + v;
      float r[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = 0.f;
      for (int a = 0; a < DIM; ++a) store_str<VEC>(gg + (int64_t)a * OV, na, r);
    }
  }
}

// =============================================================================================
// self-composition  phi <- phi o phi  (input == grid, DIM channels, border padding)
//   final_mode: 0  out = sample
//               1  out = (sample - phi0) + identity   (last squaring: emits the sampling positions
//                  'integrated_offsets + base_grid' of adv_morph.py:143,176 + 474,483; Q1 aliasing)
// =============================================================================================
template <int DIM, int VEC>
__device__ __forceinline__ void
compose_self_fwd_body(const float* __restrict__ phi, float* __restrict__ out, const float* __restrict__ phi0,
                      const Dims& d, int final_mode, float* __restrict__ disp_out, int n, int64_t v) {
  const int64_t V = d.voxels();
  const int na = active_count(v, V, VEC);
  if (na == 0) return;
  const float* pn = phi + (int64_t)n * DIM * V;
  float gx[VEC], gy[VEC], gz[VEC];
  load_str<VEC>(pn + v, na, gx);
  load_str<VEC>(pn + V + v, na, gy);
  if (DIM == 3) load_str<VEC>(pn + 2 * V + v, na, gz);
  // final mode: phi0 is requested with the own values (in flight while the taps are built), not after the gathers
  float p0[DIM][VEC];
  if (final_mode == 1) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) load_str<VEC>(phi0 + ((int64_t)n * DIM + c) * V + v, na, p0[c]);
  }
  Taps<DIM, PAD_BORDER> t[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) t[k].build(gx[k], gy[k], DIM == 3 ? gz[k] : 0.f, d);
  float* on = out + (int64_t)n * DIM * V + v;
  // all gathers first (no control flow between the channels: their loads stay in flight together)
  float r[DIM][VEC];
#pragma unroll
  for (int c = 0; c < DIM; ++c)
#pragma unroll
    for (int k = 0; k < VEC; ++k) r[c][k] = sample_linear<DIM, PAD_BORDER>(pn + (int64_t)c * V, t[k], d);
  if (final_mode == 1) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const int64_t vv = v + (int64_t)k * kBlock;
        int idx;
        int S;
        if (c == 0) { idx = (int)(vv % d.s2); S = d.s2; }
        else if (c == 1) { idx = (int)((vv / d.s2) % d.s1); S = d.s1; }
        else { idx = (int)(vv / ((int64_t)d.s2 * d.s1)); S = d.s0; }
        r[c][k] = (r[c][k] - p0[c][k]) + lin_coord(idx, S);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) store_str<VEC>(on + (int64_t)c * V, na, r[c]);
  if (disp_out) {
    // voxel coordinates of the VEC strided outputs: one 32-bit division for the first, carries for the rest
    unsigned vx, vy, vz;
    if (((d.s2 & (d.s2 - 1)) | (d.s1 & (d.s1 - 1))) == 0) {   // powers of two (uniform branch): shifts and masks
      const int l2 = 31 - __builtin_clz((unsigned)d.s2), l1 = 31 - __builtin_clz((unsigned)d.s1);
      vx = (unsigned)v & (unsigned)(d.s2 - 1);
      vy = ((unsigned)v >> l2) & (unsigned)(d.s1 - 1);
      vz = (unsigned)v >> (l2 + l1);
    } else {
      const unsigned vr = (unsigned)v / (unsigned)d.s2;
      vx = (unsigned)v - vr * (unsigned)d.s2;
      vz = vr / (unsigned)d.s1;
      vy = vr - vz * (unsigned)d.s1;
    }
    float dmax = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (k < na) {
        dmax = fmaxf(dmax, fmaxf(voxel_displacement(r[0][k], d.s2, (int)vx), voxel_displacement(r[1][k], d.s1, (int)vy)));
        if (DIM == 3) dmax = fmaxf(dmax, voxel_displacement(r[DIM - 1][k], d.s0, (int)vz));
      }
      vx += kBlock;
      while (vx >= (unsigned)d.s2) { vx -= (unsigned)d.s2; if (++vy >= (unsigned)d.s1) { vy = 0; ++vz; } }
    }
    wave_max_to_slots(dmax, disp_out);
  }
}

template <int DIM, int VEC>
__global__ void __launch_bounds__(kBlock)
k_compose_self_fwd(const float* __restrict__ phi, float* __restrict__ out, const float* __restrict__ phi0,
                   Dims d, int final_mode, float* __restrict__ disp_out) {
  compose_self_fwd_body<DIM, VEC>(phi, out, phi0, d, final_mode, disp_out, blockIdx.y,
                                  (int64_t)blockIdx.x * (kBlock * VEC) + threadIdx.x);
}

// The REPEAT behind the fused kernel of expo_fused2d.hip.  That kernel has normally produced phi_1..phi_k already; *flag holds
// the largest number of trailing levels any of its workgroups could NOT do (0 = none: this launch returns at once -- ONE
// launch whatever k, because the host side of a small problem (cfg-1) pays for every launch it enqueues).  Otherwise the
// missing levels k - deficit + 1 .. k are run here the ordinary way (compose_self_fwd_body<2, 2>: the same bits, also where
// the fused kernel had got that far) on a small persistent grid, with a grid-wide barrier between two levels: every
// workgroup of the grid is resident (the host sizes the grid per device from the occupancy of this kernel: at most half of
// what the device holds at once, two per CU on MI355X where eight fit), an arrival counter in device memory, and
// agent-scope fences either side so that what another XCD's L2 still holds is written back / read again.
// `barrier`: one zero-initialised 32-bit counter per call (the word behind the flag).
__global__ void __launch_bounds__(kBlock)
k_expo_repeat2d(const float* __restrict__ phi0, float* __restrict__ fields, int64_t F, Dims d, int k, int N, int nbx,
                float* __restrict__ disp_rows, const float* __restrict__ flag, unsigned int* __restrict__ barrier) {
  const int deficit = (int)*flag;
  if (deficit <= 0) return;
  const int items = N * nbx;
  int passed = 0;
  for (int lev = max(k - deficit + 1, 1); lev <= k; ++lev) {
    const float* src = lev == 1 ? phi0 : fields + (int64_t)(lev - 2) * F;
    float* dst = fields + (int64_t)(lev - 1) * F;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
      const int n = it / nbx, b = it - n * nbx;
      compose_self_fwd_body<2, 2>(src, dst, nullptr, d, 0, disp_rows ? disp_rows + (int64_t)lev * kDispSlots : nullptr, n,
                                  (int64_t)b * (kBlock * 2) + threadIdx.x);
    }
    if (lev == k) break;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                   // this workgroup's stores are visible device-wide
      atomicAdd(barrier, 1u);
      const unsigned int target = (unsigned int)(++passed) * gridDim.x;
      while (atomicAdd(barrier, 0u) < target) __builtin_amdgcn_s_sleep(4);
      __threadfence();                                   // and nothing stale is read behind the barrier
    }
    __syncthreads();
  }
}

// gphi must be zero-initialised by the caller (scatter target).
template <int DIM, int VEC>
__global__ void __launch_bounds__(kBlock)
k_compose_self_bwd(const float* __restrict__ gout, const float* __restrict__ phi, float* __restrict__ gphi, Dims d) {
  const int64_t V = d.voxels();
  const int n = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * (kBlock * VEC) + threadIdx.x;
  const int na = active_count(v, V, VEC);
  if (na == 0) return;
  const float* pn = phi + (int64_t)n * DIM * V;
  float* gpn = gphi + (int64_t)n * DIM * V;
  const float* gon = gout + (int64_t)n * DIM * V + v;
  float gx[VEC], gy[VEC], gz[VEC];
  load_str<VEC>(pn + v, na, gx);
  load_str<VEC>(pn + V + v, na, gy);
  if (DIM == 3) load_str<VEC>(pn + 2 * V + v, na, gz);
  Taps<DIM, PAD_BORDER> t[VEC];
  float ax[VEC], ay[VEC], az[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    t[k].build(gx[k], gy[k], DIM == 3 ? gz[k] : 0.f, d);
    ax[k] = ay[k] = az[k] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    float go[VEC];
    load_str<VEC>(gon + (int64_t)c * V, na, go);
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      sample_linear_bwd<DIM, PAD_BORDER, true, true>(pn + (int64_t)c * V, gpn + (int64_t)c * V, go[k], t[k], d, ax[k],
                                                     ay[k], az[k]);
  }
  // the coordinate path lands on the same tensor (input == grid)
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    if (k >= na) continue;
    const int64_t vk = v + (int64_t)k * kBlock;
    if (t[k].x.mult != 0.f) atomic_add_f32(gpn + vk, t[k].x.mult * ax[k]);
    if (t[k].y.mult != 0.f) atomic_add_f32(gpn + V + vk, t[k].y.mult * ay[k]);
    if (DIM == 3 && t[k].z.mult != 0.f) atomic_add_f32(gpn + 2 * V + vk, t[k].z.mult * az[k]);
  }
}

// =============================================================================================
// affine warp: grid = theta_n * (x, y[, z], 1) evaluated in registers
// =============================================================================================
template <int DIM>
__device__ __forceinline__ void affine_position(const Theta<DIM>& th, int64_t v, const Dims& d, float& bx, float& by,
                                                float& bz, float& gx, float& gy, float& gz) {
  const int ix = (int)(v % d.s2);
  const int iy = (int)((v / d.s2) % d.s1);
  const int iz = DIM == 3 ? (int)(v / ((int64_t)d.s2 * d.s1)) : 0;
  affine_position_xyz<DIM>(th, ix, iy, iz, d, bx, by, bz, gx, gy, gz);
}

// Which voxel a thread of the affine kernels handles.  A wave normally covers a 64-voxel line along x; under rotation
// that line crosses ~64 |dy/dx| rows of the input and every gather instruction touches up to 64 cache lines (measured
// at 20 degrees: 3D C=4 forward 103 -> 252 us).  When the line would cross more than 7 rows the wave covers an 8 x 8
// (x, y) patch instead (~11 row segments; 252 -> 162 us); for near-axis-aligned maps the line stays (the patch costs
// ~15 % there).  Decided per sample from theta, no host involvement.  Launch with affine_grid_blocks().
constexpr int kPatch = 8;
__host__ __device__ inline int64_t affine_waves(const Dims& d) {
  const int64_t line = (d.voxels() + 63) / 64;
  const int64_t patch = (int64_t)((d.s2 + kPatch - 1) / kPatch) * ((d.s1 + kPatch - 1) / kPatch) * d.s0;
  return line > patch ? line : patch;
}
static inline int affine_grid_blocks(const Dims& d) { return (int)((affine_waves(d) + kBlock / 64 - 1) / (kBlock / 64)); }

// `slope` = rows (and slices) of the gathered tensor crossed per voxel step along x of the iterated one
__device__ __forceinline__ int64_t mapped_thread_voxel(float slope, const Dims& d, float thr = 7.f) {
  const bool patch = 64.f * slope > thr && d.s1 >= kPatch && d.s2 >= kPatch;
  const int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (!patch) {
    const int64_t v = w * 64 + lane;
    return v < d.voxels() ? v : -1;
  }
  const int nx = (d.s2 + kPatch - 1) / kPatch, ny = (d.s1 + kPatch - 1) / kPatch;
  const int px = (int)(w % nx), py = (int)((w / nx) % ny);
  const int64_t pz = w / ((int64_t)nx * ny);
  const int ix = px * kPatch + (lane & (kPatch - 1)), iy = py * kPatch + lane / kPatch;
  if (ix >= d.s2 || iy >= d.s1 || pz >= d.s0) return -1;
  return (pz * d.s1 + iy) * d.s2 + ix;
}

template <int DIM>
__device__ __forceinline__ int64_t affine_thread_voxel(const Theta<DIM>& th, const Dims& d, float thr = 7.f) {
  const float den = (float)max(d.s2 - 1, 1);
  float rows = fabsf(th.m[1][0]) * (float)(d.s1 - 1) / den;
  if (DIM == 3) rows += fabsf(th.m[DIM - 1][0]) * (float)(d.s0 - 1) / den;
  return mapped_thread_voxel(rows, d, thr);
}

template <int DIM, int INTERP, int PAD>
__global__ void __launch_bounds__(kBlock)
k_affine_warp_fwd(const float* __restrict__ in, const float* __restrict__ theta, float* __restrict__ out, int C,
                  Dims d, float thr) {
  const int64_t V = d.voxels();
  const int n = blockIdx.y;
  Theta<DIM> th;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) th.m[r][c] = theta[(int64_t)n * DIM * (DIM + 1) + r * (DIM + 1) + c];
  const int64_t v = affine_thread_voxel<DIM>(th, d, thr);
  if (v < 0) return;
  float bx, by, bz, gx, gy, gz;
  affine_position<DIM>(th, v, d, bx, by, bz, gx, gy, gz);
  const float* inn = in + (int64_t)n * C * V;
  float* on = out + (int64_t)n * C * V + v;
  if (INTERP == INTERP_LINEAR) {
    Taps<DIM, PAD> t;
    t.build(gx, gy, gz, d);
    for (int c = 0; c < C; ++c) on[(int64_t)c * V] = sample_linear<DIM, PAD>(inn + (int64_t)c * V, t, d);
  } else {
    bool vx, vy, vz = true;
    const int ix = nearest_index<PAD>(gx, d.s2, vx);
    const int iy = nearest_index<PAD>(gy, d.s1, vy);
    const int iz = DIM == 3 ? nearest_index<PAD>(gz, d.s0, vz) : 0;
    const bool ok = vx && vy && vz;
    const int off = (iz * d.s1 + iy) * d.s2 + ix;
    for (int c = 0; c < C; ++c) on[(int64_t)c * V] = ok ? inn[(int64_t)c * V + off] : 0.f;
  }
}

// The same map with the voxel coordinates of wave `w` returned directly (32-bit arithmetic: the 64-bit `%` and `/` of
// affine_position() are ~100 instructions each).
__device__ __forceinline__ bool mapped_wave_coords(float slope, const Dims& d, unsigned w, int lane, float thr, int& ix, int& iy,
                                                   int& iz) {
  const bool patch = 64.f * slope > thr && d.s1 >= kPatch && d.s2 >= kPatch;
  if (!patch) {
    const unsigned v = w * 64u + (unsigned)lane;
    const unsigned r = v / (unsigned)d.s2;
    ix = (int)(v - r * (unsigned)d.s2);
    iz = (int)(r / (unsigned)d.s1);
    iy = (int)(r - (unsigned)iz * (unsigned)d.s1);
    return v < (unsigned)d.voxels();
  }
  const unsigned nx = (unsigned)(d.s2 + kPatch - 1) / kPatch, ny = (unsigned)(d.s1 + kPatch - 1) / kPatch;
  const unsigned r = w / nx;
  iz = (int)(r / ny);
  ix = (int)(w - r * nx) * kPatch + (lane & (kPatch - 1));
  iy = (int)(r - (unsigned)iz * ny) * kPatch + lane / kPatch;
  return ix < d.s2 && iy < d.s1 && iz < d.s0;
}

// Linear forward with CT channels known at compile time and VEC voxels per thread (consecutive waves of the mapping
// above): every corner load of every channel and voxel of a thread is issued before the first is consumed.  The
// one-voxel kernel above with its run-time channel loop made 1 (C = 1) to 4 serial round trips of 2^d loads and spent
// 61% of its wave-cycles waiting (profiles/r02/sq_issue_summary.txt).
template <int DIM, int PAD, int CT, int VEC>
__global__ void __launch_bounds__(kBlock)
k_affine_warp_fwd_v(const float* __restrict__ in, const float* __restrict__ theta, float* __restrict__ out, Dims d, float thr,
                    unsigned nwaves) {
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  Theta<DIM> th;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) th.m[r][c] = theta[(int64_t)n * DIM * (DIM + 1) + r * (DIM + 1) + c];
  const float den = (float)max(d.s2 - 1, 1);
  float slope = fabsf(th.m[1][0]) * (float)(d.s1 - 1) / den;
  if (DIM == 3) slope += fabsf(th.m[DIM - 1][0]) * (float)(d.s0 - 1) / den;
  const unsigned w0 = ((unsigned)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * VEC;
  const int lane = threadIdx.x & 63;
  const float* inn = in + (int64_t)n * CT * V;
  float* on = out + (int64_t)n * CT * V;
  Taps<DIM, PAD> t[VEC];
  int off[VEC];
  bool ok[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    int ix, iy, iz;
    ok[k] = w0 + k < nwaves && mapped_wave_coords(slope, d, w0 + k, lane, thr, ix, iy, iz);
    if (!ok[k]) { ix = iy = iz = 0; }
    off[k] = (iz * d.s1 + iy) * d.s2 + ix;
    float bx, by, bz, gx, gy, gz;
    affine_position_xyz<DIM>(th, ix, iy, iz, d, bx, by, bz, gx, gy, gz);
    t[k].build(gx, gy, gz, d);
  }
  float r[VEC][CT];
#pragma unroll
  for (int k = 0; k < VEC; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c) r[k][c] = sample_linear<DIM, PAD>(inn + (int64_t)c * V, t[k], d);
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    if (!ok[k]) continue;
#pragma unroll
    for (int c = 0; c < CT; ++c) on[(int64_t)c * V + off[k]] = r[k][c];
  }
}

// gtheta_partial: (N, gridDim.x, DIM*(DIM+1)) block partial sums, reduced by k_reduce_partials.
template <int DIM, int INTERP, int PAD, bool NEED_GIN, bool NEED_GTHETA>
__global__ void __launch_bounds__(kBlock)
k_affine_warp_bwd(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ theta,
                  float* __restrict__ gin, float* __restrict__ gtheta_partial, int C, Dims d,
                  const int* __restrict__ mode, const float* __restrict__ red_partial, float* __restrict__ red_out, int red_nb) {
  constexpr int NT = DIM * (DIM + 1);
  __shared__ float smem[4 * NT];
  const int64_t V = d.voxels();
  const int n = blockIdx.y;
  // red_out != null (round 5): this launch also is the second stage of the theta gradient the box kernel left as block
  // partials -- block k < NT of sample n does what block (n, k) of k_reduce_partials does, the same sums in the same order.
  // (When the box kernels took every sample this launch has nothing else to do: one launch of ~5 us instead of two.)
  if (red_out && (int)blockIdx.x < NT) {
    float s[1] = {0.f};
    for (int b = threadIdx.x; b < red_nb; b += blockDim.x) s[0] += red_partial[((int64_t)n * red_nb + b) * NT + blockIdx.x];
    block_sum<1>(s, smem);
    if (threadIdx.x == 0) red_out[(int64_t)n * NT + blockIdx.x] = s[0];
    __syncthreads();
  }
  // mode != null: grad_in of samples with mode[n] == 0 is produced by k_affine_gather_bwd; scatter only the rest
  const bool do_gin = NEED_GIN && (mode == nullptr || mode[n] != 0);
  if (!NEED_GTHETA && !do_gin) return;
  Theta<DIM> th;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) th.m[r][c] = theta[(int64_t)n * NT + r * (DIM + 1) + c];
  const int64_t v = affine_thread_voxel<DIM>(th, d);
  float acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) acc[k] = 0.f;
  if (v >= 0) {
    float bx, by, bz, gx, gy, gz;
    affine_position<DIM>(th, v, d, bx, by, bz, gx, gy, gz);
    const float* inn = in + (int64_t)n * C * V;
    float* ginn = NEED_GIN ? gin + (int64_t)n * C * V : nullptr;
    const float* gon = gout + (int64_t)n * C * V + v;
    if (INTERP == INTERP_LINEAR) {
      Taps<DIM, PAD> t;
      t.build(gx, gy, gz, d);
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int c = 0; c < C; ++c) {
        if (do_gin) sample_linear_bwd<DIM, PAD, NEED_GIN, NEED_GTHETA>(inn + (int64_t)c * V, NEED_GIN ? ginn + (int64_t)c * V : nullptr,
                                                                       gon[(int64_t)c * V], t, d, ax, ay, az);
        else sample_linear_bwd<DIM, PAD, false, NEED_GTHETA>(inn + (int64_t)c * V, nullptr, gon[(int64_t)c * V], t, d, ax, ay, az);
      }
      if (NEED_GTHETA) {
        const float ggx = t.x.mult * ax, ggy = t.y.mult * ay, ggz = DIM == 3 ? t.z.mult * az : 0.f;
        const float base[4] = {bx, by, DIM == 3 ? bz : 1.f, 1.f};
#pragma unroll
        for (int c = 0; c < DIM + 1; ++c) {
          acc[0 * (DIM + 1) + c] = ggx * base[c];
          acc[1 * (DIM + 1) + c] = ggy * base[c];
          if constexpr (DIM == 3) acc[2 * (DIM + 1) + c] = ggz * base[c];
        }
      }
    } else if (do_gin) {
      bool vx, vy, vz = true;
      const int ix = nearest_index<PAD>(gx, d.s2, vx);
      const int iy = nearest_index<PAD>(gy, d.s1, vy);
      const int iz = DIM == 3 ? nearest_index<PAD>(gz, d.s0, vz) : 0;
      if (vx && vy && vz) {
        const int off = (iz * d.s1 + iy) * d.s2 + ix;
        for (int c = 0; c < C; ++c) atomic_add_f32(ginn + (int64_t)c * V + off, gon[(int64_t)c * V]);
      }
    }
  }
  if (NEED_GTHETA) {
    block_sum<NT>(acc, smem);
    if (threadIdx.x == 0) {
      float* dst = gtheta_partial + ((int64_t)n * gridDim.x + blockIdx.x) * NT;
#pragma unroll
      for (int k = 0; k < NT; ++k) dst[k] = acc[k];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Affine warp, grad_in as a GATHER (zeros padding, linear).  The sampling positions x(v) = M v + t form a
// lattice, so the samples that deposit on voxel u are the lattice points inside M^-1((u-1,u+1)^d - t): a
// handful of candidates found with slab tests; each candidate's taps are recomputed with exactly the forward's
// arithmetic, so the weights are bit-identical to the scatter formulation -- without a single atomic.
//   geo[n] = { M (3x3, xyz order), t (3), Minv (3x3), ext (3) }, mode[n] = 0 gather | 1 fall back to atomics
// ---------------------------------------------------------------------------------------------
template <int DIM>
__global__ void k_affine_geometry(const float* __restrict__ theta, float* __restrict__ geo, int* __restrict__ mode,
                                  int N, Dims d) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  affine_geometry_one<DIM>(theta, geo, mode, n, d);
}

constexpr int kGatherCand = 12;   // listed candidates per voxel (LDS column); the rest are accumulated where found

template <int DIM, int CMAX>
__global__ void __launch_bounds__(kBlock)
k_affine_gather_bwd(const float* __restrict__ gout, const float* __restrict__ theta, const float* __restrict__ geo,
                    const int* __restrict__ mode, float* __restrict__ gin, int C, Dims d) {
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const float* gn0 = geo + (int64_t)n * kGeoFloats;
  // the candidates of consecutive u along x walk along Minv[:, 0] in the gradient tensor: patch mapping when that is steep
  // (32-bit coordinates: the 64-bit divisions of mapped_thread_voxel() were ~350 of this kernel's ~1600 instructions)
  int ux, uy, uz;
  if (!mapped_wave_coords(fabsf(gn0[12 + 3]) + (DIM == 3 ? fabsf(gn0[12 + 6]) : 0.f), d,
                          (unsigned)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), threadIdx.x & 63, 7.f, ux, uy, uz)) return;
  const int u = (uz * d.s1 + uy) * d.s2 + ux;
  float* ginn = gin + (int64_t)n * C * V + u;
  if (mode[n] != 0) {  // this sample goes through the atomic kernel: start from zero
    for (int c = 0; c < C; ++c) ginn[(int64_t)c * V] = 0.f;
    return;
  }
  Theta<DIM> th;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) th.m[r][c] = theta[(int64_t)n * DIM * (DIM + 1) + r * (DIM + 1) + c];
  const float* gn = geo + (int64_t)n * kGeoFloats;
  float M[3][3], Mi[3][3], t[3], ext[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { M[r][q] = gn[r * 3 + q]; Mi[r][q] = gn[12 + r * 3 + q]; }
    t[r] = gn[9 + r];
    ext[r] = gn[21 + r];
  }
  const float uu[3] = {(float)ux - t[0], (float)uy - t[1], DIM == 3 ? (float)uz - t[2] : 0.f};
  float cen[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) cen[q] = Mi[q][0] * uu[0] + Mi[q][1] * uu[1] + Mi[q][2] * uu[2];
  const int zlo = DIM == 3 ? max(0, (int)ceilf(cen[2] - ext[2])) : 0;
  const int zhi = DIM == 3 ? min(d.s0 - 1, (int)floorf(cen[2] + ext[2])) : 0;
  const int ylo = max(0, (int)ceilf(cen[1] - ext[1])), yhi = min(d.s1 - 1, (int)floorf(cen[1] + ext[1]));
  const int xlo0 = max(0, (int)ceilf(cen[0] - ext[0])), xhi0 = min(d.s2 - 1, (int)floorf(cen[0] + ext[0]));
  float acc[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
  const float* gon = gout + (int64_t)n * C * V;
  // Two passes: the candidates (sample index, weight) are first listed in a per-thread LDS column, then their grad_out
  // values are loaded four at a time, unconditionally.  Loading each candidate where it is found made the loop
  // load -> wait -> accumulate: 8-16 serial memory round trips per voxel.
  __shared__ int2 cand[CMAX > 1 ? kGatherCand : 1][kBlock];
  float minv[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) minv[r] = fabsf(M[r][0]) > 1e-3f ? 1.f / M[r][0] : 0.f;   // six divisions per candidate row otherwise
  int nc = 0;
  for (int vz = zlo; vz <= zhi; ++vz)
    for (int vy = ylo; vy <= yhi; ++vy) {
      // slab tests along x: |M[r][0] vx + k_r| < 1 for every output axis r
      float lo = (float)xlo0, hi = (float)xhi0;
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        const float k = M[r][1] * (float)vy + M[r][2] * (float)vz - uu[r];
        const float m = M[r][0];
        if (fabsf(m) > 1e-3f) {
          const float p = (-1.f - k) * minv[r], q = (1.f - k) * minv[r];   // (bounds carry 1e-3 of slack)
          lo = fmaxf(lo, fminf(p, q) - 1e-3f);
          hi = fminf(hi, fmaxf(p, q) + 1e-3f);
        }
      }
      const int xl = (int)ceilf(lo), xh = (int)floorf(hi);
      for (int vx = xl; vx <= xh; ++vx) {
        // the sample's position with the forward's arithmetic; its weight on voxel u in tent form
        // max(0, 1 - |x - u|) (equal to the corner weight (i0 + 1) - x or x - i0 whichever cell x is in, so no floor,
        // no validity flags: a third of the instructions of a full tap build)
        float bx, by, bz, gx, gy, gz;
        affine_position_xyz<DIM>(th, vx, vy, vz, d, bx, by, bz, gx, gy, gz);
        const float px = ((gx + 1.f) * 0.5f) * (float)(d.s2 - 1), py = ((gy + 1.f) * 0.5f) * (float)(d.s1 - 1);
        float w = fmaxf(0.f, 1.f - fabsf(px - (float)ux)) * fmaxf(0.f, 1.f - fabsf(py - (float)uy));
        if (DIM == 3) w *= fmaxf(0.f, 1.f - fabsf(((gz + 1.f) * 0.5f) * (float)(d.s0 - 1) - (float)uz));
        if (!(w > 0.f)) continue;
        const int v = (vz * d.s1 + vy) * d.s2 + vx;
        if (CMAX > 1 && nc < kGatherCand) {   // one channel: a single load per candidate, listing costs more than it hides
          cand[nc][threadIdx.x] = make_int2(v, __float_as_int(w));
          ++nc;
        } else {      // more candidates than the column holds (strong minification): the old way
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (c < C) acc[c] += w * gon[(int64_t)c * V + v];
        }
      }
    }
  for (int i0 = 0; i0 < nc; i0 += 4) {
    int vi[4];
    float wi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int2 e = cand[min(i0 + k, nc - 1)][threadIdx.x];
      vi[k] = e.x;
      wi[k] = i0 + k < nc ? __int_as_float(e.y) : 0.f;
    }
    float val[4][CMAX];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) val[k][c] = gon[(int64_t)(c < C ? c : 0) * V + vi[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) acc[c] += wi[k] > 0.f ? wi[k] * val[k][c] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) ginn[(int64_t)c * V] = acc[c];
}

// out[row][k] = sum_b partial[row][b][k]      (deterministic second stage)
__global__ void k_reduce_partials(const float* __restrict__ partial, float* __restrict__ out, int nb, int K) {
  const int row = blockIdx.x;
  const int k = blockIdx.y;
  __shared__ float smem[4];
  float s[1] = {0.f};
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s[0] += partial[((int64_t)row * nb + b) * K + k];
  block_sum<1>(s, smem);
  if (threadIdx.x == 0) out[(int64_t)row * K + k] = s[0];
}

}  // namespace advchain

using namespace advchain;

// scatter_tiled.hip
int advchain_scatter_tiled_launch(bool self, const float* gout, const float* in, const float* grid, float* gin,
                                  float* ggrid, int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                  int32_t* workspace, int chain, int halo, hipStream_t st);

// adjoint_gather.hip
int advchain_self_adjoint_gather_launch(const float* gout, const float* phi, float* gphi, int64_t N, int ndim, Dims d,
                                        int32_t* workspace, int chain, int halo, hipStream_t st);

int advchain_warp_adjoint_gather_launch(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                        int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                        int32_t* workspace, int halo, hipStream_t st);

// scatter_march.hip: owner-computes z-march for exact bounds of 2..4 voxels (3D)
int advchain_scatter_march_launch(bool self, const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                  int64_t N, int64_t C, Dims d, int padding, int clamp_grid, int H, int32_t* workspace,
                                  hipStream_t st);

// scatter_window.hip
int advchain_scatter_window_launch(bool self, const float* gout, const float* in, const float* grid, float* gin,
                                   float* ggrid, int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                   int halo, int32_t* workspace, hipStream_t st, int32_t* det_ws);
int advchain_scatter_rows2d_launch(bool self, const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                   int64_t N, int64_t C, Dims d, int padding, int clamp_grid, int H, int32_t* workspace,
                                   hipStream_t st, int rm_flags = 0);
bool advchain_scatter_rows2d_takes(bool self, int64_t C, Dims d, int padding, int H, int64_t N);

// affine_box.hip: LDS-staged source box (linear, zeros padding, rows of 4k voxels)
bool advchain_affine_box_fwd_launch(const float* in, const float* theta, float* out, int64_t N, int64_t C, int ndim, Dims d,
                                    hipStream_t st, const float* ride_in = nullptr, float* ride_out = nullptr,
                                    int ride_nonzero = 0);
int advchain_affine_box_tiles(int ndim, Dims d);
int advchain_affine_box_gtheta_launch(const float* gout, const float* in, const float* theta, float* gpart, int64_t N,
                                      int64_t C, int ndim, Dims d, int max_blocks, hipStream_t st, float* tilemax, float* geo = nullptr, int* mode = nullptr);
bool advchain_affine_box_gin_launch(const float* gout, const float* theta, const float* geo, const int* mode, float* gin,
                                    int64_t N, int64_t C, int ndim, Dims d, hipStream_t st, const float* tilemax);

// expo_fused2d.hip / adjoint_fused2d.hip: the sub-pixel squarings of a 2D chain in one launch (forward / backward)
int advchain_expo_fused_fwd2d_launch(const float* phi0, float* fields, int64_t N, advchain::Dims d, int k, int halos,
                                     float* disp_rows, float* fail_flag, hipStream_t stream, bool query = false);
int advchain_adjoint_fused2d_launch(const float* gk, const float* phi0, const float* fields, float* g0, int64_t N, advchain::Dims d,
                                    int k, int32_t* workspace, hipStream_t st);
// gather_tiled.hip
int advchain_sample_tiled_launch(bool self, const float* in, const float* grid, float* out, const float* phi0,
                                 int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid, int final_mode,
                                 int halo, float* disp_out, hipStream_t st, int disp_hint);

// ---------------------------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------------------------
// UNR = 4 independent voxels per thread once the volume fills the chip
// Measured on MI355X (tools/kernel_bench.py): 2D gathers gain 1.2x from 4 chains per thread; in 3D the 4 chains
// sit 4 rows apart in 2 z-planes x C channels and thrash the 32 KiB L1, so one voxel per thread wins (1.3x).
static inline bool use_unroll(int64_t voxels, int ndim) {
  static const bool force1 = false;
  static const bool force4 = false;
  if (force4) return true;
  return !force1 && ndim == 2 && voxels >= 4 * kBlock;
}

#define DISPATCH_PAD(PADV, ...)                                            \
  switch (PADV) {                                                          \
    case PAD_ZEROS: { constexpr int PAD = PAD_ZEROS; __VA_ARGS__; } break; \
    case PAD_BORDER: { constexpr int PAD = PAD_BORDER; __VA_ARGS__; } break; \
    default: { constexpr int PAD = PAD_REFLECTION; __VA_ARGS__; } break;   \
  }

template <int DIM>
static int launch_grid_sample_fwd(const float* in, const float* grid, float* out, int64_t N, int64_t C, Dims id, Dims od,
                                  int interp, int padding, int clamp_grid, hipStream_t st, const float* ride_in = nullptr,
                                  float* ride_out = nullptr, int ride_nonzero = 0) {
  const int64_t OV = od.voxels();
  const bool vec4 = use_unroll(OV, DIM);
  const int vec = vec4 ? 4 : 1;
  dim3 g(advchain_blocks(OV, kBlock * vec), (unsigned)N), b(kBlock);
  DISPATCH_PAD(padding, {
    if (interp == INTERP_LINEAR) {
      if (vec4) hipLaunchKernelGGL((k_grid_sample_fwd<DIM, INTERP_LINEAR, PAD, 4>), g, b, 0, st, in, grid, out, (int)C, id, od, clamp_grid, ride_in, ride_out, ride_nonzero);
      else hipLaunchKernelGGL((k_grid_sample_fwd<DIM, INTERP_LINEAR, PAD, 1>), g, b, 0, st, in, grid, out, (int)C, id, od, clamp_grid, ride_in, ride_out, ride_nonzero);
    } else {
      if (vec4) hipLaunchKernelGGL((k_grid_sample_fwd<DIM, INTERP_NEAREST, PAD, 4>), g, b, 0, st, in, grid, out, (int)C, id, od, clamp_grid, ride_in, ride_out, ride_nonzero);
      else hipLaunchKernelGGL((k_grid_sample_fwd<DIM, INTERP_NEAREST, PAD, 1>), g, b, 0, st, in, grid, out, (int)C, id, od, clamp_grid, ride_in, ride_out, ride_nonzero);
    }
  });
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

template <int DIM, int INTERP, int PAD, int VEC>
static void launch_gs_bwd_flags(dim3 g, dim3 b, hipStream_t st, const float* gout, const float* in, const float* grid,
                                float* gin, float* ggrid, int C, Dims id, Dims od, int clamp_grid) {
  if (gin && ggrid) hipLaunchKernelGGL((k_grid_sample_bwd<DIM, INTERP, PAD, VEC, true, true>), g, b, 0, st, gout, in, grid, gin, ggrid, C, id, od, clamp_grid);
  else if (gin) hipLaunchKernelGGL((k_grid_sample_bwd<DIM, INTERP, PAD, VEC, true, false>), g, b, 0, st, gout, in, grid, gin, ggrid, C, id, od, clamp_grid);
  else hipLaunchKernelGGL((k_grid_sample_bwd<DIM, INTERP, PAD, VEC, false, true>), g, b, 0, st, gout, in, grid, gin, ggrid, C, id, od, clamp_grid);
}

template <int DIM>
static int launch_grid_sample_bwd(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                  int64_t N, int64_t C, Dims id, Dims od, int interp, int padding, int clamp_grid,
                                  hipStream_t st) {
  const int64_t OV = od.voxels();
  const bool vec4 = use_unroll(OV, DIM);
  const int vec = vec4 ? 4 : 1;
  dim3 g(advchain_blocks(OV, kBlock * vec), (unsigned)N), b(kBlock);
  DISPATCH_PAD(padding, {
    if (interp == INTERP_LINEAR) {
      if (vec4) launch_gs_bwd_flags<DIM, INTERP_LINEAR, PAD, 4>(g, b, st, gout, in, grid, gin, ggrid, (int)C, id, od, clamp_grid);
      else launch_gs_bwd_flags<DIM, INTERP_LINEAR, PAD, 1>(g, b, st, gout, in, grid, gin, ggrid, (int)C, id, od, clamp_grid);
    } else {
      if (vec4) launch_gs_bwd_flags<DIM, INTERP_NEAREST, PAD, 4>(g, b, st, gout, in, grid, gin, ggrid, (int)C, id, od, clamp_grid);
      else launch_gs_bwd_flags<DIM, INTERP_NEAREST, PAD, 1>(g, b, st, gout, in, grid, gin, ggrid, (int)C, id, od, clamp_grid);
    }
  });
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// min_voxels = 2 for a tensor that is GATHERED from (the paired corner gathers read two neighbouring elements:
// sampler_common.h); the output side of a resampling warp may be a single voxel
static inline bool dims_ok(int ndim, const int64_t* s, int64_t min_voxels = 2) {
  if (ndim != 2 && ndim != 3) return false;
  int64_t v = 1;
  for (int i = 0; i < ndim; ++i) {
    if (s[i] < 1 || s[i] > (1 << 24)) return false;
    v *= s[i];
  }
  return v >= min_voxels;
}
static inline Dims make_dims(int ndim, const int64_t* s) {
  Dims d;
  if (ndim == 3) { d.s0 = (int)s[0]; d.s1 = (int)s[1]; d.s2 = (int)s[2]; }
  else { d.s0 = 1; d.s1 = (int)s[0]; d.s2 = (int)s[1]; }
  return d;
}

// =============================================================================================
// C ABI
// =============================================================================================
template <int DIM, int INTERP, int PAD>
static void launch_affine_bwd(dim3 g, dim3 b, hipStream_t st, const float* gout, const float* in, const float* theta,
                              float* gin, float* gpart, int C, Dims d, const int* mode, const float* red_partial = nullptr,
                              float* red_out = nullptr, int red_nb = 0) {
  if (gin && gpart) hipLaunchKernelGGL((k_affine_warp_bwd<DIM, INTERP, PAD, true, true>), g, b, 0, st, gout, in, theta, gin, gpart, C, d, mode, nullptr, nullptr, 0);
  else if (gin) hipLaunchKernelGGL((k_affine_warp_bwd<DIM, INTERP, PAD, true, false>), g, b, 0, st, gout, in, theta, gin, gpart, C, d, mode, red_partial, red_out, red_nb);
  else hipLaunchKernelGGL((k_affine_warp_bwd<DIM, INTERP, PAD, false, true>), g, b, 0, st, gout, in, theta, gin, gpart, C, d, mode, nullptr, nullptr, 0);
}

extern "C" {

int advchain_nonzero_mask(const float* x, float* out, int64_t n, void* stream);   // fields.hip

int advchain_grid_sample_fwd(const float* in, const float* grid, float* out, int64_t N, int64_t C, int ndim,
                             const int64_t* in_dims, const int64_t* out_dims, int interp, int padding, int clamp_grid,
                             void* stream) {
  const int disp_hint = (clamp_grid >> 8) & 0xff;   // bits 8..15: displacement estimate in voxels (0 = unknown), a performance hint
  clamp_grid &= 1;
  ADVCHAIN_CHECK_ARG(in && grid && out, "grid_sample_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, in_dims) && dims_ok(ndim, out_dims, 1), "grid_sample_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "grid_sample_fwd: bad N/C");
  ADVCHAIN_CHECK_ARG(interp == INTERP_LINEAR || interp == INTERP_NEAREST, "grid_sample_fwd: interp must be 0 (linear) or 1 (nearest)");
  ADVCHAIN_CHECK_ARG(padding >= 0 && padding <= 2, "grid_sample_fwd: padding must be 0/1/2");
  if (N == 0) return ADVCHAIN_OK;
  const Dims id = make_dims(ndim, in_dims), od = make_dims(ndim, out_dims);
  ADVCHAIN_CHECK_ARG(id.voxels() < (1ll << 31) && od.voxels() < (1ll << 31), "grid_sample_fwd: per-sample volume too large");
  if (interp == INTERP_LINEAR && id.s0 == od.s0 && id.s1 == od.s1 && id.s2 == od.s2) {   // LDS-staged tiles
    const int rc = advchain_sample_tiled_launch(false, in, grid, out, nullptr, N, C, ndim, id, padding, clamp_grid, 0, 0, nullptr,
                                                (hipStream_t)stream, disp_hint);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
  return ndim == 3 ? launch_grid_sample_fwd<3>(in, grid, out, N, C, id, od, interp, padding, clamp_grid, (hipStream_t)stream)
                   : launch_grid_sample_fwd<2>(in, grid, out, N, C, id, od, interp, padding, clamp_grid, (hipStream_t)stream);
}

// out = warp(in) and ride_out = warp(ride_in) (one channel) through the same grid.  2D: ONE launch of the direct-gather
// kernel (the taps of a sample are built once); 3D: the two launches advchain_grid_sample_fwd would make.  Either way
// every value is what two calls of advchain_grid_sample_fwd return.  flags bit 0: ride_out = (warp(ride_in) != 0).
int advchain_grid_sample_fwd_ride(const float* in, const float* grid, float* out, const float* ride_in, float* ride_out,
                                  int64_t N, int64_t C, int ndim, const int64_t* in_dims, const int64_t* out_dims, int interp,
                                  int padding, int clamp_grid, int flags, void* stream) {
  ADVCHAIN_CHECK_ARG(ride_in && ride_out && ride_out != out, "grid_sample_fwd_ride: null/aliased rider");
  ADVCHAIN_CHECK_ARG((flags & ~1) == 0, "grid_sample_fwd_ride: unknown flags");
  if (ndim == 2 && in && grid && out && dims_ok(ndim, in_dims) && dims_ok(ndim, out_dims, 1) && N > 0 && N < 65536 && C >= 1 &&
      (interp == INTERP_LINEAR || interp == INTERP_NEAREST) && padding >= 0 && padding <= 2) {
    const Dims id = make_dims(ndim, in_dims), od = make_dims(ndim, out_dims);
    ADVCHAIN_CHECK_ARG(id.voxels() < (1ll << 31) && od.voxels() < (1ll << 31), "grid_sample_fwd_ride: per-sample volume too large");
    return launch_grid_sample_fwd<2>(in, grid, out, N, C, id, od, interp, padding, clamp_grid & 1, (hipStream_t)stream, ride_in,
                                     ride_out, flags & 1);
  }
  int rc = advchain_grid_sample_fwd(in, grid, out, N, C, ndim, in_dims, out_dims, interp, padding, clamp_grid, stream);
  if (rc != ADVCHAIN_OK) return rc;
  rc = advchain_grid_sample_fwd(ride_in, grid, ride_out, N, 1, ndim, in_dims, out_dims, interp, padding, clamp_grid, stream);
  if (rc != ADVCHAIN_OK || !(flags & 1) || N == 0) return rc;
  int64_t total = N;
  for (int a = 0; a < ndim; ++a) total *= out_dims[a];
  return advchain_nonzero_mask(ride_out, ride_out, total, stream);
}

int advchain_grid_sample_bwd(const float* grad_out, const float* in, const float* grid, float* grad_in,
                             float* grad_grid, int32_t* workspace, int64_t N, int64_t C, int ndim,
                             const int64_t* in_dims, const int64_t* out_dims, int interp, int padding, int clamp_grid,
                             int halo, void* stream) {
  ADVCHAIN_CHECK_ARG(grad_out && in && grid, "grid_sample_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(grad_in || grad_grid, "grid_sample_bwd: nothing to compute");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, in_dims) && dims_ok(ndim, out_dims, 1), "grid_sample_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "grid_sample_bwd: bad N/C");
  ADVCHAIN_CHECK_ARG(interp == INTERP_LINEAR || interp == INTERP_NEAREST, "grid_sample_bwd: interp");
  ADVCHAIN_CHECK_ARG(padding >= 0 && padding <= 2, "grid_sample_bwd: padding");
  if (N == 0) return ADVCHAIN_OK;
  const Dims id = make_dims(ndim, in_dims), od = make_dims(ndim, out_dims);
  ADVCHAIN_CHECK_ARG(id.voxels() < (1ll << 31) && od.voxels() < (1ll << 31), "grid_sample_bwd: per-sample volume too large");
  const bool same = id.s0 == od.s0 && id.s1 == od.s1 && id.s2 == od.s2;
  if (workspace && grad_in && same && interp == INTERP_LINEAR && C <= 4) {
    if (ndim == 2 && halo <= -2) {   // exact bound of a few pixels: whole-row owner-computes scatter (scatter_march.hip)
      const int rr = advchain_scatter_rows2d_launch(false, grad_out, in, grid, grad_in, grad_grid, N, C, id, padding, clamp_grid,
                                                    -halo, workspace, (hipStream_t)stream);
      if (rr != ADVCHAIN_ERR_UNSUPPORTED) return rr;
    }
    // small displacement bound: gather form (adjoint_gather.hip); otherwise the LDS-tiled owner-computes scatter
    const int rc = advchain_warp_adjoint_gather_launch(grad_out, in, grid, grad_in, grad_grid, N, C, ndim, id, padding,
                                                       clamp_grid, workspace, halo, (hipStream_t)stream);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
    if (ndim == 3 && halo <= -2) {   // exact bound of 2..4 voxels: owner-computes march, no global atomics
      const int rm = advchain_scatter_march_launch(false, grad_out, in, grid, grad_in, grad_grid, N, C, id, padding, clamp_grid,
                                                   -halo, workspace, (hipStream_t)stream);
      if (rm != ADVCHAIN_ERR_UNSUPPORTED) return rm;
    }
    const int rw = advchain_scatter_window_launch(false, grad_out, in, grid, grad_in, grad_grid, N, C, ndim, id, padding,
                                                  clamp_grid, halo, nullptr, (hipStream_t)stream, workspace);   // source-tiled window
    if (rw != ADVCHAIN_ERR_UNSUPPORTED) return rw;
    return advchain_scatter_tiled_launch(false, grad_out, in, grid, grad_in, grad_grid, N, C, ndim, id, padding,
                                         clamp_grid, workspace, 0, halo < 0 ? -halo : halo, (hipStream_t)stream);
  }
  if (workspace && grad_in) advchain_zero_async(grad_in, sizeof(float) * N * C * id.voxels(), (hipStream_t)stream);
  return ndim == 3 ? launch_grid_sample_bwd<3>(grad_out, in, grid, grad_in, grad_grid, N, C, id, od, interp, padding, clamp_grid, (hipStream_t)stream)
                   : launch_grid_sample_bwd<2>(grad_out, in, grid, grad_in, grad_grid, N, C, id, od, interp, padding, clamp_grid, (hipStream_t)stream);
}

int advchain_compose_self_fwd(const float* phi, float* out, const float* phi0, int64_t N, int ndim,
                              const int64_t* dims, int final_mode, float* disp_out, void* stream) {
  const int disp_hint = (final_mode >> 8) & 0xff;   // bits 8..15: displacement estimate of phi in voxels (0 = unknown), a performance hint
  final_mode &= 0xff;
  ADVCHAIN_CHECK_ARG(phi && out && phi != out, "compose_self_fwd: null/aliased pointer");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "compose_self_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(final_mode == 0 || (final_mode == 1 && phi0), "compose_self_fwd: final_mode 1 needs phi0");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "compose_self_fwd: bad N");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = make_dims(ndim, dims);
  const int64_t V = d.voxels();
  ADVCHAIN_CHECK_ARG(V < (1ll << 31), "compose_self_fwd: per-sample volume too large");
  {
    const int rc = advchain_sample_tiled_launch(true, phi, nullptr, out, phi0, N, ndim, ndim, d, PAD_BORDER, 0,
                                                final_mode, 0, disp_out, (hipStream_t)stream, disp_hint);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
  const bool vec4 = use_unroll(V, ndim);
  hipStream_t st = (hipStream_t)stream;
  // 2D: two voxels per thread (twice the waves of the 4-voxel form at fewer registers: 12.0 against 13.4 us per cfg-2
  // squaring once the gathers are issued together)
  static const bool unr2 = true && true;
  if (unr2 && vec4 && ndim == 2) {
    dim3 g2(advchain_blocks(V, kBlock * 2), (unsigned)N);
    hipLaunchKernelGGL((k_compose_self_fwd<2, 2>), g2, dim3(kBlock), 0, st, phi, out, phi0, d, final_mode, disp_out);
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  dim3 g(advchain_blocks(V, kBlock * (vec4 ? 4 : 1)), (unsigned)N), b(kBlock);
  if (ndim == 3) {
    if (vec4) hipLaunchKernelGGL((k_compose_self_fwd<3, 4>), g, b, 0, st, phi, out, phi0, d, final_mode, disp_out);
    else hipLaunchKernelGGL((k_compose_self_fwd<3, 1>), g, b, 0, st, phi, out, phi0, d, final_mode, disp_out);
  } else {
    if (vec4) hipLaunchKernelGGL((k_compose_self_fwd<2, 4>), g, b, 0, st, phi, out, phi0, d, final_mode, disp_out);
    else hipLaunchKernelGGL((k_compose_self_fwd<2, 1>), g, b, 0, st, phi, out, phi0, d, final_mode, disp_out);
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

static int compose_self_bwd_impl(const float* grad_out, const float* phi, float* grad_phi, int32_t* workspace, int chain,
                                 int halo, int64_t N, int ndim, const int64_t* dims, void* stream, int rm_flags);

int advchain_compose_self_bwd(const float* grad_out, const float* phi, float* grad_phi, int32_t* workspace, int chain,
                              int halo, int64_t N, int ndim, const int64_t* dims, void* stream) {
  return compose_self_bwd_impl(grad_out, phi, grad_phi, workspace, chain, halo, N, ndim, dims, stream, 0);
}

// rm_flags: see advchain_scatter_rows2d_launch (consecutive whole-row launches of a chain hand their row maxima on)
static int compose_self_bwd_impl(const float* grad_out, const float* phi, float* grad_phi, int32_t* workspace, int chain,
                                 int halo, int64_t N, int ndim, const int64_t* dims, void* stream, int rm_flags) {
  ADVCHAIN_CHECK_ARG(grad_out && phi && grad_phi, "compose_self_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "compose_self_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "compose_self_bwd: bad N");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = make_dims(ndim, dims);
  const int64_t V = d.voxels();
  ADVCHAIN_CHECK_ARG(V < (1ll << 31), "compose_self_bwd: per-sample volume too large");
  if (workspace) {
    if (ndim == 2 && halo <= -2) {   // exact bound of a few pixels: whole-row owner-computes scatter (scatter_march.hip)
      const int rr = advchain_scatter_rows2d_launch(true, grad_out, phi, phi, grad_phi, nullptr, N, ndim, d, PAD_BORDER, 0, -halo,
                                                    workspace, (hipStream_t)stream, rm_flags);
      if (rr != ADVCHAIN_ERR_UNSUPPORTED) return rr;
    }
    // sub-voxel steps of the squaring chain: gather form (adjoint_gather.hip); otherwise the LDS-tiled scatter
    const int rc = advchain_self_adjoint_gather_launch(grad_out, phi, grad_phi, N, ndim, d, workspace, chain, halo,
                                                       (hipStream_t)stream);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
    if (ndim == 3 && halo <= -2) {   // exact bound of 2..4 voxels: owner-computes march, no global atomics
      const int rm = advchain_scatter_march_launch(true, grad_out, phi, phi, grad_phi, nullptr, N, ndim, d, PAD_BORDER, 0, -halo,
                                                   workspace, (hipStream_t)stream);
      if (rm != ADVCHAIN_ERR_UNSUPPORTED) return rm;
    }
    const int rw = advchain_scatter_window_launch(true, grad_out, phi, phi, grad_phi, nullptr, N, ndim, ndim, d, PAD_BORDER,
                                                  0, halo, workspace, (hipStream_t)stream, workspace);   // source-tiled window
    if (rw != ADVCHAIN_ERR_UNSUPPORTED) return rw;
    return advchain_scatter_tiled_launch(true, grad_out, phi, phi, grad_phi, nullptr, N, ndim, ndim, d, PAD_BORDER, 0,
                                         workspace, chain, halo < 0 ? -halo : halo, (hipStream_t)stream);
  }
  const bool vec4 = use_unroll(V, ndim);
  dim3 g(advchain_blocks(V, kBlock * (vec4 ? 4 : 1)), (unsigned)N), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  if (ndim == 3) {
    if (vec4) hipLaunchKernelGGL((k_compose_self_bwd<3, 4>), g, b, 0, st, grad_out, phi, grad_phi, d);
    else hipLaunchKernelGGL((k_compose_self_bwd<3, 1>), g, b, 0, st, grad_out, phi, grad_phi, d);
  } else {
    if (vec4) hipLaunchKernelGGL((k_compose_self_bwd<2, 4>), g, b, 0, st, grad_out, phi, grad_phi, d);
    else hipLaunchKernelGGL((k_compose_self_bwd<2, 1>), g, b, 0, st, grad_out, phi, grad_phi, d);
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_affine_warp_fwd(const float* in, const float* theta, float* out, int64_t N, int64_t C, int ndim,
                             const int64_t* dims, int interp, int padding, void* stream) {
  ADVCHAIN_CHECK_ARG(in && theta && out, "affine_warp_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "affine_warp_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "affine_warp_fwd: bad N/C");
  ADVCHAIN_CHECK_ARG(interp == INTERP_LINEAR || interp == INTERP_NEAREST, "affine_warp_fwd: interp");
  ADVCHAIN_CHECK_ARG(padding >= 0 && padding <= 2, "affine_warp_fwd: padding");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = make_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "affine_warp_fwd: per-sample volume too large");
  dim3 g(affine_grid_blocks(d), (unsigned)N), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  static const float thr = 7.f;   // measured optimum (was a tuning knob until round 4) (break-even measured at ~6 degrees)
  static const bool no_v = getenv("ADVCHAIN_NO_AFFINE_V") != nullptr;   // A/B knob
  if (interp == INTERP_LINEAR && padding == PAD_ZEROS && advchain_affine_box_fwd_launch(in, theta, out, N, C, ndim, d, st)) {
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  if (interp == INTERP_LINEAR && C <= 4 && !no_v) {
    const unsigned nw = (unsigned)affine_waves(d);
#define GO_V(DIM_, CT_, VEC_) do { \
    dim3 gv((nw + (kBlock / 64) * VEC_ - 1) / ((kBlock / 64) * VEC_), (unsigned)N); \
    DISPATCH_PAD(padding, { hipLaunchKernelGGL((k_affine_warp_fwd_v<DIM_, PAD, CT_, VEC_>), gv, b, 0, st, in, theta, out, d, thr, nw); }); } while (0)
    if (ndim == 3) {
      switch (C) { case 1: GO_V(3, 1, 2); break; case 2: GO_V(3, 2, 2); break; case 3: GO_V(3, 3, 1); break; default: GO_V(3, 4, 1); break; }
    } else {
      switch (C) { case 1: GO_V(2, 1, 4); break; case 2: GO_V(2, 2, 2); break; case 3: GO_V(2, 3, 2); break; default: GO_V(2, 4, 2); break; }
    }
#undef GO_V
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  DISPATCH_PAD(padding, {
    if (ndim == 3) {
      if (interp == INTERP_LINEAR) hipLaunchKernelGGL((k_affine_warp_fwd<3, INTERP_LINEAR, PAD>), g, b, 0, st, in, theta, out, (int)C, d, thr);
      else hipLaunchKernelGGL((k_affine_warp_fwd<3, INTERP_NEAREST, PAD>), g, b, 0, st, in, theta, out, (int)C, d, thr);
    } else {
      if (interp == INTERP_LINEAR) hipLaunchKernelGGL((k_affine_warp_fwd<2, INTERP_LINEAR, PAD>), g, b, 0, st, in, theta, out, (int)C, d, thr);
      else hipLaunchKernelGGL((k_affine_warp_fwd<2, INTERP_NEAREST, PAD>), g, b, 0, st, in, theta, out, (int)C, d, thr);
    }
  });
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// out = warp(in) and ride_out = warp(ride_in) (one channel) under the same theta.  Linear / zeros on 16-byte rows: ONE
// launch of the box kernel (taps and box geometry once, the rider is one more channel of the walk); otherwise the two
// launches advchain_affine_warp_fwd would make.  Every value is what two calls of advchain_affine_warp_fwd return.
// flags bit 0: ride_out = (warp(ride_in) != 0).
int advchain_affine_warp_fwd_ride(const float* in, const float* theta, float* out, const float* ride_in, float* ride_out,
                                  int64_t N, int64_t C, int ndim, const int64_t* dims, int interp, int padding, int flags,
                                  void* stream) {
  ADVCHAIN_CHECK_ARG(ride_in && ride_out && ride_out != out, "affine_warp_fwd_ride: null/aliased rider");
  ADVCHAIN_CHECK_ARG((flags & ~1) == 0, "affine_warp_fwd_ride: unknown flags");
  if (in && theta && out && dims_ok(ndim, dims) && N > 0 && N < 65536 && C >= 1 && interp == INTERP_LINEAR && padding == PAD_ZEROS) {
    const Dims d = make_dims(ndim, dims);
    if (d.voxels() < (1ll << 31) &&
        advchain_affine_box_fwd_launch(in, theta, out, N, C, ndim, d, (hipStream_t)stream, ride_in, ride_out, flags & 1)) {
      ADVCHAIN_LAUNCH_CHECK();
      return ADVCHAIN_OK;
    }
  }
  int rc = advchain_affine_warp_fwd(in, theta, out, N, C, ndim, dims, interp, padding, stream);
  if (rc != ADVCHAIN_OK) return rc;
  rc = advchain_affine_warp_fwd(ride_in, theta, ride_out, N, 1, ndim, dims, interp, padding, stream);
  if (rc != ADVCHAIN_OK || !(flags & 1) || N == 0) return rc;
  int64_t total = N;
  for (int a = 0; a < ndim; ++a) total *= dims[a];
  return advchain_nonzero_mask(ride_out, ride_out, total, stream);
}


// ---- the whole scaling-and-squaring chain in one call (the same launches as n calls of the two entries above; the host
// side of a solver step is as long as its GPU side, and a chain is 2 x n of its ~700 launches)
// How many leading squarings of a 2D chain the hints allow in one fused launch, and their row halos (4 bits a level): the
// squarings whose hinted input displacement (bits 8.. of a hint: 1/1024 pixel) leaves a quarter of room below one pixel for
// the field to grow between two ascent steps, and the squarings behind them while the window (the sum of the levels' row
// halos either side) stays small: a level whose input moves less than h pixels takes its corners from h rows either side.
static int fuse_rule(int n, const int32_t* hints, int fuse_max, int* halos_out) {
  static const int hs_max = getenv("ADVCHAIN_FUSE2D_HS") ? atoi(getenv("ADVCHAIN_FUSE2D_HS")) : 5;   // A/B knob: rows of halo in all
  int k = 0, halos = 0, hs = 0;
  while (k < n - 1 && k < fuse_max && ((unsigned)hints[k] >> 8) != 0) {
    const float e = (float)((unsigned)hints[k] >> 8) * (1.25f / 1024.f);
    const int h = e < 1.f ? 1 : (int)e + 1;
    if (h > 15 || hs + h > (hs_max > k + 1 ? hs_max : k + 1)) break;     // (sub-pixel levels always fit, as before)
    halos |= h << (4 * k);
    hs += h;
    ++k;
  }
  *halos_out = halos;
  return k;
}

static int fuse_max_levels() {
  static const int fuse_max = getenv("ADVCHAIN_FUSE2D_MAX") ? atoi(getenv("ADVCHAIN_FUSE2D_MAX")) : 5;   // A/B knob (0 = off)
  return fuse_max;
}

int advchain_expo_chain_fused_levels(int64_t N, int ndim, const int64_t* dims, int n, const int32_t* hints) {
  if (ndim != 2 || !hints || !dims_ok(ndim, dims) || n < 1 || n > 64 || fuse_max_levels() < 2) return 0;
  int halos = 0;
  const int k = fuse_rule(n, hints, fuse_max_levels(), &halos);
  if (k < 2) return 0;
  return advchain_expo_fused_fwd2d_launch(nullptr, nullptr, N, make_dims(ndim, dims), k, halos, nullptr, nullptr, nullptr, true) == ADVCHAIN_OK ? k : 0;
}

int advchain_expo_chain_fwd(const float* phi0, float* fields, float* pos, int64_t N, int ndim, const int64_t* dims, int n,
                            float* disp_rows, const int32_t* hints, float* fuse_flag, void* stream) {
  ADVCHAIN_CHECK_ARG(phi0 && pos && n >= 1 && n <= 64 && (n == 1 || fields), "expo_chain_fwd: null pointer / bad n");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "expo_chain_fwd: bad dims");
  const int64_t F = N * ndim * make_dims(ndim, dims).voxels();
  const float* src = phi0;
  // 2D: the leading squarings whose inputs the hints put below one pixel run as ONE launch (expo_fused2d.hip: whole-row LDS
  // windows, bit-identical fields).  The kernel verifies the premise itself and raises *fuse_flag when a window moves too
  // far for some of its levels; ONE repeat launch behind it (k_expo_repeat2d) then runs exactly those levels the ordinary
  // way, and returns at once while the flag is down.
  const int fuse_max = fuse_max_levels();
  int fused = 0;
  if (ndim == 2 && fuse_flag && hints && fuse_max >= 2) {
    // how many: the leading squarings whose hinted input displacement (bits 8.. of a hint: 1/1024 pixel) leaves a quarter
    // of room below one pixel for the field to grow between two ascent steps; a level some window cannot do after all is
    // repeated by the launch behind the fused kernel
    // ... and the squarings behind them while the window (the sum of the levels' row halos either side) stays small: a
    // level whose input moves less than h pixels takes its corners from h rows either side
    int halos = 0;
    const int k = fuse_rule(n, hints, fuse_max, &halos);
    if (k >= 2) {
      const int rf = advchain_expo_fused_fwd2d_launch(phi0, fields, N, make_dims(ndim, dims), k, halos, disp_rows, fuse_flag,
                                                      (hipStream_t)stream);
      if (rf == ADVCHAIN_OK) {
        // ONE repeat launch behind it: returns at once unless some window of the fused kernel stopped early, else runs the
        // missing levels on a persistent grid every workgroup of which is resident (2 per CU asked for, 8 fit)
        fused = k;
        const Dims d2 = make_dims(ndim, dims);
        const int nbx = advchain_blocks(d2.voxels(), kBlock * 2);
        // grid: at most HALF of what the CURRENT device holds at once (occupancy of this kernel x its CU count, cached per
        // device) -- every workgroup of the grid is then resident together with room to spare, which the counter barrier needs
        static int resident[64] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        int& cap_wg = resident[dev & 63];
        if (!cap_wg) {
          int cus = 0, per_cu = 0;
          (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_expo_repeat2d, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
          if (cus <= 0) cus = 64;
          const int half = per_cu * cus / 2;
          cap_wg = half > 2 * cus ? 2 * cus : (half < 1 ? 1 : half);
        }
        const int64_t items = N * (int64_t)nbx, cap = cap_wg;
        hipLaunchKernelGGL(k_expo_repeat2d, dim3((unsigned)(items < cap ? items : cap)), dim3(kBlock), 0, (hipStream_t)stream,
                           phi0, fields, F, d2, k, (int)N, nbx, disp_rows, fuse_flag, reinterpret_cast<unsigned int*>(fuse_flag + 1));
        ADVCHAIN_LAUNCH_CHECK();
      } else if (rf != ADVCHAIN_ERR_UNSUPPORTED) return rf;
    }
  }
  if (fused) src = fields + (int64_t)(fused - 1) * F;
  for (int m = fused; m + 1 < n; ++m) {
    float* dst = fields + (int64_t)m * F;
    const int rc = advchain_compose_self_fwd(src, dst, nullptr, N, ndim, dims, hints ? (hints[m] & 0xff) << 8 : 0,
                                             disp_rows ? disp_rows + (int64_t)(m + 1) * kDispSlots : nullptr, stream);
    if (rc != ADVCHAIN_OK) return rc;
    src = dst;
  }
  return advchain_compose_self_fwd(src, pos, phi0, N, ndim, dims, 1 | (hints ? (hints[n - 1] & 0xff) << 8 : 0),
                                   disp_rows ? disp_rows + (int64_t)n * kDispSlots : nullptr, stream);
}

int advchain_expo_chain_bwd(const float* grad_pos, const float* phi0, const float* fields, float* grad_phi0, float* scratch,
                            int32_t* workspace, const int32_t* halos, int64_t N, int ndim, const int64_t* dims, int n,
                            void* stream) {
  ADVCHAIN_CHECK_ARG(grad_pos && phi0 && grad_phi0 && n >= 1 && n <= 64 && (n == 1 || (fields && scratch)) && halos,
                     "expo_chain_bwd: null pointer / bad n");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "expo_chain_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(grad_phi0 != scratch && grad_pos != grad_phi0 && grad_pos != scratch, "expo_chain_bwd: aliased buffers");
  const int64_t F = N * ndim * make_dims(ndim, dims).voxels();
  // 2D: the steps at the END of the backward (squarings kf-1 .. 0) whose inputs have an EXACT sub-pixel bound run as one
  // launch (adjoint_fused2d.hip: k levels of the gather form without leaving LDS, bit-identical to the separate launches)
  int kf = 0;
  if (ndim == 2 && workspace && n >= 2) {
    while (kf < n && kf < 4 && halos[n - 1 - kf] == -1) ++kf;
    if (kf < 2) kf = 0;
  }
  const float* g = grad_pos;
  // 2D: consecutive whole-row scatters hand the row maxima of their output to the next one (its fixed-point scale): no
  // k_march_rowmax pre-pass between them
  const Dims dd = make_dims(ndim, dims);
  static const bool no_handover = getenv("ADVCHAIN_NO_ROWMAX_HANDOVER") != nullptr;   // A/B knob: a k_march_rowmax pre-pass per launch
  auto rows = [&](int i) {
    return !no_handover && ndim == 2 && workspace && i >= 0 && i < n - kf && halos[i] <= -2 && advchain_scatter_rows2d_takes(true, 2, dd, PAD_BORDER, -halos[i], N);
  };
  for (int i = 0; i < n - kf; ++i) {                 // squaring m = n-1 .. kf; the last step of the call writes grad_phi0
    const int m = n - 1 - i;
    const float* phi = m == 0 ? phi0 : fields + (int64_t)(m - 1) * F;
    float* out = ((m - kf) % 2 == 0) ? (kf ? scratch : grad_phi0) : (kf ? grad_phi0 : scratch);
    const int rm = rows(i) ? ((rows(i - 1) ? 1 : 0) | (rows(i + 1) ? 2 : 0) | ((i & 1) ? 4 : 0)) : 0;
    const int rc = compose_self_bwd_impl(g, phi, out, workspace, i > 0 ? 1 : 0, halos[i], N, ndim, dims, stream, rm);
    if (rc != ADVCHAIN_OK) return rc;
    g = out;
  }
  if (kf) {
    const int rf = advchain_adjoint_fused2d_launch(g, phi0, fields, grad_phi0, N, make_dims(ndim, dims), kf, workspace,
                                                   (hipStream_t)stream);
    if (rf == ADVCHAIN_ERR_UNSUPPORTED) {            // the shape does not fit: the separate launches (g sits in `scratch`)
      for (int m = kf - 1; m >= 0; --m) {
        const float* phi = m == 0 ? phi0 : fields + (int64_t)(m - 1) * F;
        float* out = (m % 2 == 0) ? grad_phi0 : scratch;
        if (out == g) {                              // keep input and output apart: one copy, only on this fallback route
          float* other = out == grad_phi0 ? scratch : grad_phi0;
          (void)hipMemcpyAsync(other, g, sizeof(float) * F, hipMemcpyDeviceToDevice, (hipStream_t)stream);
          g = other;
        }
        const int rc = advchain_compose_self_bwd(g, phi, out, workspace, (n - 1 - m) > 0 ? 1 : 0, halos[n - 1 - m], N, ndim, dims, stream);
        if (rc != ADVCHAIN_OK) return rc;
        g = out;
      }
    } else if (rf != ADVCHAIN_OK) {
      return rf;
    } else {
      ADVCHAIN_LAUNCH_CHECK();
    }
  }
  return ADVCHAIN_OK;
}

int64_t advchain_affine_warp_bwd_workspace(int64_t N, int ndim, const int64_t* dims) {
  if (!dims_ok(ndim, dims)) return -1;
  const Dims d = make_dims(ndim, dims);
  return N * (int64_t)affine_grid_blocks(d) * ndim * (ndim + 1) + N * (kGeoFloats + 1) + N * (int64_t)advchain_affine_box_tiles(ndim, d);  // floats
}

int advchain_affine_warp_bwd(const float* grad_out, const float* in, const float* theta, float* grad_in,
                             float* grad_theta, float* workspace, int64_t N, int64_t C, int ndim,
                             const int64_t* dims, int interp, int padding, void* stream) {
  ADVCHAIN_CHECK_ARG(grad_out && in && theta, "affine_warp_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(grad_in || grad_theta, "affine_warp_bwd: nothing to compute");
  ADVCHAIN_CHECK_ARG(workspace, "affine_warp_bwd: workspace required (advchain_affine_warp_bwd_workspace floats)");
  ADVCHAIN_CHECK_ARG(dims_ok(ndim, dims), "affine_warp_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "affine_warp_bwd: bad N/C");
  ADVCHAIN_CHECK_ARG(interp == INTERP_LINEAR || interp == INTERP_NEAREST, "affine_warp_bwd: interp");
  ADVCHAIN_CHECK_ARG(padding >= 0 && padding <= 2, "affine_warp_bwd: padding");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = make_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "affine_warp_bwd: per-sample volume too large");
  const int nb = affine_grid_blocks(d);
  dim3 g(nb, (unsigned)N), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  float* gpart = grad_theta ? workspace : nullptr;
  if (interp == INTERP_NEAREST && !grad_in) {
    // nearest: zero gradient w.r.t. theta
    advchain_zero_async(grad_theta, sizeof(float) * N * ndim * (ndim + 1), st);
    return ADVCHAIN_OK;
  }
  const int* mode = nullptr;
  int nbx_theta = -1;     // >= 0: the box kernel was asked for the theta gradient already (0 = it declined the shape)
  static const bool no_gather = getenv("ADVCHAIN_AFFINE_ATOMIC") != nullptr;  // A/B knob
  if (grad_in && workspace && interp == INTERP_LINEAR && padding == PAD_ZEROS && C <= 8 && !no_gather) {
    // gather formulation of grad_in (no atomics); samples it cannot handle are flagged and scattered below
    float* geo = workspace + N * (int64_t)nb * ndim * (ndim + 1);
    int* md = reinterpret_cast<int*>(geo + N * kGeoFloats);
    // the theta gradient FIRST when it is asked for: its walk over grad_out leaves max |grad_out| per output tile, the
    // fixed-point scale of the grad_in kernel (which otherwise reads its box of grad_out twice)
    float* tilemax = nullptr;
    if (gpart) {
      float* tm = reinterpret_cast<float*>(md + N);
      // (... and the per-sample geometry of the grad_in kernels on the side: no k_affine_geometry launch then)
      nbx_theta = advchain_affine_box_gtheta_launch(grad_out, in, theta, gpart, N, C, ndim, d, nb, st, tm, geo, md);
      if (nbx_theta > 0) tilemax = tm;
    }
    const bool have_geo = nbx_theta > 0;
    if (ndim == 3) {
      if (!have_geo) hipLaunchKernelGGL(k_affine_geometry<3>, dim3(advchain_blocks(N, 64)), dim3(64), 0, st, theta, geo, md, (int)N, d);
      if (advchain_affine_box_gin_launch(grad_out, theta, geo, md, grad_in, N, C, ndim, d, st, tilemax)) {}
      else if (C <= 1) hipLaunchKernelGGL((k_affine_gather_bwd<3, 1>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
      else if (C <= 4) hipLaunchKernelGGL((k_affine_gather_bwd<3, 4>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
      else hipLaunchKernelGGL((k_affine_gather_bwd<3, 8>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
    } else {
      if (!have_geo) hipLaunchKernelGGL(k_affine_geometry<2>, dim3(advchain_blocks(N, 64)), dim3(64), 0, st, theta, geo, md, (int)N, d);
      if (advchain_affine_box_gin_launch(grad_out, theta, geo, md, grad_in, N, C, ndim, d, st, tilemax)) {}
      else if (C <= 1) hipLaunchKernelGGL((k_affine_gather_bwd<2, 1>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
      else if (C <= 4) hipLaunchKernelGGL((k_affine_gather_bwd<2, 4>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
      else hipLaunchKernelGGL((k_affine_gather_bwd<2, 8>), g, b, 0, st, grad_out, theta, geo, md, grad_in, (int)C, d);
    }
    mode = md;
  } else if (grad_in) {
    advchain_zero_async(grad_in, sizeof(float) * N * C * d.voxels(), st);
  }
  int nb_theta = nb;      // block partials per sample of grad_theta
  if (gpart && interp == INTERP_LINEAR && padding == PAD_ZEROS && (!grad_in || mode)) {
    // theta gradient through the LDS-staged source box; grad_in (when asked for) comes from the lattice gather above,
    // and only the samples it flagged still need the scatter kernel -- without its theta part
    const int nbx = nbx_theta >= 0 ? nbx_theta : advchain_affine_box_gtheta_launch(grad_out, in, theta, gpart, N, C, ndim, d, nb, st, nullptr);
    if (nbx > 0) { nb_theta = nbx; gpart = nullptr; }
  }
  // the launch below scatters grad_in for the samples the gather form flagged (none, as a rule).  When the theta gradient is
  // waiting as block partials of the box kernel, it reduces them on the way: no k_reduce_partials launch behind it
  const int Kt = ndim * (ndim + 1);
  const bool fold_reduce = grad_theta && !gpart && grad_in && interp == INTERP_LINEAR && nb >= Kt;
  if (gpart || grad_in)
  DISPATCH_PAD(padding, {
    if (ndim == 3) {
      if (interp == INTERP_LINEAR) launch_affine_bwd<3, INTERP_LINEAR, PAD>(g, b, st, grad_out, in, theta, grad_in, gpart, (int)C, d, mode,
                                                                            fold_reduce ? workspace : nullptr, fold_reduce ? grad_theta : nullptr, nb_theta);
      else launch_affine_bwd<3, INTERP_NEAREST, PAD>(g, b, st, grad_out, in, theta, grad_in, nullptr, (int)C, d, mode);
    } else {
      if (interp == INTERP_LINEAR) launch_affine_bwd<2, INTERP_LINEAR, PAD>(g, b, st, grad_out, in, theta, grad_in, gpart, (int)C, d, mode,
                                                                            fold_reduce ? workspace : nullptr, fold_reduce ? grad_theta : nullptr, nb_theta);
      else launch_affine_bwd<2, INTERP_NEAREST, PAD>(g, b, st, grad_out, in, theta, grad_in, nullptr, (int)C, d, mode);
    }
  });
  ADVCHAIN_LAUNCH_CHECK();
  if (grad_theta && !fold_reduce) {
    if (interp == INTERP_NEAREST) {
      advchain_zero_async(grad_theta, sizeof(float) * N * ndim * (ndim + 1), st);
    } else {
      const int K = ndim * (ndim + 1);
      hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)N, K), dim3(kBlock), 0, st, workspace, grad_theta, nb_theta, K);
      ADVCHAIN_LAUNCH_CHECK();
    }
  }
  return ADVCHAIN_OK;
}

}  // extern "C"
