// Owner-computes scatters for the sampler backward with an EXACT displacement bound (gfx950): z-marching in 3D (bounds of
// 2..4 voxels; 5..8 as one launch per channel, see the launcher), whole-row tiles in 2D (4, 8, 16 px; k_scatter_rows2d below).
//
// Above one voxel the gather form is too expensive ((2H+1)^3 tent products per output) and the source-tiled window
// scatter (scatter_window.hip) pays for its freedom from any bound with global float atomics: every window cell is
// flushed with one (4.2 per sample and channel for its 4 x 8 x 32 tiles), the destination has to be zero-filled first,
// and the coordinate path of a self-composition goes through three more atomics per sample.  With the bound H the
// forward MEASURED (ops.squaring_halo / ops.warp_halo) the owner can do everything itself:
//   * a workgroup owns TY output rows (whole x rows, lane <-> x) and walks a chunk of planes; at step zp it takes the
//     samples of plane zp in rows y0-H .. y0+TY+H-1 straight from global memory (coalesced 4-byte loads: a sample row
//     is read once per workgroup, nothing to stage), builds their taps with the sampler's own arithmetic and deposits
//     the corners that fall in ITS rows and ITS chunk into a ring of 2H+3 accumulator planes in LDS -- 32-bit fixed
//     point, value * 2^(21..23) / max|grad_out| over the rows the workgroup visits (march_fix_scale; LDS integer atomics
//     run at LDS rate, LDS float atomics do not);
//   * after plane zp the output plane zp-H has seen every sample that can reach it: it is converted, the coordinate
//     path of its own samples is added (self-composition) or stored (grad_grid), and it leaves with PLAIN stores.
// No global atomic, no zero-fill, fixed summation order up to the commutativity of integer adds: deterministic.
// The price is halo work: samples are visited (TY+2H)/TY x (ZC+2H)/ZC times (loads, taps, range tests -- not atomics).
//
// Contract: |unnormalize(grid) - s| < H voxels for every sample, guaranteed by the caller (negative `halo`).
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

// d(sample)/d(unnormalised coordinate) * go in difference form: ax = go * sum_zy wz wy (v[z][y][1] - v[z][y][0]) and so on
// -- a third of the instructions of the signed-product form of sample_linear_bwd (these kernels are VALU bound), the same
// value up to the rounding of the differences.
template <int DIM, int PAD>
__device__ __forceinline__ void coord_path_diff(const float* __restrict__ in, float go, const Taps<DIM, PAD>& t, const Dims& d,
                                                float& ax, float& ay, float& az) {
  const CornerOffsets<DIM, PAD> o(t, d);
  float v[2][2][2];
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) v[cz][cy][cx] = in[o.at(cz, cy, cx)];
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) v[cz][cy][cx] = t.ok(cz, cy, cx) ? v[cz][cy][cx] : 0.f;
  if (DIM == 3) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int cz = 0; cz < 2; ++cz) {
      sx = fmaf(t.wz(cz), fmaf(t.y.w1, v[cz][1][1] - v[cz][1][0], t.y.w0 * (v[cz][0][1] - v[cz][0][0])), sx);
      sy = fmaf(t.wz(cz), fmaf(t.x.w1, v[cz][1][1] - v[cz][0][1], t.x.w0 * (v[cz][1][0] - v[cz][0][0])), sy);
    }
    float sz = 0.f;
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
      sz = fmaf(t.wy(cy), fmaf(t.x.w1, v[1][cy][1] - v[0][cy][1], t.x.w0 * (v[1][cy][0] - v[0][cy][0])), sz);
    ax = fmaf(go, sx, ax);
    ay = fmaf(go, sy, ay);
    az = fmaf(go, sz, az);
  } else {
    ax = fmaf(go, fmaf(t.y.w1, v[0][1][1] - v[0][1][0], t.y.w0 * (v[0][0][1] - v[0][0][0])), ax);
    ay = fmaf(go, fmaf(t.x.w1, v[0][1][1] - v[0][0][1], t.x.w0 * (v[0][1][0] - v[0][0][0])), ay);
  }
}

// Fixed-point resolution: a cell can receive a corner of every sample within H+1 voxels of it, (2H+2)^3 at most, each of
// weight <= 1 and |grad_out| <= the workgroup's max: 2^23 / 2^22 / 2^21 for H = 2 / 3 / 4 keeps any sum below 2^31.
// (H = 5..8: 18^3 deposits, 2^18.)
__device__ __forceinline__ float march_fix_scale(int H) { return H <= 2 ? 8388608.f : (H == 3 ? 4194304.f : (H == 4 ? 2097152.f : 262144.f)); }

// max |grad_out| of every x row (over its channels) -> rowmax[n][z][y].  The fixed-point scale of a workgroup comes from
// the rows IT visits, not from the whole batch: gradients are heavy-tailed (edges), and a global scale left the small
// ones with a median relative error of 2.5e-4 (measured, randn^5 grad_out); a local one also makes the result of a
// sample independent of what else is in the batch (a sharded batch reproduces the whole one).
template <int C>
__global__ void __launch_bounds__(kBlock) k_march_rowmax(const float* __restrict__ x, float* __restrict__ rowmax, Dims d, int rows_per_n,
                                                                int ctot = C) {   // ctot: channels per batch entry of x (a slice of C of them is looked at)
  // four rows per wave, their loads in flight together (a row per wave: 65536 waves of one dependent load each, 10.8 us
  // for 16.8 MB)
  constexpr int RW = 4;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RW;
  const int n = blockIdx.y;
  if (row0 >= rows_per_n) return;
  const int V = (int)d.voxels();
  const float* p = x + (int64_t)n * ctot * V;
  float m[RW];
  bool bad[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) { m[r] = 0.f; bad[r] = false; }
  for (int xx = lane; xx < d.s2; xx += 64) {
    float v[RW][C];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int row = min(row0 + r, rows_per_n - 1);
#pragma unroll
      for (int c = 0; c < C; ++c) v[r][c] = p[(int64_t)c * V + (int64_t)row * d.s2 + xx];
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        m[r] = fmaxf(m[r], fabsf(v[r][c]));
        bad[r] = bad[r] || !(fabsf(v[r][c]) <= 3.0e38f);      // NaN (which fmaxf drops) or inf
      }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
    // a non-finite gradient must not come out as finite numbers: the row's maximum becomes +inf, the fixed-point scale of
    // every workgroup that visits it 0 and its conversion factor inf -- its outputs are 0 * inf = NaN
    if (__ballot(bad[r]) != 0) m[r] = __int_as_float(0x7f800000);
  }
  if (lane < RW && row0 + lane < rows_per_n) {
    float mine = m[0];
#pragma unroll
    for (int r = 1; r < RW; ++r) mine = lane == r ? m[r] : mine;
    rowmax[(int64_t)n * rows_per_n + row0 + lane] = mine;
  }
}

// SELF : in == grid == phi (C == 3); gin receives value path + coordinate path     (advchain_compose_self_bwd)
// !SELF: gin <- value path; GG: ggrid <- coordinate path                            (advchain_grid_sample_bwd)
// NWV waves per workgroup share one accumulator tile, so LDS does not cap the waves of a CU: 8 for 8 owned rows, 4 for 4
// (with 8 waves on 4 + 2H rows half of them idle in the second round of a step: measured 380 vs 252 us, C = 4, H = 4).
// CT > C (C == 1): channel-sliced launch for bounds of 5..8 voxels -- a ring of 2H+3 planes of ONE channel still fits 8
// owned rows (19 x 8 x 64 cells = 38 KiB); the launcher runs one launch per channel c0, each keeps the deposits of its
// channel; the coordinate path (all CT channels) is evaluated by every launch of a self-composition (it adds its own axis)
// and by the first one of an image warp (GG).
template <int PAD, int C, bool SELF, bool GG, int NWV, int CT = C, bool BIG = false>
__global__ void __launch_bounds__(NWV * 64)
k_scatter_march3d(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                  float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int zc, int TY, int H, int NS,
                  int clamp_grid, int32_t* __restrict__ ws, int nseg, int c0) {
  constexpr bool SL = CT != C;
  extern __shared__ int acc[];                   // [slot 2H+3][C][TY][64]
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows longer than 64 voxels: x segments of 64 - 2H owned lanes with H halo lanes either side (a sample in a halo lane
  // is visited by both neighbours; each keeps what lands in its own columns)
  int rem = blockIdx.x;
  const int seg = rem % nseg;
  rem /= nseg;
  const int ty = rem % n1, tz = rem / n1;
  const int xbase = nseg > 1 ? seg * (64 - 2 * H) - H : 0;                      // x of lane 0
  const int xo0 = nseg > 1 ? xbase + H : 0, xo1 = nseg > 1 ? min(xbase + 64 - H, d.s2) : d.s2;   // owned columns
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const int plane_cells = C * TY * 64;
  const float* gn = grid + (int64_t)n * 3 * V;
  const float* gon_all = gout + (int64_t)n * CT * V;
  const float* gon = gon_all + (int64_t)c0 * V;
  const float* inn = in + (int64_t)n * CT * V;
  float* ginn = gin + (int64_t)n * CT * V + (int64_t)c0 * V;
  // fixed-point scale from the rows this workgroup visits
  __shared__ float wmax[NWV];
  {
    const float* rowmax = reinterpret_cast<const float*>(ws + 4) + (int64_t)n * d.s0 * d.s1;
    const int ya = max(y0 - H, 0), yn = min(y0 + TY + H, d.s1) - ya;
    const int zlo = max(za - H, 0), zn = min(zb + H, d.s0) - zlo;
    float m = 0.f;
    for (int i = threadIdx.x; i < yn * zn; i += NWV * 64) m = fmaxf(m, rowmax[(zlo + i / yn) * d.s1 + ya + i % yn]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) wmax[wave] = m;
  }
  for (int i = threadIdx.x; i < NS * plane_cells; i += NWV * 64) acc[i] = 0;
  __syncthreads();
  float gmax = wmax[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, wmax[w]);
  const float fix = march_fix_scale(H);
  const float scale = gmax > 0.f ? fix / gmax : 0.f, inv = gmax / fix;
  if (SELF && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[3] = -1;   // no max|result| from this launch
  const int xs = xbase + lane;
  const bool xin = xs >= 0 && xs < d.s2, xown = xs >= xo0 && xs < xo1;
  const int xl = min(max(xs, 0), d.s2 - 1);
  const int yend = min(y0 + TY, d.s1);

  constexpr int MAXR = (NWV == 8 && !BIG) ? 2 : 3;  // sample rows per wave and step: TY + 2H <= 8 + 8 (8 waves), 4 + 8 (4), 8 + 16 (BIG: H = 5..8)
  constexpr int MAXF = 1;                        // output rows per wave and step: TY <= NWV
  // requests one step ahead; not for C = 4 on 4 waves: 181 VGPRs, two workgroups a CU, 414 us where 252 is possible
  constexpr bool PF = C < 4 || NWV == 8;
  const int nrows = TY + 2 * H;
  // Everything a step needs from global memory is requested one step ahead (a step at a time the kernel was a chain of
  // exposed memory round trips: 16 waves a CU do not hide them).
  auto load_samples = [&](int zp, float (&g)[MAXR][3], float (&go)[MAXR][C]) {
    const int zq = min(max(zp, 0), d.s0 - 1);
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int r = wave + k * NWV;
      if (r >= nrows) continue;                                      // wave-uniform
      const int ys = min(max(y0 - H + r, 0), d.s1 - 1);              // clamped: no per-lane branch around the loads
      const int s = (zq * d.s1 + ys) * d.s2 + xl;
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = gn[(int64_t)a * V + s];
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = gon[(int64_t)c * V + s];
    }
  };
  auto load_own = [&](int zt, float (&g)[MAXF][3], float (&go)[MAXF][C]) {
    if (!(SELF || GG)) return;
    const int zq = min(max(zt, 0), d.s0 - 1);
#pragma unroll
    for (int k = 0; k < MAXF; ++k) {
      const int uy = min(y0 + wave + k * NWV, d.s1 - 1);
      const int s = (zq * d.s1 + uy) * d.s2 + xl;
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = gn[(int64_t)a * V + s];
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = gon[(int64_t)c * V + s];
    }
  };
  float g[MAXR][3], go[MAXR][C], fg[MAXF][3], fgo[MAXF][C];
  if (PF) {
    load_samples(za - H, g, go);
    load_own(za - 2 * H, fg, fgo);
  }
  for (int zp = za - H; zp < zb + H; ++zp) {
    float g1[PF ? MAXR : 1][3], go1[PF ? MAXR : 1][C], fg1[MAXF][3], fgo1[MAXF][C];
    if constexpr (PF) {
      load_samples(zp + 1, g1, go1);
      load_own(zp + 1 - H, fg1, fgo1);
    } else {
      load_samples(zp, g, go);
      load_own(zp - H, fg, fgo);
    }
    // ---- deposits of sample plane zp (rows y0-H .. y0+TY+H-1): only what lands in the owned rows and planes is kept
    if (zp >= 0 && zp < d.s0) {
      const int sz = ((zp % NS) + NS) % NS;                          // slot of plane zp (scalar)
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const int r = wave + k * NWV;
        const int ys = y0 - H + r;
        if (r >= nrows || ys < 0 || ys >= d.s1) continue;            // wave-uniform
        if (clamp_grid) { g[k][0] = clamp_unit(g[k][0]); g[k][1] = clamp_unit(g[k][1]); g[k][2] = clamp_unit(g[k][2]); }
        // The kernel is VALU bound (4 waves a SIMD, each 22% of its cycles in VALU issue) and most visits of halo rows
        // and halo planes deposit nothing here: the y and z taps alone decide that, for the whole wave.
        Taps<3, PAD> t;
        t.y = make_tap<PAD>(g[k][1], d.s1);
        t.z = make_tap<PAD>(g[k][2], d.s0);
        const bool reach = xin && t.y.i0 + 1 >= y0 && t.y.i0 < yend && t.z.i0 + 1 >= max(za, zp - H) && t.z.i0 < min(zb, zp + H + 2);
        if (__ballot(reach) == 0) continue;
        t.x = make_tap<PAD>(g[k][0], d.s2);
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
          const int pz = t.z.i0 + cz;
          const bool okz = pz >= za && pz < zb && pz >= zp - H && pz <= zp + H + 1;
          int slot = sz + (pz - zp);
          slot += slot < 0 ? NS : 0;
          slot -= slot >= NS ? NS : 0;
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const int py = t.y.i0 + cy;
            const bool oky = py >= y0 && py < yend;
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
              const int pxx = t.x.i0 + cx;
              if (!(xin && okz && oky && t.ok(cz, cy, cx) && pxx >= xo0 && pxx < xo1)) continue;
              const float wsc = t.w(cz, cy, cx) * scale;
              int* cell = acc + slot * plane_cells + (py - y0) * 64 + (pxx - xbase);
#pragma unroll
              for (int c = 0; c < C; ++c) atomicAdd(cell + c * TY * 64, __float2int_rn(wsc * go[k][c]));
            }
          }
        }
      }
    }
    // ---- coordinate path of the rows this wave finishes below (independent of the accumulator: before the barrier)
    const int zt = zp - H;
    const bool fin = zt >= za && zt < zb;
    float gg[MAXF][3];
    if (fin && (SELF || GG)) {
#pragma unroll
      for (int k = 0; k < MAXF; ++k) {
        float q[3] = {fg[k][0], fg[k][1], fg[k][2]};
        bool pass[3] = {true, true, true};
        if (clamp_grid) {
#pragma unroll
          for (int a = 0; a < 3; ++a) { pass[a] = q[a] >= -1.f && q[a] <= 1.f; q[a] = clamp_unit(q[a]); }
        }
        Taps<3, PAD> t;
        t.build(q[0], q[1], q[2], d);
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) if (!SL) coord_path_diff<3, PAD>(inn + (int64_t)c * V, fgo[k][c], t, d, ax, ay, az);
        if (SL) {
          const int so = (min(max(zt, 0), d.s0 - 1) * d.s1 + min(y0 + wave + k * NWV, d.s1 - 1)) * d.s2 + xl;   // the own sample
#pragma unroll
          for (int c = 0; c < CT; ++c) coord_path_diff<3, PAD>(inn + (int64_t)c * V, gon_all[(int64_t)c * V + so], t, d, ax, ay, az);
        }
        gg[k][0] = pass[0] ? t.x.mult * ax : 0.f;
        gg[k][1] = pass[1] ? t.y.mult * ay : 0.f;
        gg[k][2] = pass[2] ? t.z.mult * az : 0.f;
      }
    }
    __syncthreads();
    // ---- output plane zp-H has seen every sample that can reach it: convert, add / store the coordinate path, store.
    // With a ring of 2H+3 planes (one more than the reach of a step) the deposits of the NEXT step cannot touch the
    // plane being emptied here: one barrier per step.  With 2H+2 (when LDS is short) a second one closes the step.
    if (fin) {
      const int slot = ((zt % NS) + NS) % NS;
#pragma unroll
      for (int k = 0; k < MAXF; ++k) {
        const int row = wave + k * NWV;
        const int uy = y0 + row;
        if (row >= TY || uy >= d.s1) continue;                       // wave-uniform
        int* cell = acc + slot * plane_cells + row * 64 + lane;
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          v[c] = (float)cell[c * TY * 64] * inv;
          cell[c * TY * 64] = 0;
        }
        const int s = (zt * d.s1 + uy) * d.s2 + xl;
        if (SELF) {
#pragma unroll
          for (int c = 0; c < C; ++c) v[c] += SL ? (c0 == 0 ? gg[k][0] : (c0 == 1 ? gg[k][1] : gg[k][2])) : gg[k][c < 3 ? c : 0];
        } else if (GG && xown) {
          float* gq = ggrid + (int64_t)n * 3 * V + s;
#pragma unroll
          for (int a = 0; a < 3; ++a) gq[(int64_t)a * V] = gg[k][a];
        }
        if (xown) {
#pragma unroll
          for (int c = 0; c < C; ++c) ginn[(int64_t)c * V + s] = v[c];
        }
      }
    }
    if (NS == 2 * H + 2) __syncthreads();
    if constexpr (PF) {
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = g1[k][a];
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = go1[k][c];
    }
#pragma unroll
    for (int k = 0; k < MAXF; ++k) {
#pragma unroll
      for (int a = 0; a < 3; ++a) fg[k][a] = fg1[k][a];
#pragma unroll
      for (int c = 0; c < C; ++c) fgo[k][c] = fgo1[k][c];
    }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2D: the same owner-computes scatter without a march.  A workgroup owns TY whole rows of one image (an accumulator of
// C x TY x W 32-bit cells in LDS), visits the sample rows y0-H .. y0+TY+H-1 in 64-lane segments and keeps what lands in
// its rows; then every output leaves with a plain store, the coordinate path of its own sample added (self-composition)
// or stored (grad_grid).  Whole rows: the halo work is in y only, (TY+2H)/TY (the 2D tiles of scatter_tiled.hip pay it
// on both axes and need an overflow list; the window scatter pays 1.7 global float atomics per sample and channel).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rows2d_fix_scale(int H) {   // (2H+2)^2 deposits of weight <= 1 stay below 2^31
  return H <= 2 ? 33554432.f : (H <= 4 ? 16777216.f : (H <= 8 ? 4194304.f : 1048576.f));
}

template <int PAD, int C, bool SELF, bool GG>
__global__ void __launch_bounds__(512)
k_scatter_rows2d(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                 float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int TY, int H, int clamp_grid,
                 int32_t* __restrict__ ws) {
  extern __shared__ int acc[];                   // [C][TY][W]
  constexpr int NWV = 8;
  const int W = d.s2, V = d.s1 * d.s2;
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y0 = blockIdx.x * TY, yend = min(y0 + TY, d.s1);
  const int nseg = (W + 63) >> 6;
  const float* gn = grid + (int64_t)n * 2 * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
  const int ya = max(y0 - H, 0), yb = min(y0 + TY + H, d.s1);
  __shared__ float wmax[NWV];
  {
    const float* rowmax = reinterpret_cast<const float*>(ws + 4) + (int64_t)n * d.s1;
    float m = 0.f;
    for (int y = ya + (int)threadIdx.x; y < yb; y += NWV * 64) m = fmaxf(m, rowmax[y]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) wmax[wave] = m;
  }
  for (int i = threadIdx.x; i < C * TY * W; i += NWV * 64) acc[i] = 0;
  if (SELF && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[3] = -1;   // no max|result| from this launch
  __syncthreads();
  float gmax = wmax[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, wmax[w]);
  const float fix = rows2d_fix_scale(H);
  const float scale = gmax > 0.f ? fix / gmax : 0.f, inv = gmax / fix;

  // ---- deposits: (row, segment) items, two per wave and round with their loads issued together
  constexpr int U = 2;
  const int items = (yb - ya) * nseg;
  for (int it0 = wave * U; it0 < items; it0 += NWV * U) {
    float g[U][2], go[U][C];
    int xs[U], ys[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(it0 + u, items - 1);
      const int r = it / nseg;
      ys[u] = ya + r;
      xs[u] = (it - r * nseg) * 64 + lane;
      const int s = ys[u] * W + min(xs[u], W - 1);
      g[u][0] = gn[s];
      g[u][1] = gn[V + s];
#pragma unroll
      for (int c = 0; c < C; ++c) go[u][c] = gon[(int64_t)c * V + s];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (it0 + u >= items) continue;                                 // wave-uniform
      const bool xin = xs[u] < W;
      if (clamp_grid) { g[u][0] = clamp_unit(g[u][0]); g[u][1] = clamp_unit(g[u][1]); }
      Taps<2, PAD> t;
      t.y = make_tap<PAD>(g[u][1], d.s1);
      if (__ballot(xin && t.y.i0 + 1 >= y0 && t.y.i0 < yend) == 0) continue;
      t.x = make_tap<PAD>(g[u][0], d.s2);
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        const int py = t.y.i0 + cy;
        const bool oky = xin && py >= y0 && py < yend && (cy ? t.y.v1 : t.y.v0);
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          if (!(oky && (cx ? t.x.v1 : t.x.v0))) continue;
          const float wsc = (cx ? t.x.w1 : t.x.w0) * (cy ? t.y.w1 : t.y.w0) * scale;
          int* cell = acc + (py - y0) * W + t.x.i0 + cx;
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(cell + c * TY * W, __float2int_rn(wsc * go[u][c]));
        }
      }
    }
  }
  // ---- the coordinate path of the own samples needs nothing from the accumulator: before the barrier
  const int oitems = (yend - y0) * nseg;
  constexpr int MAXO = 8;                         // own items per wave: TY * nseg / 8 <= 32 * 4 / 8... capped by the launcher
  float ggv[MAXO][2];
  if (SELF || GG) {
#pragma unroll
    for (int k = 0; k < MAXO; ++k) {
      const int it = wave + k * NWV;
      ggv[k][0] = ggv[k][1] = 0.f;
      if (it >= oitems) continue;                                     // wave-uniform
      const int r = it / nseg;
      const int x = (it - r * nseg) * 64 + lane;
      const int s = (y0 + r) * W + min(x, W - 1);
      float q[2] = {gn[s], gn[V + s]};
      float o[C];
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = gon[(int64_t)c * V + s];
      bool pass[2] = {true, true};
      if (clamp_grid) {
#pragma unroll
        for (int a = 0; a < 2; ++a) { pass[a] = q[a] >= -1.f && q[a] <= 1.f; q[a] = clamp_unit(q[a]); }
      }
      Taps<2, PAD> t;
      t.build(q[0], q[1], 0.f, d);
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) coord_path_diff<2, PAD>(inn + (int64_t)c * V, o[c], t, d, ax, ay, az);
      ggv[k][0] = pass[0] ? t.x.mult * ax : 0.f;
      ggv[k][1] = pass[1] ? t.y.mult * ay : 0.f;
    }
  }
  __syncthreads();
  // ---- every owned output: convert, add / store the coordinate path, plain stores
#pragma unroll
  for (int k = 0; k < MAXO; ++k) {
    const int it = wave + k * NWV;
    if (it >= oitems) continue;                                       // wave-uniform
    const int r = it / nseg;
    const int x = (it - r * nseg) * 64 + lane;
    if (x >= W) continue;
    const int s = (y0 + r) * W + x;
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = (float)acc[(c * TY + r) * W + x] * inv;
    if (SELF) {
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] += ggv[k][c < 2 ? c : 0];
    } else if (GG) {
      float* gq = ggrid + (int64_t)n * 2 * V + s;
      gq[0] = ggv[k][0];
      gq[V] = ggv[k][1];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) ginn[(int64_t)c * V + s] = v[c];
  }
}

}  // namespace advchain

using namespace advchain;

// 2D, exact bound of H = 2..16 pixels.  ADVCHAIN_ERR_UNSUPPORTED: use the gather form / the window scatter.
int advchain_scatter_rows2d_launch(bool self, const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                   int64_t N, int64_t C, Dims d, int padding, int clamp_grid, int H, int32_t* workspace,
                                   hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_SCATTER_ROWS2D") != nullptr;   // A/B knob
  static const int hmin = getenv("ADVCHAIN_SCATTER_ROWS2D_HMIN") ? atoi(getenv("ADVCHAIN_SCATTER_ROWS2D_HMIN")) : 3;   // tuning knob
  if (off || !workspace || !gin || padding == PAD_REFLECTION || H < hmin || H > 16 || d.s0 != 1) return ADVCHAIN_ERR_UNSUPPORTED;
  if (d.s2 < 16 || d.s2 > 512 || d.voxels() * 4 >= (1ll << 31)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (self ? C != 2 : (C != 1 && C != 4)) return ADVCHAIN_ERR_UNSUPPORTED;
  static const int ty_forced = getenv("ADVCHAIN_SCATTER_ROWS2D_TY") ? atoi(getenv("ADVCHAIN_SCATTER_ROWS2D_TY")) : 0;
  const int nseg = (d.s2 + 63) / 64;
  int TY = H >= 8 ? 32 : 16;
  if (ty_forced > 0) TY = ty_forced;
  static const size_t lds_cap = getenv("ADVCHAIN_SCATTER_ROWS2D_LDS") ? (size_t)atoi(getenv("ADVCHAIN_SCATTER_ROWS2D_LDS")) : 49152;   // tuning knob
  while (TY > 4 && ((size_t)C * TY * d.s2 * 4 > lds_cap || TY * nseg > 64)) TY >>= 1;   // 48 KiB of cells, 8 own items a wave
  const size_t lds = (size_t)C * TY * d.s2 * sizeof(int);
  if (lds > 65536 - 64 || TY * nseg > 64) return ADVCHAIN_ERR_UNSUPPORTED;
  {
    dim3 rg((unsigned)((d.s1 + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    float* rowmax = reinterpret_cast<float*>(workspace + 4);       // the overflow list of the tiled kernels: unused here
    if (C == 1) hipLaunchKernelGGL(k_march_rowmax<1>, rg, dim3(kBlock), 0, st, gout, rowmax, d, d.s1);
    else if (C == 2) hipLaunchKernelGGL(k_march_rowmax<2>, rg, dim3(kBlock), 0, st, gout, rowmax, d, d.s1);
    else hipLaunchKernelGGL(k_march_rowmax<4>, rg, dim3(kBlock), 0, st, gout, rowmax, d, d.s1);
  }
  dim3 g((unsigned)((d.s1 + TY - 1) / TY), (unsigned)N), b(512);
  const bool gg = ggrid != nullptr;
#define GO(PAD_, C_, SELF_, GG_) \
  hipLaunchKernelGGL((k_scatter_rows2d<PAD_, C_, SELF_, GG_>), g, b, lds, st, gout, in, grid, gin, ggrid, d, TY, H, clamp_grid, workspace)
#define GO_PAD(C_, GG_) do { if (padding == PAD_BORDER) GO(PAD_BORDER, C_, false, GG_); else GO(PAD_ZEROS, C_, false, GG_); } while (0)
  if (self) GO(PAD_BORDER, 2, true, false);
  else if (C == 1) { if (gg) GO_PAD(1, true); else GO_PAD(1, false); }
  else { if (gg) GO_PAD(4, true); else GO_PAD(4, false); }
#undef GO_PAD
#undef GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// Exact bound of H = 2..4 voxels, 3D, rows of at most 64 voxels.  ADVCHAIN_ERR_UNSUPPORTED: use the window scatter.
int advchain_scatter_march_launch(bool self, const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                  int64_t N, int64_t C, Dims d, int padding, int clamp_grid, int H, int32_t* workspace,
                                  hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_SCATTER_MARCH") != nullptr;   // A/B knob
  static const int hmax = getenv("ADVCHAIN_SCATTER_MARCH_HMAX") ? atoi(getenv("ADVCHAIN_SCATTER_MARCH_HMAX")) : 8;   // tuning knob
  if (off || !workspace || !gin || padding == PAD_REFLECTION || H < 2 || H > hmax || H > 8) return ADVCHAIN_ERR_UNSUPPORTED;
  if (d.s2 > 1024 || d.s2 < 8 || d.s0 < 2 || d.voxels() * 4 >= (1ll << 31)) return ADVCHAIN_ERR_UNSUPPORTED;
  const int nseg = d.s2 <= 64 ? 1 : (int)((d.s2 + (64 - 2 * H) - 1) / (64 - 2 * H));
  if (self ? C != 3 : (C != 1 && C != 4)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (H > 4 && C == 1 && N * d.voxels() < (6ll << 20)) return ADVCHAIN_ERR_UNSUPPORTED;   // (4 x 128 x 128 x 64, C = 1: 166 us against 139 for the window scatter; twice that batch: 231 against 282)
  if (H > 4) {
    // bounds of 5..8 voxels: one launch per channel (a ring of 2H+3 planes of all channels would leave 2-4 owned rows and
    // a 5-9x y halo).  Against the window scatter's 5.8 global float atomics per sample (C = 4, 8 x 128 x 128 x 64:
    // 1.45 ms) four passes over the samples are still the cheaper way.
    const int NSb = 2 * H + 3, TYb = 8;
    const size_t ldsb = (size_t)NSb * TYb * 64 * sizeof(int);
    const int n1b = (d.s1 + TYb - 1) / TYb;
    int zcb = d.s0;
    while (zcb > 32 && N * n1b * nseg * ((d.s0 + zcb - 1) / zcb) < 512) zcb = (zcb + 1) / 2;
    const int n0b = (d.s0 + zcb - 1) / zcb;
    dim3 gb((unsigned)(n1b * n0b * nseg), (unsigned)N), bb(512);
    const int rows = (int)(d.s0 * d.s1);
    dim3 rg((unsigned)((rows + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    float* rowmax = reinterpret_cast<float*>(workspace + 4);
    const bool ggb = ggrid != nullptr;
#define GOB(PAD_, SELF_, GG_, CT_) \
    hipLaunchKernelGGL((k_scatter_march3d<PAD_, 1, SELF_, GG_, 8, CT_, true>), gb, bb, ldsb, st, gout, in, grid, gin, ggrid, d, n1b, zcb, TYb, H, NSb, clamp_grid, workspace, nseg, c0)
#define GOB_PAD(GG_, CT_) do { if (padding == PAD_BORDER) GOB(PAD_BORDER, false, GG_, CT_); else GOB(PAD_ZEROS, false, GG_, CT_); } while (0)
    for (int c0 = 0; c0 < (int)C; ++c0) {
      hipLaunchKernelGGL(k_march_rowmax<1>, rg, dim3(kBlock), 0, st, gout + (int64_t)c0 * d.voxels(), rowmax, d, rows, (int)C);
      if (self) GOB(PAD_BORDER, true, false, 3);
      else if (C == 1) { if (ggb) GOB_PAD(true, 1); else GOB_PAD(false, 1); }
      else if (ggb && c0 == 0) GOB_PAD(true, 4);
      else GOB_PAD(false, 4);
    }
#undef GOB_PAD
#undef GOB
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  // rows per workgroup: as many as 60 KiB of accumulator planes allow, at most 8
  static const int ty_forced = getenv("ADVCHAIN_SCATTER_MARCH_TY") ? atoi(getenv("ADVCHAIN_SCATTER_MARCH_TY")) : 0;
  int NS = 2 * H + 3;
  int TY = 8;    // (16 halves the y halo work but leaves 512 workgroups at 4 x 128 x 128 x 64)
  if ((size_t)NS * C * TY * 64 * 4 > 65536) NS = 2 * H + 2;
  while (TY > 4 && (size_t)NS * C * TY * 64 * 4 > 65536) TY >>= 1;
  if (ty_forced == 4 || ty_forced == 8) TY = ty_forced;
  const size_t lds = (size_t)NS * C * TY * 64 * sizeof(int);
  if (lds > 65536) return ADVCHAIN_ERR_UNSUPPORTED;
  const int n1 = (d.s1 + TY - 1) / TY;
  static const int zc_forced = getenv("ADVCHAIN_SCATTER_MARCH_ZC") ? atoi(getenv("ADVCHAIN_SCATTER_MARCH_ZC")) : 0;
  int zc = d.s0;
  while (zc > 8 && N * n1 * nseg * ((d.s0 + zc - 1) / zc) < 512) zc = (zc + 1) / 2;   // (16 planes: 1003 GB/s, 8: 942, 32: 834)
  if (zc_forced > 0) zc = zc_forced;
  const int n0 = (d.s0 + zc - 1) / zc;
  {
    const int rows = (int)(d.s0 * d.s1);
    dim3 rg((unsigned)((rows + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    float* rowmax = reinterpret_cast<float*>(workspace + 4);       // the overflow list of the tiled kernels: unused here
    if (C == 1) hipLaunchKernelGGL(k_march_rowmax<1>, rg, dim3(kBlock), 0, st, gout, rowmax, d, rows);
    else if (C == 3) hipLaunchKernelGGL(k_march_rowmax<3>, rg, dim3(kBlock), 0, st, gout, rowmax, d, rows);
    else hipLaunchKernelGGL(k_march_rowmax<4>, rg, dim3(kBlock), 0, st, gout, rowmax, d, rows);
  }
  const int nwv = TY > 4 ? 8 : 4;
  dim3 g((unsigned)(n1 * n0 * nseg), (unsigned)N), b(nwv * 64);
  const bool gg = ggrid != nullptr;
#define GO(PAD_, C_, SELF_, GG_) \
  do { if (nwv == 8) hipLaunchKernelGGL((k_scatter_march3d<PAD_, C_, SELF_, GG_, 8>), g, b, lds, st, gout, in, grid, gin, ggrid, d, n1, zc, TY, H, NS, clamp_grid, workspace, nseg, 0); \
  else hipLaunchKernelGGL((k_scatter_march3d<PAD_, C_, SELF_, GG_, 4>), g, b, lds, st, gout, in, grid, gin, ggrid, d, n1, zc, TY, H, NS, clamp_grid, workspace, nseg, 0); } while (0)
#define GO_PAD(C_, GG_) do { if (padding == PAD_BORDER) GO(PAD_BORDER, C_, false, GG_); else GO(PAD_ZEROS, C_, false, GG_); } while (0)
  if (self) GO(PAD_BORDER, 3, true, false);
  else if (C == 1) { if (gg) GO_PAD(1, true); else GO_PAD(1, false); }
  else { if (gg) GO_PAD(4, true); else GO_PAD(4, false); }
#undef GO_PAD
#undef GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
