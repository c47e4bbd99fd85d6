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

// 32-bit byte offsets from a wave-uniform base: the load takes the base from SGPRs (global_load v, v_off, s[base]) instead of
// a 64-bit address built in VGPRs per load (39 v_lshl_add_u64 in the C = 1 kernel).  The launchers bound a volume by 2^31 bytes.
__device__ __forceinline__ float ld_off(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st_off(float* __restrict__ base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// d(sample)/d(unnormalised coordinate) * go in difference form: ax = go * sum_zy wz wy (v[z][y][1] - v[z][y][0]) and so on
// -- a third of the instructions of the signed-product form of sample_linear_bwd (these kernels are VALU bound), the same
// value up to the rounding of the differences.
// (the arithmetic on corner values that are already in registers: a caller with several channels / samples loads all their
// corners first -- CornerOffsets::load -- so that their round trips overlap)
template <int DIM, int PAD>
__device__ __forceinline__ void coord_path_values(const float (&vl)[8], float go, const Taps<DIM, PAD>& t, float& ax, float& ay,
                                                  float& az);

template <int DIM, int PAD>
__device__ __forceinline__ void coord_path_diff(const float* __restrict__ in, float go, const Taps<DIM, PAD>& t, const Dims& d,
                                                float& ax, float& ay, float& az) {
  const CornerOffsets<DIM, PAD> o(t, d);
  float vl[8];
  o.load(in, vl);
  coord_path_values<DIM, PAD>(vl, go, t, ax, ay, az);
}

template <int DIM, int PAD>
__device__ __forceinline__ void coord_path_values(const float (&vl)[8], float go, const Taps<DIM, PAD>& t, float& ax, float& ay,
                                                  float& az) {
  float v[2][2][2];
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) v[cz][cy][cx] = vl[(cz * 2 + cy) * 2 + cx];
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

// The same for rows of exactly 64 voxels, 16 bytes per lane: 16 lanes a row, 4 rows per load instruction, 16 rows a wave
// in flight (7.1 -> 4 us at 4 x 1 x 128 x 128 x 64: the dword form is a quarter of the bytes per vector-memory instruction).
template <int C>
__global__ void __launch_bounds__(kBlock) k_march_rowmax64(const float* __restrict__ x, float* __restrict__ rowmax, int rows_per_n, int ctot) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * (4 * U) + (lane >> 4);
  const int n = blockIdx.y;
  if (row0 - (lane >> 4) >= rows_per_n) return;
  const int64_t V = (int64_t)rows_per_n * 64;
  const float* p = x + (int64_t)n * ctot * V + (lane & 15) * 4;
  float4 v[U][C];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int row = min(row0 + 4 * u, rows_per_n - 1);
#pragma unroll
    for (int c = 0; c < C; ++c) v[u][c] = *reinterpret_cast<const float4*>(p + (int64_t)c * V + (int64_t)row * 64);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float m = 0.f;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float q = fmaxf(fmaxf(fabsf(v[u][c].x), fabsf(v[u][c].y)), fmaxf(fabsf(v[u][c].z), fabsf(v[u][c].w)));
      m = fmaxf(m, q);
      bad = bad || !(fabsf(v[u][c].x) <= 3.0e38f) || !(fabsf(v[u][c].y) <= 3.0e38f) || !(fabsf(v[u][c].z) <= 3.0e38f) ||
            !(fabsf(v[u][c].w) <= 3.0e38f);
    }
    if (bad) m = __int_as_float(0x7f800000);                       // (fmaxf keeps +inf through the reduction below)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int row = row0 + 4 * u;
    if ((lane & 15) == 0 && row < rows_per_n) rowmax[(int64_t)n * rows_per_n + row] = m;
  }
}

// launch of the row maxima (either form)
template <int C>
static void launch_rowmax(const float* x, float* rowmax, Dims d, int rows, int64_t N, int ctot, hipStream_t st) {
  if (d.s2 == 64 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int per_block = (kBlock / 64) * 16;
    hipLaunchKernelGGL(k_march_rowmax64<C>, dim3((unsigned)((rows + per_block - 1) / per_block), (unsigned)N), dim3(kBlock), 0, st, x, rowmax, rows, ctot);
  } else {
    hipLaunchKernelGGL(k_march_rowmax<C>, dim3((unsigned)((rows + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N), dim3(kBlock), 0, st, x, rowmax, d, rows, ctot);
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
  // grad_grid of an image warp: the coordinate path of a workgroup's OWN samples is evaluated when their row is visited for
  // its deposits (the taps are built once, no second set of loads), and leaves at once -- it needs nothing from the accumulator
  constexpr bool EARLY = GG && !SELF && !SL;
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
      const unsigned s = (unsigned)((zq * d.s1 + ys) * d.s2 + xl) * 4u;
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = ld_off(gn + (int64_t)a * V, s);
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = ld_off(gon + (int64_t)c * V, s);
    }
  };
  auto load_own = [&](int zt, float (&g)[MAXF][3], float (&go)[MAXF][C]) {
    if (!(SELF || GG) || EARLY) return;
    const int zq = min(max(zt, 0), d.s0 - 1);
#pragma unroll
    for (int k = 0; k < MAXF; ++k) {
      const int uy = min(y0 + wave + k * NWV, d.s1 - 1);
      const unsigned s = (unsigned)((zq * d.s1 + uy) * d.s2 + xl) * 4u;
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = ld_off(gn + (int64_t)a * V, s);
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = ld_off(gon + (int64_t)c * V, s);
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
        const bool own = EARLY && ys >= y0 && ys < yend && zp >= za && zp < zb;   // wave-uniform
        bool pass[3] = {true, true, true};
        if (clamp_grid) {
#pragma unroll
          for (int a = 0; a < 3; ++a) { pass[a] = g[k][a] >= -1.f && g[k][a] <= 1.f; g[k][a] = clamp_unit(g[k][a]); }
        }
        // The kernel is VALU bound (4 waves a SIMD, each 22% of its cycles in VALU issue) and most visits of halo rows
        // and halo planes deposit nothing here: the y and z taps alone decide that, for the whole wave.
        Taps<3, PAD> t;
        t.y = make_tap<PAD>(g[k][1], d.s1);
        t.z = make_tap<PAD>(g[k][2], d.s0);
        const bool reach = xin && t.y.i0 + 1 >= y0 && t.y.i0 < yend && t.z.i0 + 1 >= max(za, zp - H) && t.z.i0 < min(zb, zp + H + 2);
        const bool any = __ballot(reach) != 0;
        if (!any && !own) continue;
        t.x = make_tap<PAD>(g[k][0], d.s2);
        if (any) {
        // Per-axis masked weights (a corner outside this workgroup's columns / rows / planes, or outside the ring's reach
        // of this step, gets weight 0; what is owned lies inside the volume, so the corner's own validity is implied) and
        // per-axis clamped cell coordinates: the eight deposits need no per-corner branch (the branchy form spent ~20
        // instructions and two exec-mask regions per corner).  One wave-uniform test per (z, y) corner pair remains: half
        // of the visits are halo rows that reach the tile with one of their two rows / planes only.
        float wxm[2], wym[2], wzm[2];
        int colx[2], rowy[2], slotz[2];
        bool anyy[2], anyz[2];
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const int pxx = t.x.i0 + cx;
          // (a masked lane deposits its zero into its OWN column: clamped to the row ends, the lanes beyond a short row or
          // in the halo columns of an x segment all hit one cell, and same-address LDS atomics serialise -- rows of 80
          // voxels ran 2.4x slower)
          const bool okx = xin && pxx >= xo0 && pxx < xo1;
          wxm[cx] = okx ? t.wx(cx) : 0.f;
          colx[cx] = okx ? pxx - xbase : lane;
        }
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          const int py = t.y.i0 + cy;
          anyy[cy] = py >= y0 && py < yend;
          wym[cy] = anyy[cy] ? t.wy(cy) : 0.f;
          rowy[cy] = min(max(py - y0, 0), TY - 1) * 64;
        }
        const int zlo = max(za, zp - H), zhi = min(zb, zp + H + 2);
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
          const int pz = t.z.i0 + cz;
          anyz[cz] = pz >= zlo && pz < zhi;
          wzm[cz] = anyz[cz] ? t.wz(cz) : 0.f;
          int slot = sz + min(max(pz - zp, -H), H + 1);
          slot += slot < 0 ? NS : 0;
          slot -= slot >= NS ? NS : 0;
          slotz[cz] = slot * plane_cells;
        }
        float a[C][2];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float gsc = go[k][c] * scale;
          a[c][0] = wxm[0] * gsc;
          a[c][1] = wxm[1] * gsc;
        }
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            if (__ballot(anyz[cz] && anyy[cy]) == 0) continue;       // wave-uniform
            const float wzy = wzm[cz] * wym[cy];
            int* cell = acc + slotz[cz] + rowy[cy];
#pragma unroll
            for (int cx = 0; cx < 2; ++cx)
#pragma unroll
              for (int c = 0; c < C; ++c) atomicAdd(cell + colx[cx] + c * TY * 64, fix_round(wzy * a[c][cx]));
          }
        }
        if (own) {
          float ax = 0.f, ay = 0.f, az = 0.f;
          {   // the corners of every channel first, then the arithmetic: one round trip instead of one per channel
            const CornerOffsets<3, PAD> o(t, d);
            float vl[C][8];
#pragma unroll
            for (int c = 0; c < C; ++c) o.load(inn + (int64_t)c * V, vl[c]);
#pragma unroll
            for (int c = 0; c < C; ++c) coord_path_values<3, PAD>(vl[c], go[k][c], t, ax, ay, az);
          }
          if (xown) {
            float* gq = ggrid + (int64_t)n * 3 * V;
            const unsigned so = (unsigned)((zp * d.s1 + ys) * d.s2 + xl) * 4u;
            st_off(gq, so, pass[0] ? t.x.mult * ax : 0.f);
            st_off(gq + V, so, pass[1] ? t.y.mult * ay : 0.f);
            st_off(gq + (int64_t)2 * V, so, pass[2] ? t.z.mult * az : 0.f);
          }
        }
      }
    }
    // ---- coordinate path of the rows this wave finishes below (independent of the accumulator: before the barrier)
    const int zt = zp - H;
    const bool fin = zt >= za && zt < zb;
    float gg[MAXF][3];
    if (fin && (SELF || GG) && !EARLY) {
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
        if (!SL) {   // the corners of every channel first, then the arithmetic: one round trip instead of one per channel
          const CornerOffsets<3, PAD> o(t, d);
          float vl[C][8];
#pragma unroll
          for (int c = 0; c < C; ++c) o.load(inn + (int64_t)c * V, vl[c]);
#pragma unroll
          for (int c = 0; c < C; ++c) coord_path_values<3, PAD>(vl[c], fgo[k][c], t, ax, ay, az);
        }
        if (SL) {
          const int so = (min(max(zt, 0), d.s0 - 1) * d.s1 + min(y0 + wave + k * NWV, d.s1 - 1)) * d.s2 + xl;   // the own sample
#pragma unroll
          for (int c = 0; c < CT; ++c) coord_path_diff<3, PAD>(inn + (int64_t)c * V, ld_off(gon_all + (int64_t)c * V, (unsigned)so * 4u), t, d, ax, ay, az);
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
        const unsigned s = (unsigned)((zt * d.s1 + uy) * d.s2 + xl) * 4u;
        if (SELF) {
#pragma unroll
          for (int c = 0; c < C; ++c) v[c] += SL ? (c0 == 0 ? gg[k][0] : (c0 == 1 ? gg[k][1] : gg[k][2])) : gg[k][c < 3 ? c : 0];
        } else if (GG && !EARLY && xown) {
          float* gq = ggrid + (int64_t)n * 3 * V;
#pragma unroll
          for (int a = 0; a < 3; ++a) st_off(gq + (int64_t)a * V, s, gg[k][a]);
        }
        if (xown) {
#pragma unroll
          for (int c = 0; c < C; ++c) st_off(ginn + (int64_t)c * V, s, v[c]);
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
// The self-composition on rows of 68 .. 80 voxels (cfg-5: 160 x 160 x 80) with lane <-> FLAT sample.  k_scatter_march3d cuts
// such rows into x segments of 64 - 2H owned lanes: two 64-lane visits per sample row, the second one a quarter full, and
// the halo lanes of a segment visited twice.  Nothing in a deposit ties a lane to an x -- the cell is an address the lane
// computes -- so here the accumulator rows are W cells wide (no x halo at all), the (TY + 2H) x W samples of a plane are cut
// into items of 64 consecutive samples (whole rows are contiguous in memory: an item is one coalesced run) and wave w takes
// items w, w + NWV, ...; the TY x W outputs of the finished plane leave the same way.  Arithmetic, fixed-point scale and
// summation are those of k_scatter_march3d<PAD_BORDER, 3, true, false, 8>.
// ---------------------------------------------------------------------------------------------------------------------
template <int NWV, int MAXR>
__global__ void __launch_bounds__(NWV * 64)
k_scatter_march3d_flat(const float* __restrict__ gout, const float* __restrict__ phi, float* __restrict__ gin, Dims d, int n1,
                       int zc, int TY, int H, int NS, int32_t* __restrict__ ws) {
  constexpr int C = 3, PAD = PAD_BORDER;
  constexpr int MAXF = 2;                        // output items per wave and step: TY * W <= 2 * NWV * 64
  extern __shared__ int acc[];                   // [slot NS][C][TY][W]
  const int V = (int)d.voxels(), W = d.s2;
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ty = blockIdx.x % n1, tz = blockIdx.x / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const int chan_cells = TY * W, plane_cells = C * chan_cells;
  const float* gn = phi + (int64_t)n * 3 * V;
  const float* gon = gout + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
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
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[3] = -1;   // no max|result| from this launch
  const int yend = min(y0 + TY, d.s1);
  const int nrows = TY + 2 * H;
  const int nsamp = nrows * W, nitems = (nsamp + 63) >> 6;
  const int nout = TY * W, oitems = (nout + 63) >> 6;
  const unsigned plane_bytes = (unsigned)(d.s1 * W) * 4u;

  // ---- the samples / outputs of this lane (the same in every step)
  int sx[MAXR];
  bool sval[MAXR];
  unsigned soff[MAXR];                           // byte offset of the sample within its plane (row clamped into the volume)
#pragma unroll
  for (int k = 0; k < MAXR; ++k) {
    const int f = (wave + k * NWV) * 64 + lane;
    const int fcl = min(f, nsamp - 1);
    const int r = fcl / W;
    const int ys = y0 - H + r;
    sval[k] = f < nsamp && ys >= 0 && ys < d.s1;
    // (a lane past the last sample deposits its zeros into a column of its own: same-address LDS atomics serialise)
    sx[k] = f < nsamp ? fcl - r * W : lane;
    soff[k] = (unsigned)(min(max(ys, 0), d.s1 - 1) * W + sx[k]) * 4u;
  }
  int ocell[MAXF];
  bool oval[MAXF], ocells[MAXF];
  unsigned ooff[MAXF];
#pragma unroll
  for (int k = 0; k < MAXF; ++k) {
    const int f = (wave + k * NWV) * 64 + lane;
    const int fcl = min(f, nout - 1);
    const int r = fcl / W, x = fcl - r * W;
    ocells[k] = f < nout;
    oval[k] = f < nout && y0 + r < d.s1;
    ocell[k] = r * W + x;
    ooff[k] = (unsigned)(min(y0 + r, d.s1 - 1) * W + x) * 4u;
  }

  auto load_samples = [&](int zp, float (&g)[MAXR][3], float (&go)[MAXR][C]) {
    const unsigned base = (unsigned)min(max(zp, 0), d.s0 - 1) * plane_bytes;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      if (wave + k * NWV >= nitems) continue;                        // wave-uniform
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = ld_off(gn + (int64_t)a * V, base + soff[k]);
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = ld_off(gon + (int64_t)c * V, base + soff[k]);
    }
  };
  auto load_own = [&](int zt, float (&g)[MAXF][3], float (&go)[MAXF][C]) {
    const unsigned base = (unsigned)min(max(zt, 0), d.s0 - 1) * plane_bytes;
#pragma unroll
    for (int k = 0; k < MAXF; ++k) {
      if (wave + k * NWV >= oitems) continue;                        // wave-uniform
#pragma unroll
      for (int a = 0; a < 3; ++a) g[k][a] = ld_off(gn + (int64_t)a * V, base + ooff[k]);
#pragma unroll
      for (int c = 0; c < C; ++c) go[k][c] = ld_off(gon + (int64_t)c * V, base + ooff[k]);
    }
  };
  float g[MAXR][3], go[MAXR][C], fg[MAXF][3], fgo[MAXF][C];
  load_samples(za - H, g, go);
  load_own(za - 2 * H, fg, fgo);
  for (int zp = za - H; zp < zb + H; ++zp) {
    float g1[MAXR][3], go1[MAXR][C], fg1[MAXF][3], fgo1[MAXF][C];
    load_samples(zp + 1, g1, go1);
    load_own(zp + 1 - H, fg1, fgo1);
    // ---- deposits of sample plane zp: only what lands in the owned rows and planes is kept
    if (zp >= 0 && zp < d.s0) {
      const int sz = ((zp % NS) + NS) % NS;
      const int zlo = max(za, zp - H), zhi = min(zb, zp + H + 2);
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        if (wave + k * NWV >= nitems) continue;                      // wave-uniform
        Taps<3, PAD> t;
        t.y = make_tap<PAD>(g[k][1], d.s1);
        t.z = make_tap<PAD>(g[k][2], d.s0);
        const bool reach = sval[k] && t.y.i0 + 1 >= y0 && t.y.i0 < yend && t.z.i0 + 1 >= zlo && t.z.i0 < zhi;
        if (__ballot(reach) == 0) continue;
        t.x = make_tap<PAD>(g[k][0], d.s2);
        float wxm[2], wym[2], wzm[2];
        int colx[2], rowy[2], slotz[2];
        bool anyy[2], anyz[2];
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const int pxx = t.x.i0 + cx;
          const bool okx = sval[k] && pxx >= 0 && pxx < W;           // (masked lanes: own column, see k_scatter_march3d)
          wxm[cx] = okx ? t.wx(cx) : 0.f;
          colx[cx] = okx ? pxx : sx[k];
        }
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          const int py = t.y.i0 + cy;
          anyy[cy] = py >= y0 && py < yend;
          wym[cy] = anyy[cy] ? t.wy(cy) : 0.f;
          rowy[cy] = min(max(py - y0, 0), TY - 1) * W;
        }
#pragma unroll
        for (int cz = 0; cz < 2; ++cz) {
          const int pz = t.z.i0 + cz;
          anyz[cz] = pz >= zlo && pz < zhi;
          wzm[cz] = anyz[cz] ? t.wz(cz) : 0.f;
          int slot = sz + min(max(pz - zp, -H), H + 1);
          slot += slot < 0 ? NS : 0;
          slot -= slot >= NS ? NS : 0;
          slotz[cz] = slot * plane_cells;
        }
        float a[C][2];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float gsc = go[k][c] * scale;
          a[c][0] = wxm[0] * gsc;
          a[c][1] = wxm[1] * gsc;
        }
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            if (__ballot(anyz[cz] && anyy[cy]) == 0) continue;       // wave-uniform
            const float wzy = wzm[cz] * wym[cy];
            int* cell = acc + slotz[cz] + rowy[cy];
#pragma unroll
            for (int cx = 0; cx < 2; ++cx)
#pragma unroll
              for (int c = 0; c < C; ++c) atomicAdd(cell + colx[cx] + c * chan_cells, fix_round(wzy * a[c][cx]));
          }
      }
    }
    // ---- coordinate path of the outputs this wave finishes below (independent of the accumulator: before the barrier)
    const int zt = zp - H;
    const bool fin = zt >= za && zt < zb;
    float gg[MAXF][3];
    if (fin) {
      // every corner of every channel of an output item is requested before the first one is used: per channel the gathers
      // were a round trip each (load, wait, arithmetic, next channel)
#pragma unroll
      for (int k = 0; k < MAXF; ++k) {
        if (wave + k * NWV >= oitems) continue;                      // wave-uniform
        Taps<3, PAD> tq;
        tq.build(fg[k][0], fg[k][1], fg[k][2], d);
        const CornerOffsets<3, PAD> o(tq, d);
        float vl[C][8];                                              // (both items at once: 147 VGPRs, one workgroup a CU)
#pragma unroll
        for (int c = 0; c < C; ++c) o.load(gn + (int64_t)c * V, vl[c]);
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) coord_path_values<3, PAD>(vl[c], fgo[k][c], tq, ax, ay, az);
        gg[k][0] = tq.x.mult * ax;
        gg[k][1] = tq.y.mult * ay;
        gg[k][2] = tq.z.mult * az;
      }
    }
    __syncthreads();
    // ---- output plane zp-H has seen every sample that can reach it: convert, add the coordinate path, store
    if (fin) {
      const int slot = ((zt % NS) + NS) % NS;
      const unsigned base = (unsigned)zt * plane_bytes;
#pragma unroll
      for (int k = 0; k < MAXF; ++k) {
        if (wave + k * NWV >= oitems) continue;                      // wave-uniform
        if (ocells[k]) {
          int* cell = acc + slot * plane_cells + ocell[k];
          float v[C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            v[c] = (float)cell[c * chan_cells] * inv;
            cell[c * chan_cells] = 0;
          }
          if (oval[k]) {
#pragma unroll
            for (int c = 0; c < C; ++c) st_off(ginn + (int64_t)c * V, base + ooff[k], v[c] + gg[k][c]);
          }
        }
      }
    }
    if (NS == 2 * H + 2) __syncthreads();
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

// ---------------------------------------------------------------------------------------------------------------------
// The same march for an image warp (C == 1, rows of at most 64 voxels) with everything that crosses the memory pipeline
// 16 bytes wide.  k_scatter_march3d is bound by vector-memory ISSUE, not by bytes or VALU (lesson 4: a CU retires one
// vector-memory wave-instruction per ~26 clk whatever it carries): per step and CU it issues 4 dword loads per visited
// sample row (3x the rows it owns at H = 4), 8 dword gathers per own row for the coordinate path and 4 dword stores per
// own row -- 288 instructions x 26 clk x 24 steps = the 82 us measured (4 x 1 x 128 x 128 x 64, H = 4).  Here
//   * a step's sample rows (grid x, y, z and grad_out, rows y0-H .. y0+TY+H-1 of plane zp) are fetched by the whole
//     workgroup with 16-byte loads one step ahead (registers), written to an LDS stage and picked up by the wave that
//     visits the row -- 1 KiB per wave-instruction instead of 256 B;
//   * the coordinate path (grad_grid) of an own sample is evaluated when its row is visited for its deposits (taps built
//     once), its 8 corner values come from an LDS ring of 2H+1 planes x TY+2H rows of `in` (the exact bound H puts every
//     corner inside it), staged with 16-byte loads, zero outside the volume and with a zero column either side: zeros
//     padding is data, no validity selects;
//   * grad_in leaves from the accumulator ring as 16-byte stores (16 lanes a row);
//   * two stages and one spare plane in each ring: ONE barrier per step.
// 16 waves own 16 rows (2.25 visits per sample at H = 4 instead of 3); 156 KiB of LDS, one workgroup a CU.
// ---------------------------------------------------------------------------------------------------------------------
// A/B build switch (tools/ab/build_variant.sh -DADVCHAIN_WIDE_TRIM=0): the trimmed deposit arithmetic of the wide march scatter
#ifndef ADVCHAIN_WIDE_TRIM
#define ADVCHAIN_WIDE_TRIM 1
#endif
template <int H>
struct WideCfg {
  static constexpr int TY = 16, NWV = 16, NT = NWV * 64, XS = 68;
  // accumulator ring: one plane more than a step reaches, so that the plane being emptied and the next step's deposits never
  // meet; `in` ring: one plane more than the own samples reach, so that the plane joining for the next step can be written
  // while slower waves still read; two stages: ONE barrier per step
  static constexpr int NS = 2 * H + 3, NP = 2 * H + 2, R = TY + 2 * H;
  static constexpr int ACC = NS * TY * 64, STAGE = 4 * R * 64, RING = NP * R * XS + 4;
  static constexpr int NI = 4 * R * 16;                      // 16-byte items of a step's sample rows
  static constexpr int KI = (NI + NT - 1) / NT;
  static constexpr size_t lds(bool gg) { return (size_t)(ACC + 2 * STAGE + (gg ? RING : 0)) * 4; }
};

template <int PAD, bool GG, int H>
__global__ void __launch_bounds__(WideCfg<H>::NT)
k_scatter_march3d_wide(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                       float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int zc, int clamp_grid,
                       int32_t* __restrict__ ws) {
  using G = WideCfg<H>;
  constexpr int TY = G::TY, NWV = G::NWV, NT = G::NT, XS = G::XS, NS = G::NS, NP = G::NP, R = G::R;
  extern __shared__ int acc[];                                // [NS][TY][64]
  float* stage = reinterpret_cast<float*>(acc + G::ACC);      // [2][4][R][64]: grid x, y, z, grad_out of a step's rows
  float* ring = stage + 2 * G::STAGE;                         // [NP][R][XS]: `in`, data at columns 4 .. 67; column 68 of a
                                                              // row is column 0 of the next one: zero either way
  const int V = (int)d.voxels();
  // blocks are dealt to the 8 XCDs round-robin: give every XCD a contiguous run of tiles (y fastest, then z, then the batch),
  // so that the halo rows and planes neighbouring workgroups share come from one L2
  const int nb = gridDim.x, ntile = n1 * ((d.s0 + zc - 1) / zc);
  const int tl = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
  const int n = tl / ntile, trem = tl - n * ntile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty = trem % n1, tz = trem / n1;
  const int y0 = ty * TY, yend = min(y0 + TY, d.s1);
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* gn = grid + (int64_t)n * 3 * V;
  const float* gon = gout + (int64_t)n * V;
  const float* inn = in + (int64_t)n * V;
  float* ginn = gin + (int64_t)n * V;
  float* ggn = ggrid + (int64_t)n * 3 * V;
  const bool xin = lane < d.s2;
  const int xl = min(lane, d.s2 - 1);

  auto ring_slot = [&](int z) { return ((z % NP) + NP) % NP; };
  // Requests are unconditional, from addresses clamped into the volume, into registers nothing else writes (a zero-fill
  // followed by a conditional load made the compiler wait for ALL outstanding memory operations before the fill); what lies
  // outside the volume or beyond the row end becomes zero on the way to LDS.
  // one plane of `in` for the ring: rows y0-H .. y0+TY+H-1 (threads 0 .. 16 R - 1)
  const int qx = min(4 * (tid & 15), max(d.s2 - 4, 0));
  auto fetch_in = [&](int z, float4& v) {
    if (!GG) return;
    const int r = min(tid >> 4, R - 1);
    const int y = min(max(y0 - H + r, 0), d.s1 - 1), zq = min(max(z, 0), d.s0 - 1);
    v = *reinterpret_cast<const float4*>(inn + (unsigned)((zq * d.s1 + y) * d.s2 + qx));
  };
  auto commit_in = [&](int z, const float4& v) {
    if (!GG || tid >= R * 16) return;
    const int r = tid >> 4, q = tid & 15, y = y0 - H + r;
    const bool ok = z >= 0 && z < d.s0 && y >= 0 && y < d.s1 && 4 * q < d.s2;
    *reinterpret_cast<float4*>(ring + (ring_slot(z) * R + r) * XS + 4 + 4 * q) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // the sample rows of plane zp (rows outside the volume are never visited)
  auto fetch_rows = [&](int zp, float4 (&v)[G::KI]) {
    const int zq = min(max(zp, 0), d.s0 - 1);
#pragma unroll
    for (int k = 0; k < G::KI; ++k) {
      const int i = min(tid + k * NT, G::NI - 1);
      const int ch = i / (R * 16), rem = i - ch * (R * 16), r = rem >> 4;
      const int yq = min(max(y0 - H + r, 0), d.s1 - 1);
      const float* base = ch < 3 ? gn + (int64_t)ch * V : gon;
      v[k] = *reinterpret_cast<const float4*>(base + (unsigned)((zq * d.s1 + yq) * d.s2 + qx));
    }
  };
  auto commit_rows = [&](float* st, const float4 (&v)[G::KI]) {
#pragma unroll
    for (int k = 0; k < G::KI; ++k) {
      const int i = tid + k * NT;
      if (i < G::NI) *reinterpret_cast<float4*>(st + i * 4) = 4 * (tid & 15) < d.s2 ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // ---- prologue: ring planes za-H .. za (plane zp+H+1 joins during step zp), the first step's rows; requests run TWO
  // steps ahead (a step is about as long as a memory round trip under load: one step ahead every step ended waiting for
  // its own requests), in two register sets that swap roles
  // (all of the prologue's requests are in flight together: one round trip, not H + 3)
  __shared__ float wmax[NWV];
  float4 rowA[G::KI], rowB[G::KI], inA = make_float4(0.f, 0.f, 0.f, 0.f), inB = inA;
  {
    float4 pin[H + 1];
#pragma unroll
    for (int p = 0; p <= H; ++p) fetch_in(za - H + p, pin[p]);
    fetch_rows(za - H, rowB);
    fetch_rows(za - H + 1, rowA);
    fetch_in(za + 1, inA);
    // ... and while they travel: the fixed-point scale from the maximum |grad_out| over the rows this workgroup visits, zeroed
    // rings.  ws == nullptr (round 6): no row-maxima pre-pass was launched -- the workgroup reads its own (TY + 2H) x (zc + 2H)
    // rows of grad_out itself, 16 bytes per lane, four requests in flight per thread (a plane's visited rows are one
    // contiguous run; 2.25x the tensor over all workgroups, out of L2): the separate k_march_rowmax64 launch was 5.1 us of
    // the north-star backward's 67.6
    {
      const int ya = max(y0 - H, 0), yn = min(y0 + TY + H, d.s1) - ya;
      const int zlo = max(za - H, 0), zn = min(zb + H, d.s0) - zlo;
      float m = 0.f;
      if (ws) {
        const float* rowmax = reinterpret_cast<const float*>(ws + 4) + (int64_t)n * d.s0 * d.s1;
        for (int i = tid; i < yn * zn; i += NT) m = fmaxf(m, rowmax[(zlo + i / yn) * d.s1 + ya + i % yn]);
      } else {
        const int per_plane = (yn * d.s2) >> 2, total = per_plane * zn;     // float4 items (S2 % 4 == 0)
        const int pstride = d.s1 * d.s2;
        const float* base = gon + (unsigned)(zlo * pstride + ya * d.s2);
        bool bad = false;
        for (int i0 = tid; i0 < total; i0 += 4 * NT) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * NT, total - 1);
            const int pl = i / per_plane, r4 = i - pl * per_plane;
            v[u] = *reinterpret_cast<const float4*>(base + (unsigned)(pl * pstride + 4 * r4));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float q = fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
            m = fmaxf(m, q);
            bad = bad || !(fabsf(v[u].x) <= 3.0e38f) || !(fabsf(v[u].y) <= 3.0e38f) || !(fabsf(v[u].z) <= 3.0e38f) ||
                  !(fabsf(v[u].w) <= 3.0e38f);
          }
        }
        if (bad) m = __int_as_float(0x7f800000);      // a non-finite gradient must surface (lesson 36): scale 0, factor inf
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane == 0) wmax[wave] = m;
    }
    for (int i = tid; i < G::ACC; i += NT) acc[i] = 0;
    if (GG) for (int i = tid; i < G::RING; i += NT) ring[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int p = 0; p <= H; ++p) commit_in(za - H + p, pin[p]);
    commit_rows(stage, rowB);
  }
  __syncthreads();
  float gmax = wmax[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, wmax[w]);
  const float fix = march_fix_scale(H);
  const float scale = gmax > 0.f ? fix / gmax : 0.f, inv = gmax / fix;

  // Stores never make a wave wait: on gfx9 the source registers of a store stay busy until it is acknowledged (vmcnt), and
  // the compiler puts `s_waitcnt vmcnt(0)` in front of the first instruction that reuses them.  With the grad_grid values
  // stored from where they were computed, and the converted output plane from where it was read, every step waited out a
  // store round trip (42-49 % of the wave-cycles parked).  So: a step's grad_grid values are stored at the TOP of the next
  // step, ahead of its loads; the source registers of both kinds of store are kept alive (an empty asm that reads them)
  // until the wait that the step's LDS commit needs anyway -- by then a whole step has passed.
  float gg_prev[3] = {0.f, 0.f, 0.f};
  unsigned so_prev = 0;
  bool have_prev = false;
  float4 fprev = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned fo_prev = 0;
  int cur = 0;
  // rc / ic: requested a step ago for step zp+1, committed at the end of this one; rl / il: requested now for step zp+2
  auto step = [&](int zp, float4 (&rc)[G::KI], float4& ic, float4 (&rl)[G::KI], float4& il) {
    const float* st = stage + cur * G::STAGE;
    const bool more = zp + 1 < zb + H;
    if (GG && have_prev) {
      st_off(ggn, so_prev, gg_prev[0]);
      st_off(ggn + V, so_prev, gg_prev[1]);
      st_off(ggn + (int64_t)2 * V, so_prev, gg_prev[2]);
    }
    float gg_new[3] = {0.f, 0.f, 0.f};
    unsigned so_new = 0;
    bool have_new = false;
    fetch_rows(zp + 2, rl);                                       // unconditional: clamped addresses
    fetch_in(zp + 2 + H, il);
    // ---- deposits of sample plane zp; coordinate path of the own rows among them
    if (zp >= 0 && zp < d.s0) {
      const int sz = ((zp % NS) + NS) % NS;
      const bool own_plane = GG && zp >= za && zp < zb;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        // every wave visits one own row; the 2H halo rows go to waves 8 .. 8 + 2H - 1 (waves 0 .. 3 empty the output
        // plane, waves 0 .. 7 carry the second fetch item: with rows dealt out in order they also had two visits each and
        // the other half of the workgroup waited at the barrier)
        const int hrow = wave - 8;
        const int r = k == 0 ? H + wave : (hrow >= 0 && hrow < 2 * H ? (hrow < H ? hrow : TY + hrow) : R);
        const int ys = y0 - H + r;
        if (r >= R || ys < 0 || ys >= d.s1) continue;              // wave-uniform
        const bool own = own_plane && k == 0 && ys < yend;
        float g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = st[(a * R + r) * 64 + lane];
        const float go = st[(3 * R + r) * 64 + lane];
        bool pass[3] = {true, true, true};
        if (clamp_grid) {
#pragma unroll
          for (int a = 0; a < 3; ++a) { pass[a] = g[a] >= -1.f && g[a] <= 1.f; g[a] = clamp_unit(g[a]); }
        }
        Taps<3, PAD> t;
        t.y = make_tap<PAD>(g[1], d.s1);
        t.z = make_tap<PAD>(g[2], d.s0);
        const int zlo = max(za, zp - H), zhi = min(zb, zp + H + 2);
        const bool reach = xin && t.y.i0 + 1 >= y0 && t.y.i0 < yend && t.z.i0 + 1 >= zlo && t.z.i0 < zhi;
        const bool any = __ballot(reach) != 0;
        if (!any && !own) continue;
        t.x = make_tap<PAD>(g[0], d.s2);
        if (any) {
          float wxm[2], wym[2], wzm[2];
          int colx[2], rowy[2], slotz[2];
          bool anyy[2], anyz[2];
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            const int pxx = t.x.i0 + cx;
            const bool okx = xin && pxx >= 0 && pxx < d.s2;          // (masked lanes: own column, see k_scatter_march3d)
            wxm[cx] = okx ? t.wx(cx) : 0.f;
            colx[cx] = okx ? pxx : lane;
          }
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const int py = t.y.i0 + cy;
            anyy[cy] = py >= y0 && py < yend;
            wym[cy] = anyy[cy] ? t.wy(cy) : 0.f;
            rowy[cy] = min(max(py - y0, 0), TY - 1) * 64;
          }
#pragma unroll
          for (int cz = 0; cz < 2; ++cz) {
            const int pz = t.z.i0 + cz;
            anyz[cz] = pz >= zlo && pz < zhi;
            wzm[cz] = anyz[cz] ? t.wz(cz) : 0.f;
          }
#if ADVCHAIN_WIDE_TRIM
          {
            // the upper corner's slot is the lower one's successor in the ring (where the clamp engages the corner carries
            // weight 0 and any slot inside the ring will do): one wrap each instead of two clamps and four selects
            int slot = sz + min(max(t.z.i0 - zp, -H), H + 1);
            slot += slot < 0 ? NS : 0;
            slot -= slot >= NS ? NS : 0;
            slotz[0] = slot * (TY * 64);
            slot += 1;
            slot -= slot >= NS ? NS : 0;
            slotz[1] = slot * (TY * 64);
          }
#else
#pragma unroll
          for (int cz = 0; cz < 2; ++cz) {
            int slot = sz + min(max(t.z.i0 + cz - zp, -H), H + 1);
            slot += slot < 0 ? NS : 0;
            slot -= slot >= NS ? NS : 0;
            slotz[cz] = slot * (TY * 64);
          }
#endif
          const float gsc = go * scale;
          const float a0 = wxm[0] * gsc, a1 = wxm[1] * gsc;
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              const float wzy = wzm[cz] * wym[cy];
              int* cell = acc + slotz[cz] + rowy[cy];
              atomicAdd(cell + colx[0], fix_round(wzy * a0));
              atomicAdd(cell + colx[1], fix_round(wzy * a1));
            }
        }
        if (own) {
          // corners from the ring.  The exact bound puts them inside it; a sample that breaks the contract (NaN: make_tap
          // sends it to -16) reads clamped cells and contributes nothing.
          const int rx = t.x.i0 + 4, ry = t.y.i0 - (y0 - H), rz = t.z.i0 - (zp - H);
          const bool inside = rx >= 3 && rx <= 67 && ry >= 0 && ry < R - 1 && rz >= 0 && rz < 2 * H;
          const int cxr = min(max(rx, 3), 67), cyr = min(max(ry, 0), R - 2);
          const int pz0 = (zp - H) + min(max(rz, 0), 2 * H - 1);
          const int s0 = ring_slot(pz0), s1 = ring_slot(pz0 + 1);
          float v[2][2][2];
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const float* p0 = ring + (s0 * R + cyr + cy) * XS + cxr;
            const float* p1 = ring + (s1 * R + cyr + cy) * XS + cxr;
            v[0][cy][0] = p0[0]; v[0][cy][1] = p0[1];
            v[1][cy][0] = p1[0]; v[1][cy][1] = p1[1];
          }
          const float gq = inside ? go : 0.f;
          float sx = 0.f, sy = 0.f, sw = 0.f;
#pragma unroll
          for (int cz = 0; cz < 2; ++cz) {
            sx = fmaf(t.wz(cz), fmaf(t.y.w1, v[cz][1][1] - v[cz][1][0], t.y.w0 * (v[cz][0][1] - v[cz][0][0])), sx);
            sy = fmaf(t.wz(cz), fmaf(t.x.w1, v[cz][1][1] - v[cz][0][1], t.x.w0 * (v[cz][1][0] - v[cz][0][0])), sy);
          }
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
            sw = fmaf(t.wy(cy), fmaf(t.x.w1, v[1][cy][1] - v[0][cy][1], t.x.w0 * (v[1][cy][0] - v[0][cy][0])), sw);
          so_new = (unsigned)((zp * d.s1 + ys) * d.s2 + xl) * 4u;
          gg_new[0] = pass[0] ? t.x.mult * (gq * sx) : 0.f;
          gg_new[1] = pass[1] ? t.y.mult * (gq * sy) : 0.f;
          gg_new[2] = pass[2] ? t.z.mult * (gq * sw) : 0.f;
          have_new = xin;
        }
      }
    }
    // ---- the next step's rows and ring plane go to the other stage / the free ring slot; one barrier; then the output
    // plane zp-H, which has seen every sample that can reach it, leaves as 16-byte stores while the next step begins
    if (more) {
      commit_rows(stage + (cur ^ 1) * G::STAGE, rc);
      if (zp + 1 < zb) commit_in(zp + 1 + H, ic);
    }
    asm volatile("" ::"v"(gg_prev[0]), "v"(gg_prev[1]), "v"(gg_prev[2]), "v"(so_prev), "v"(fprev.x), "v"(fprev.y), "v"(fprev.z),
                 "v"(fprev.w), "v"(fo_prev));
#pragma unroll
    for (int a = 0; a < 3; ++a) gg_prev[a] = gg_new[a];
    so_prev = so_new;
    have_prev = have_new;
    __syncthreads();
    const int zt = zp - H;
    if (tid < TY * 16 && zt >= za && zt < zb) {
      const int row = tid >> 4, q = tid & 15, uy = y0 + row;
      int4* cell = reinterpret_cast<int4*>(acc + (((zt % NS) + NS) % NS) * (TY * 64) + row * 64 + 4 * q);
      const int4 c = *cell;
      *cell = make_int4(0, 0, 0, 0);
      if (uy < d.s1 && 4 * q < d.s2) {
        fprev = make_float4((float)c.x * inv, (float)c.y * inv, (float)c.z * inv, (float)c.w * inv);
        fo_prev = (unsigned)((zt * d.s1 + uy) * d.s2 + 4 * q) * 4u;
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(ginn) + fo_prev) = fprev;
      }
    }
    cur ^= 1;
  };
  for (int zp = za - H; zp < zb + H; zp += 2) {
    step(zp, rowA, inA, rowB, inB);
    if (zp + 1 < zb + H) step(zp + 1, rowB, inB, rowA, inA);
  }
  if (GG && have_prev) {       // the last step's own row
    st_off(ggn, so_prev, gg_prev[0]);
    st_off(ggn + V, so_prev, gg_prev[1]);
    st_off(ggn + (int64_t)2 * V, so_prev, gg_prev[2]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2D: the same owner-computes scatter without a march.  A workgroup owns TY whole rows of one image (an accumulator of
// C x TY x W 32-bit cells in LDS), visits the sample rows y0-H .. y0+TY+H-1 in 64-lane segments and keeps what lands in
// its rows; then every output leaves with a plain store, the coordinate path of its own sample added (self-composition)
// or stored (grad_grid).  Whole rows: the halo work is in y only, (TY+2H)/TY (the 2D tiles of scatter_tiled.hip pay it
// on both axes and need an overflow list; the window scatter pays 1.7 global float atomics per sample and channel).
// ---------------------------------------------------------------------------------------------------------------------
// A/B build switch (tools/ab/build_all_variant.sh -DADVCHAIN_ROWS2D_FLAT=0): the branch-light deposits of the whole-row scatter
#ifndef ADVCHAIN_ROWS2D_FLAT
#define ADVCHAIN_ROWS2D_FLAT 1
#endif
__device__ __forceinline__ float rows2d_fix_scale(int H) {   // (2H+2)^2 deposits of weight <= 1 stay below 2^31
  return H <= 2 ? 33554432.f : (H <= 4 ? 16777216.f : (H <= 8 ? 4194304.f : (H <= 16 ? 1048576.f : 262144.f)));
}

template <int PAD, int C, bool SELF, bool GG>
__global__ void __launch_bounds__(512)
k_scatter_rows2d(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                 float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int TY, int H, int clamp_grid,
                 int32_t* __restrict__ ws, const float* __restrict__ rowmax_in, float* __restrict__ rowmax_next) {
  extern __shared__ int acc[];                   // [C][TY][W]
  __shared__ int rmax[64];                       // max |output| of the owned rows (float bits), for rowmax_next
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
    const float* rowmax = rowmax_in + (int64_t)n * d.s1;
    float m = 0.f;
    for (int y = ya + (int)threadIdx.x; y < yb; y += NWV * 64) m = fmaxf(m, rowmax[y]);
    if (threadIdx.x < 64) rmax[threadIdx.x] = 0;
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

  // ---- deposits: (row, 64-pixel segment) items, two per wave and round with their loads issued together.  The OWN rows
  // first: their coordinate path (which needs nothing from the accumulator) is evaluated where the row is visited for its
  // deposits -- taps built once, grid / grad_out loaded once (round 5: the second visit was a quarter of the vector-memory
  // instructions of a workgroup, and this kernel is bound by their issue, lesson 4) -- then the halo rows above and below.
  // Integer adds commute: the accumulator, and with it every output bit, is what the single loop over all rows left.
  constexpr int U = 2;
  constexpr int MAXO = 8;                         // own items per wave: TY * nseg / 8, capped by the launcher
  const int oitems = (yend - y0) * nseg;
  // item -> (row, segment) through a reciprocal (exact: items < 2^12): an integer division by the run-time nseg is ~20
  // instructions, three times per item
  const float inv_nseg = 1.f / (float)nseg;
  auto item_row = [&](int it) { return (int)(((float)it + 0.5f) * inv_nseg); };
  float ggv[MAXO][2];
  auto visit = [&](int ys, int xs0, bool own, float (&g)[2], const float (&go)[C], float (&gg)[2]) {
    const bool xin = xs0 < W;
    bool pass[2] = {true, true};
    if (clamp_grid) {
#pragma unroll
      for (int a = 0; a < 2; ++a) { pass[a] = g[a] >= -1.f && g[a] <= 1.f; g[a] = clamp_unit(g[a]); }
    }
    Taps<2, PAD> t;
    t.y = make_tap<PAD>(g[1], d.s1);
    const bool any = __ballot(xin && t.y.i0 + 1 >= y0 && t.y.i0 < yend) != 0;
    if (!any && !(own && (SELF || GG))) return;
    t.x = make_tap<PAD>(g[0], d.s2);
    t.z.i0 = 0; t.z.w0 = 1.f; t.z.w1 = 0.f; t.z.mult = 0.f; t.z.v0 = true; t.z.v1 = false;
    float vl[C][8];
    if (own && (SELF || GG)) {                      // corner values requested ahead of the deposits
      const CornerOffsets<2, PAD> o(t, d);
#pragma unroll
      for (int c = 0; c < C; ++c) o.load(inn + (int64_t)c * V, vl[c]);
    }
    if (any) {
      if constexpr (ADVCHAIN_ROWS2D_FLAT && SELF) {
      // branch-light deposits for the squarings (round 6; the 3D march scatters do the same since lesson 37 b): per-axis masked weights, and
      // a masked corner adds its zero at the lane's OWN column of an owned row -- never two lanes on one cell -- instead of
      // four exec-mask regions per item; 24-bit multiplies for the cell rows (a 32-bit v_mul_lo costs four VALU slots)
      const int ownr = __mul24(min(max(ys - y0, 0), TY - 1), W), ownc = min(xs0, W - 1);
      int rowc[2], colc[2];
      float wxm[2], wym[2];
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        const bool ok = xin && (cx ? t.x.v1 : t.x.v0);
        wxm[cx] = ok ? (cx ? t.x.w1 : t.x.w0) * scale : 0.f;
        colc[cx] = ok ? t.x.i0 + cx : ownc;
      }
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        const int py = t.y.i0 + cy;
        const bool ok = py >= y0 && py < yend && (cy ? t.y.v1 : t.y.v0);
        wym[cy] = ok ? (cy ? t.y.w1 : t.y.w0) : 0.f;
        rowc[cy] = ok ? __mul24(py - y0, W) : ownr;
      }
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const float wsc = wxm[cx] * wym[cy];
          int* cell = acc + rowc[cy] + colc[cx];
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(cell + c * TY * W, fix_round(wsc * go[c]));
        }
      } else {
      // (the image warps keep the per-corner branches: with four channels the masked form holds 89 VGPRs and measured
      // 82 against 74 us at 12 px)
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
          for (int c = 0; c < C; ++c) atomicAdd(cell + c * TY * W, fix_round(wsc * go[c]));
        }
      }
      }
    }
    if (own && (SELF || GG)) {
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) coord_path_values<2, PAD>(vl[c], go[c], t, ax, ay, az);
      gg[0] = pass[0] ? t.x.mult * ax : 0.f;
      gg[1] = pass[1] ? t.y.mult * ay : 0.f;
    }
    (void)ys;
  };
  // own items it = wave + k * NWV (static register slots for their coordinate path), two per round
#pragma unroll
  for (int k0 = 0; k0 < MAXO; k0 += U) {
    float g[U][2], go[U][C];
    int xs[U], ys[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ggv[k0 + u][0] = ggv[k0 + u][1] = 0.f;
      const int it = min(wave + (k0 + u) * NWV, max(oitems - 1, 0));
      const int r = item_row(it);
      ys[u] = y0 + r;
      xs[u] = (it - r * nseg) * 64 + lane;
      const int s = ys[u] * W + min(xs[u], W - 1);
      g[u][0] = gn[s];
      g[u][1] = gn[V + s];
#pragma unroll
      for (int c = 0; c < C; ++c) go[u][c] = gon[(int64_t)c * V + s];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (wave + (k0 + u) * NWV >= oitems) continue;                  // wave-uniform
      visit(ys[u], xs[u], true, g[u], go[u], ggv[k0 + u]);
    }
  }
  // halo items: rows ya .. y0-1 and yend .. yb-1
  const int hlo = y0 - ya, hitems = (hlo + (yb - yend)) * nseg;
  for (int it0 = wave * U; it0 < hitems; it0 += NWV * U) {
    float g[U][2], go[U][C];
    int xs[U], ys[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(it0 + u, hitems - 1);
      const int r = item_row(it);
      ys[u] = r < hlo ? ya + r : yend + (r - hlo);
      xs[u] = (it - r * nseg) * 64 + lane;
      const int s = ys[u] * W + min(xs[u], W - 1);
      g[u][0] = gn[s];
      g[u][1] = gn[V + s];
#pragma unroll
      for (int c = 0; c < C; ++c) go[u][c] = gon[(int64_t)c * V + s];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (it0 + u >= hitems) continue;                                // wave-uniform
      float dummy[2];
      visit(ys[u], xs[u], false, g[u], go[u], dummy);
    }
  }
  __syncthreads();
  // ---- every owned output: convert, add / store the coordinate path, plain stores.  rowmax_next: the row maxima of what
  // is written here -- the next squaring's backward reads this tensor as ITS grad_out and would otherwise launch
  // k_march_rowmax over it first (the values and the non-finite rule of that kernel)
#pragma unroll
  for (int k = 0; k < MAXO; ++k) {
    const int it = wave + k * NWV;
    if (it >= oitems) continue;                                       // wave-uniform
    const int r = item_row(it);
    const int x = (it - r * nseg) * 64 + lane;
    const bool on = x < W;
    const int s = (y0 + r) * W + min(x, W - 1);
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = (float)acc[(c * TY + r) * W + min(x, W - 1)] * inv;
    if (SELF) {
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] += ggv[k][c < 2 ? c : 0];
    } else if (GG) {
      float* gq = ggrid + (int64_t)n * 2 * V + s;
      if (on) { gq[0] = ggv[k][0]; gq[V] = ggv[k][1]; }
    }
    if (on) {
#pragma unroll
      for (int c = 0; c < C; ++c) ginn[(int64_t)c * V + s] = v[c];
    }
    if (rowmax_next) {                                                // (uniform)
      float m = 0.f;
      bool bad = false;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        m = fmaxf(m, on ? fabsf(v[c]) : 0.f);
        bad = bad || (on && !(fabsf(v[c]) <= 3.0e38f));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (__ballot(bad) != 0) m = __int_as_float(0x7f800000);
      if (lane == 0) atomicMax(&rmax[r], __float_as_int(m));
    }
  }
  if (rowmax_next) {
    __syncthreads();
    if ((int)threadIdx.x < yend - y0) rowmax_next[(int64_t)n * d.s1 + y0 + threadIdx.x] = __int_as_float(rmax[threadIdx.x]);
  }
}

}  // namespace advchain

using namespace advchain;

// The shapes / bounds the whole-row scatter takes; TY = owned rows per workgroup (0: not taken).
static int rows2d_tile(bool self, int64_t C, const Dims& d, int padding, int H, int64_t N) {
  static const bool off = getenv("ADVCHAIN_NO_SCATTER_ROWS2D") != nullptr;   // A/B knob
  // from which bound on: 3 pixels (below, the gather form of adjoint_gather.hip is faster) -- for a squaring whose launch fills
  // the machine, 2 (round 5, once the own rows' coordinate path moved into their deposit visit: 35 against 46 us in a chain
  // at 64 x 2 x 256 x 256; with 96 workgroups, 8 x 2 x 192 x 192, the gather form wins, 8.6 against 14.3 us)
  const int hmin = (self && N * ((d.s1 + 15) / 16) >= 512) ? 2 : 3;
  if (off || padding == PAD_REFLECTION || H < hmin || H > (self ? 32 : 16) || d.s0 != 1) return 0;
  if (d.s2 < 16 || d.s2 > 512 || d.voxels() * 4 >= (1ll << 31)) return 0;
  if (self ? C != 2 : (C != 1 && C != 4)) return 0;
  const int nseg = (d.s2 + 63) / 64;
  int TY = H >= 8 ? 32 : 16;
  static const size_t lds_cap = 49152;   // measured optimum (was a tuning knob until round 4)
  while (TY > 4 && ((size_t)C * TY * d.s2 * 4 > lds_cap || TY * nseg > 64)) TY >>= 1;   // 48 KiB of cells, 8 own items a wave
  if ((size_t)C * TY * d.s2 * sizeof(int) > 65536 - 64 || TY * nseg > 64) return 0;
  return TY;
}
bool advchain_scatter_rows2d_takes(bool self, int64_t C, Dims d, int padding, int H, int64_t N) { return rows2d_tile(self, C, d, padding, H, N) > 0; }

// 2D, exact bound of H = 3..16 pixels (squarings: ..32).  ADVCHAIN_ERR_UNSUPPORTED: use the gather form / the window scatter.
// rm_flags (the chain's consecutive launches; 0 = a launch on its own): bit 0 = the row maxima of `gout` are already in
// this launch's buffer (left by the launch that wrote gout) -- no k_march_rowmax pre-pass; bit 1 = leave the row maxima
// of `gin` for the next launch; bit 2 = which of the two buffers in `workspace` this launch reads (it writes the other).
int advchain_scatter_rows2d_launch(bool self, const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                   int64_t N, int64_t C, Dims d, int padding, int clamp_grid, int H, int32_t* workspace,
                                   hipStream_t st, int rm_flags) {
  const int TY = (workspace && gin) ? rows2d_tile(self, C, d, padding, H, N) : 0;
  if (TY == 0) return ADVCHAIN_ERR_UNSUPPORTED;
  const size_t lds = (size_t)C * TY * d.s2 * sizeof(int);
  float* rm_a = reinterpret_cast<float*>(workspace + 4);       // (the overflow list of the tiled kernels: unused here)
  float* rm_b = rm_a + N * d.s1;
  float* rm_in = (rm_flags & 4) ? rm_b : rm_a;
  float* rm_next = (rm_flags & 2) ? ((rm_flags & 4) ? rm_a : rm_b) : nullptr;
  if (!(rm_flags & 1)) {
    dim3 rg((unsigned)((d.s1 + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    if (C == 1) hipLaunchKernelGGL(k_march_rowmax<1>, rg, dim3(kBlock), 0, st, gout, rm_in, d, d.s1);
    else if (C == 2) hipLaunchKernelGGL(k_march_rowmax<2>, rg, dim3(kBlock), 0, st, gout, rm_in, d, d.s1);
    else hipLaunchKernelGGL(k_march_rowmax<4>, rg, dim3(kBlock), 0, st, gout, rm_in, d, d.s1);
  }
  dim3 g((unsigned)((d.s1 + TY - 1) / TY), (unsigned)N), b(512);
  const bool gg = ggrid != nullptr;
#define GO(PAD_, C_, SELF_, GG_) \
  hipLaunchKernelGGL((k_scatter_rows2d<PAD_, C_, SELF_, GG_>), g, b, lds, st, gout, in, grid, gin, ggrid, d, TY, H, clamp_grid, workspace, rm_in, rm_next)
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
  static const int hmax = 8;   // measured optimum (was a tuning knob until round 4)
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
      launch_rowmax<1>(gout + (int64_t)c0 * d.voxels(), rowmax, d, rows, N, (int)C, st);
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
  // self-composition on rows of 68 .. 80 voxels: lane <-> flat sample (k_scatter_march3d_flat)
  static const bool no_flat = getenv("ADVCHAIN_NO_SCATTER_MARCH_FLAT") != nullptr;   // A/B knob
  if (!no_flat && self && H <= 4 && d.s2 > 64 && d.s2 <= 80) {
    const int TYf = 8, W = (int)d.s2;
    int NSf = 2 * H + 3;
    if ((size_t)NSf * 3 * TYf * W * 4 > 80 * 1024) NSf = 2 * H + 2;       // two workgroups a CU
    const size_t ldsf = (size_t)NSf * 3 * TYf * W * sizeof(int);
    const int n1f = (int)((d.s1 + TYf - 1) / TYf);
    int zcf = (int)d.s0;
    while (zcf > 8 && N * n1f * ((d.s0 + zcf - 1) / zcf) < 512) zcf = (zcf + 1) / 2;
    static const int zcf_forced = 0;   // (0: the rule below; was a tuning knob until round 4)
    if (zcf_forced > 0) zcf = zcf_forced;
    const int n0f = (int)((d.s0 + zcf - 1) / zcf);
    const int rows = (int)(d.s0 * d.s1);
    launch_rowmax<3>(gout, reinterpret_cast<float*>(workspace + 4), d, rows, N, 3, st);
    const int nitems = ((TYf + 2 * H) * W + 63) / 64;
    dim3 gf((unsigned)(n1f * n0f), (unsigned)N);
#define GOF(MAXR_) do { \
      auto kern = k_scatter_march3d_flat<8, MAXR_>; \
      static bool attr_set = false; \
      if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set = true; } \
      hipLaunchKernelGGL(kern, gf, dim3(512), ldsf, st, gout, grid, gin, d, n1f, zcf, TYf, H, NSf, workspace); } while (0)
    if (nitems <= 16) GOF(2); else GOF(3);
#undef GOF
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  // image warps with rows of at most 64 voxels: the 16-byte form
  static const bool no_wide = getenv("ADVCHAIN_NO_SCATTER_MARCH_WIDE") != nullptr;   // A/B knob
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!no_wide && !self && C == 1 && H <= 4 && d.s2 <= 64 && d.s2 % 4 == 0 && aligned16(gout) && aligned16(in) && aligned16(grid) &&
      aligned16(gin) && aligned16(ggrid)) {
    const bool ggw = ggrid != nullptr;
    const int n1w = (int)((d.s1 + 15) / 16);
    int zcw = (int)d.s0;
    while (zcw > 16 && N * n1w * ((d.s0 + zcw - 1) / zcw) < 256) zcw = (zcw + 1) / 2;
    static const int zcw_forced = 0;   // (0: the rule below; was a tuning knob until round 4)
    if (zcw_forced > 0) zcw = zcw_forced;
    const int n0w = (int)((d.s0 + zcw - 1) / zcw);
    const int rows = (int)(d.s0 * d.s1);
    dim3 rg((unsigned)((rows + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    // round 6: the workgroups take the maximum over their own rows themselves (ADVCHAIN_WIDE_ROWMAX_PASS: the pre-pass, A/B)
    static const bool rowmax_pass = getenv("ADVCHAIN_WIDE_ROWMAX_PASS") != nullptr;
    if (rowmax_pass) launch_rowmax<1>(gout, reinterpret_cast<float*>(workspace + 4), d, rows, N, 1, st);
    int32_t* const ws_arg = rowmax_pass ? workspace : nullptr;
    dim3 gw((unsigned)(n1w * n0w * N));
#define GOW(PAD_, GG_, H_) do { \
      auto kern = k_scatter_march3d_wide<PAD_, GG_, H_>; \
      static bool attr_set = false; \
      if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideCfg<H_>::lds(GG_)); attr_set = true; } \
      hipLaunchKernelGGL(kern, gw, dim3(WideCfg<H_>::NT), WideCfg<H_>::lds(GG_), st, gout, in, grid, gin, ggrid, d, n1w, zcw, clamp_grid, ws_arg); } while (0)
#define GOW_H(PAD_, GG_) do { if (H == 2) GOW(PAD_, GG_, 2); else if (H == 3) GOW(PAD_, GG_, 3); else GOW(PAD_, GG_, 4); } while (0)
    if (padding == PAD_BORDER) { if (ggw) GOW_H(PAD_BORDER, true); else GOW_H(PAD_BORDER, false); }
    else { if (ggw) GOW_H(PAD_ZEROS, true); else GOW_H(PAD_ZEROS, false); }
#undef GOW_H
#undef GOW
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  // rows per workgroup: as many as 60 KiB of accumulator planes allow, at most 8
  static const int ty_forced = 0;
  int NS = 2 * H + 3;
  int TY = 8;    // (16 halves the y halo work but leaves 512 workgroups at 4 x 128 x 128 x 64)
  if ((size_t)NS * C * TY * 64 * 4 > 65536) NS = 2 * H + 2;
  while (TY > 4 && (size_t)NS * C * TY * 64 * 4 > 65536) TY >>= 1;
  if (ty_forced == 4 || ty_forced == 8) TY = ty_forced;
  const size_t lds = (size_t)NS * C * TY * 64 * sizeof(int);
  if (lds > 65536) return ADVCHAIN_ERR_UNSUPPORTED;
  const int n1 = (d.s1 + TY - 1) / TY;
  static const int zc_forced = 0;   // (0: the rule below; was a tuning knob until round 4)
  int zc = d.s0;
  while (zc > 8 && N * n1 * nseg * ((d.s0 + zc - 1) / zc) < 512) zc = (zc + 1) / 2;   // (16 planes: 1003 GB/s, 8: 942, 32: 834)
  if (zc_forced > 0) zc = zc_forced;
  const int n0 = (d.s0 + zc - 1) / zc;
  {
    const int rows = (int)(d.s0 * d.s1);
    dim3 rg((unsigned)((rows + kBlock / 16 - 1) / (kBlock / 16)), (unsigned)N);
    float* rowmax = reinterpret_cast<float*>(workspace + 4);       // the overflow list of the tiled kernels: unused here
    if (C == 1) launch_rowmax<1>(gout, rowmax, d, rows, N, 1, st);
    else if (C == 3) launch_rowmax<3>(gout, rowmax, d, rows, N, 3, st);
    else launch_rowmax<4>(gout, rowmax, d, rows, N, 4, st);
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
