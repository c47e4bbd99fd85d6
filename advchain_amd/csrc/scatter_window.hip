// Source-tiled window scatter for the sampler backward above the gather form's displacement bound (gfx950): 2D and 3D.
//
// The owner-computes tiles of scatter_tiled.hip walk a halo as wide as the largest displacement: at cfg-2 the fields
// reach 70 px, the halo is 16 px, every sample is processed by 4 owners and what lands beyond the halo goes through
// an overflow list (image-warp backward: 195 MB of traffic for 96 MB of data, 0.37 TB/s).  But the fields are smooth:
// the targets of a 32 x 32 tile of SAMPLES form a compact patch -- the tile shifted by its mean displacement and
// stretched by the local gradient -- however far it moved.  So here a workgroup owns a tile of samples:
//   1. builds the taps of its 1024 samples once (4 per thread) and reduces the bounding box of their corners and the
//      tile's max |grad_out|;
//   2. accumulates the deposits in an LDS window over that box as 32-bit fixed point (value * 2^20 / max|grad_out|:
//      at most 1024 deposits of weight <= 1 reach a cell, so the sum stays below 2^30; LDS integer atomics run at LDS
//      rate, LDS float atomics do not); corners outside a window that had to be capped go straight to global atomics;
//   3. flushes the non-zero cells with global float atomics -- rows of consecutive addresses, ~1.7 per sample instead
//      of 4 scattered ones.
// Every sample is read and processed exactly once, there is no halo, no overflow list and no max|grad_out| pre-pass;
// the price is a zero-filled destination and float-atomic (order-dependent, ~1e-7 relative) accumulation across tiles.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

// Deterministic mode (advchain_set_deterministic, round 6): the three places where tiles meet in global memory -- the flush
// of a window, a deposit outside a capped window, the coordinate path of a self-composition -- add 64-bit FIXED POINT
// (value * 2^40 / max|grad_out| of the batch entry, integer atomics: the order of arrival no longer matters) into an int64
// image of grad_in that lives in the caller's workspace, and a last pass converts it (k_det_convert): bit-reproducible run
// to run.  The scale comes from a max |grad_out| per batch entry (k_det_absmax; a NaN / inf there turns the entry's
// outputs into NaN) -- per entry, so that a sample's result does not depend on what else is in the batch.  Resolution
// 2^-40 of that maximum per addition; 2^22 additions of the maximum itself fit below 2^63.
constexpr float kDetFix = 1099511627776.f;   // 2^40
template <bool DET>
__device__ __forceinline__ void win_global_add(float* __restrict__ gin, unsigned long long* __restrict__ acc, int64_t idx,
                                               float v, float sdet) {
  if (DET) atomicAdd(acc + idx, (unsigned long long)__float2ll_rn(v * sdet));
  else atomic_add_f32(gin + idx, v);
}
__device__ __forceinline__ float det_scale(const float* __restrict__ maxn, int n) {
  const float m = maxn[n];
  return (m > 0.f && m <= 3.0e38f) ? kDetFix / m : 0.f;
}

// max |x| per batch entry (over `per_n` floats) -> maxn[n] (zeroed by the caller); non-finite -> +inf
__global__ void __launch_bounds__(kBlock) k_det_absmax(const float* __restrict__ x, float* __restrict__ maxn, int64_t per_n) {
  const int n = blockIdx.y;
  const float* p = x + (int64_t)n * per_n;
  float m = 0.f;
  bool bad = false;
  const int64_t stride = (int64_t)gridDim.x * kBlock * 4;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; i < per_n; i += stride) {
    float v[4];
    if (i + 4 <= per_n && ((uintptr_t)(p + i) & 15) == 0) {
      const float4 q = *reinterpret_cast<const float4*>(p + i);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i + j < per_n ? p[i + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { m = fmaxf(m, fabsf(v[j])); bad = bad || !(fabsf(v[j]) <= 3.0e38f); }
  }
  if (bad) m = __int_as_float(0x7f800000);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, wm[w]);
    atomicMax(reinterpret_cast<int*>(maxn) + n, __float_as_int(m));      // (non-negative floats order as their bit patterns)
  }
}

// grad_in = int64 image * max / 2^40 (a non-finite maximum: 0 * inf = NaN, as the owner-computes scatters do)
__global__ void __launch_bounds__(kBlock) k_det_convert(const long long* __restrict__ acc, const float* __restrict__ maxn,
                                                        float* __restrict__ gin, int64_t per_n) {
  const int n = blockIdx.y;
  const float inv = maxn[n] * (1.f / kDetFix);
  const long long* a = acc + (int64_t)n * per_n;
  float* g = gin + (int64_t)n * per_n;
  const int64_t stride = (int64_t)gridDim.x * kBlock * 2;
  for (int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2; i < per_n; i += stride) {
    if (i + 2 <= per_n) {
      const longlong2 q = *reinterpret_cast<const longlong2*>(a + i);
      g[i] = (float)q.x * inv;
      g[i + 1] = (float)q.y * inv;
    } else {
      g[i] = (float)a[i] * inv;
    }
  }
}

// A/B build switch (tools/ab/build_all_variant.sh -DADVCHAIN_WINDOW_FLAT=0): the branch-light deposits of the 3D window scatter
#ifndef ADVCHAIN_WINDOW_FLAT
#define ADVCHAIN_WINDOW_FLAT 1
#endif
constexpr int kWinT = 32;               // sample tile edge
constexpr int kWinCells = 8192;         // LDS window budget in cells (all channels together): 32 KiB
constexpr int kWinCellsC4 = 12288;      // four channels: 48 KiB (2048 cells per channel = 45 x 45 capped stretched 32 x 32 tiles)

// SELF : in == grid == phi (C == 2); the coordinate-path gradient is added to the same tensor (atomics: other tiles
//        deposit there too).  Otherwise GG: grad_grid is written with plain stores.
template <int PAD, int C, bool SELF, bool GG, bool DET = false>
__global__ void __launch_bounds__(kBlock)
k_scatter_window2d(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                   float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n2, int clamp_grid,
                   int32_t* __restrict__ ws, unsigned long long* __restrict__ acc64 = nullptr,
                   const float* __restrict__ maxn = nullptr) {
  constexpr int kCells2 = C == 4 ? kWinCellsC4 : kWinCells;
  __shared__ int win[kCells2];
  if (ws && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[3] = -1;   // no max|result| from this launch
  __shared__ int red[8][kBlock / 64];
  constexpr int DIM = 2;
  constexpr int SPT = kWinT * kWinT / kBlock;   // samples per thread (4)
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int tx = blockIdx.x % n2, ty = blockIdx.x / n2;
  const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;      // x in tile, first row; rows ly0 + 8 j
  const int sx = tx * kWinT + lx;
  const float* gn = grid + (int64_t)n * DIM * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
  unsigned long long* accn = DET ? acc64 + (int64_t)n * C * V : nullptr;
  const float sdet = DET ? det_scale(maxn, n) : 0.f;

  // ---- 1. taps of this thread's samples, bounding box of the valid corners, max |grad_out|
  Taps<DIM, PAD> t[SPT];
  float go[SPT][C];
  bool live[SPT], px[SPT], py[SPT];
  int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30);
  float gmax = 0.f;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int sy = ty * kWinT + ly0 + 8 * j;
    live[j] = sx < d.s2 && sy < d.s1;
    const int s = live[j] ? sy * d.s2 + sx : 0;
    float gx = gn[s], gy = gn[V + s];
    px[j] = py[j] = true;
    if (clamp_grid) {
      px[j] = gx >= -1.f && gx <= 1.f; py[j] = gy >= -1.f && gy <= 1.f;
      gx = clamp_unit(gx); gy = clamp_unit(gy);
    }
    t[j].build(gx, gy, 0.f, d);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float g = gon[(int64_t)c * V + s];     // unconditional (s = 0 for a dead sample): no branch around the load
      go[j][c] = live[j] ? g : 0.f;
      gmax = fmaxf(gmax, fabsf(go[j][c]));
    }
    if (live[j]) {
      if (t[j].x.v0 || t[j].x.v1) { bx0 = min(bx0, t[j].x.i0 + (t[j].x.v0 ? 0 : 1)); bx1 = max(bx1, t[j].x.i0 + (t[j].x.v1 ? 1 : 0)); }
      if (t[j].y.v0 || t[j].y.v1) { by0 = min(by0, t[j].y.i0 + (t[j].y.v0 ? 0 : 1)); by1 = max(by1, t[j].y.i0 + (t[j].y.v1 ? 1 : 0)); }
    }
  }
  // block reduction: 4 ints + 1 float through the wave, then LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, o, 64)); bx1 = max(bx1, __shfl_xor(bx1, o, 64));
    by0 = min(by0, __shfl_xor(by0, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64));
    gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = bx0; red[1][wave] = bx1; red[2][wave] = by0; red[3][wave] = by1; red[4][wave] = __float_as_int(gmax);
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    bx0 = min(bx0, red[0][w]); bx1 = max(bx1, red[1][w]); by0 = min(by0, red[2][w]); by1 = max(by1, red[3][w]);
    gmax = fmaxf(gmax, __int_as_float(red[4][w]));
  }
  // window = box, capped to the LDS budget (keeps the low corner; what falls outside uses global atomics)
  int ww = max(bx1 - bx0 + 1, 0), wh = max(by1 - by0 + 1, 0);
  constexpr int cells_per_ch = kCells2 / C;
  if (ww > 128) ww = 128;
  if (ww > 0 && wh > cells_per_ch / ww) wh = cells_per_ch / ww;
  const int cells = ww * wh;
  for (int i = threadIdx.x; i < C * cells; i += kBlock) win[i] = 0;
  const float scale = gmax > 0.f ? 1048576.f / gmax : 0.f;   // 2^20
  __syncthreads();

  // ---- coordinate path of all four samples first: 4 x 4 x C corner loads in flight together, no control flow between
  // them (a dead sample has grad_out 0 and valid addresses)
  float cgx[SPT], cgy[SPT];
  if (SELF || GG) {
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      float ax = 0.f, ay = 0.f, dummy = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c)
        sample_linear_bwd<DIM, PAD, false, true>(inn + (int64_t)c * V, nullptr, go[j][c], t[j], d, ax, ay, dummy);
      cgx[j] = px[j] ? t[j].x.mult * ax : 0.f;
      cgy[j] = py[j] ? t[j].y.mult * ay : 0.f;
    }
  }

  // ---- 2. deposits
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    if (!live[j]) continue;
    const int sy = ty * kWinT + ly0 + 8 * j;
    const int s = sy * d.s2 + sx;
    const int wx0 = t[j].x.i0 - bx0, wy0 = t[j].y.i0 - by0;
    const int cell0 = __mul24(wy0, ww) + wx0;                          // corner (0,0); the others are +1, +ww away
    const int vox0 = __mul24(t[j].y.i0, d.s2) + t[j].x.i0;
    const bool inx[2] = {wx0 >= 0 && wx0 < ww, wx0 + 1 >= 0 && wx0 + 1 < ww};
    const bool iny[2] = {wy0 >= 0 && wy0 < wh, wy0 + 1 >= 0 && wy0 + 1 < wh};
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        if (!t[j].ok(0, cy, cx)) continue;
        const float w = t[j].w(0, cy, cx);
        if (inx[cx] && iny[cy]) {
          const float ws = w * scale;
          int* cell = win + cell0 + (cy ? ww : 0) + cx;
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(cell + c * cells, fix_round(ws * go[j][c]));
        } else {
          const int64_t dst = vox0 + (cy ? d.s2 : 0) + cx;
#pragma unroll
          for (int c = 0; c < C; ++c) win_global_add<DET>(ginn, accn, dst + (int64_t)c * V, w * go[j][c], sdet);
        }
      }
    if (SELF || GG) {
      const float ggx = cgx[j], ggy = cgy[j];
      if (SELF) {
        if (ggx != 0.f) win_global_add<DET>(ginn, accn, s, ggx, sdet);
        if (ggy != 0.f) win_global_add<DET>(ginn, accn, (int64_t)V + s, ggy, sdet);
      } else {
        float* gg = ggrid + (int64_t)n * DIM * V + s;
        gg[0] = ggx;
        gg[V] = ggy;
      }
    }
  }
  __syncthreads();

  // ---- 3. flush the non-zero cells (rows of the window are runs of consecutive addresses)
  const float inv = gmax * (1.f / 1048576.f);
  for (int i = threadIdx.x; i < C * cells; i += kBlock) {
    const int a = win[i];
    if (a == 0) continue;
    const int c = i / cells, r = i - c * cells;
    const int wy = r / ww, wx = r - wy * ww;
    win_global_add<DET>(ginn, accn, (int64_t)c * V + (by0 + wy) * d.s2 + (bx0 + wx), (float)a * inv, sdet);
  }
}

// ---------------------------------------------------------------------------------------------
// 3D form.  At first it only paid above the 4-voxel halo of the owner-computes tiles (whose overflow list then explodes:
// at cfg-5 an image-warp backward spent 1.1 ms draining it on top of a 1.3 ms kernel); with unconditional corner loads
// it wins from one voxel up (4x128x128x64 self-composition: 207-216 us against 226-300 us for halo 2-4).
// Tile of 4 x 8 x 32 samples (4 per thread along z); the bounding box pass keeps no taps (they are rebuilt from the
// L1/L2-resident grid in the deposit pass: 4 x 3 axis taps per thread would not fit the register budget).
// ---------------------------------------------------------------------------------------------
// (tile shape: template parameters TX x TY (= 256 threads) x TZ samples per thread; see the launcher)
#ifndef ADVCHAIN_WIN3_CP4
#define ADVCHAIN_WIN3_CP4 2
#endif
#ifndef ADVCHAIN_WIN3_CPS
#define ADVCHAIN_WIN3_CPS 3
#endif
constexpr int kWin3CP4 = ADVCHAIN_WIN3_CP4, kWin3CPS = ADVCHAIN_WIN3_CPS;
constexpr int kWin3Cells = 12288;        // 48 KiB of LDS: 3 workgroups per CU
// C = 4: the window holds TWO channels at a time (6144 cells each) and the deposit + flush passes run twice -- 3072 cells
// per channel capped the windows of the 5-8 voxel fields, and what falls outside a window goes to global atomics
// (8x4x128x128x64: 1449 us; with a 63-KiB window for all four channels 1057 us, but two workgroups a CU).

template <int PAD, int C, bool SELF, bool GG, int TX, int TZ, bool DET = false>
__global__ void __launch_bounds__(kBlock)
k_scatter_window3d(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                   float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int n2, int clamp_grid,
                   int32_t* __restrict__ ws, unsigned long long* __restrict__ acc64 = nullptr,
                   const float* __restrict__ maxn = nullptr) {
  __shared__ int win[kWin3Cells];
  constexpr int CP = C == 4 ? kWin3CP4 : (SELF ? kWin3CPS : C);       // channels per deposit / flush pass
  if (ws && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[3] = -1;   // no max|result| from this launch
  __shared__ int red[7][kBlock / 64];
  constexpr int DIM = 3;
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  int b = blockIdx.x;
  const int tx = b % n2; b /= n2;
  const int ty = b % n1, tz = b / n1;
  constexpr int kWin3X = TX, kWin3Y = kBlock / TX, kWin3Z = TZ;
  const int sx = tx * kWin3X + (threadIdx.x % TX), sy = ty * kWin3Y + (threadIdx.x / TX);
  const bool col_live = sx < d.s2 && sy < d.s1;
  const float* gn = grid + (int64_t)n * DIM * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
  unsigned long long* accn = DET ? acc64 + (int64_t)n * C * V : nullptr;
  const float sdet = DET ? det_scale(maxn, n) : 0.f;

  // ---- 1. bounding box of the valid corners, max |grad_out|
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
  float gmax = 0.f;
#pragma unroll
  for (int j = 0; j < kWin3Z; ++j) {
    const int sz = tz * kWin3Z + j;
    const bool live = col_live && sz < d.s0;
    const int s = live ? (sz * d.s1 + sy) * d.s2 + sx : 0;
    float g[3] = {gn[s], gn[V + s], gn[2 * V + s]};
    if (clamp_grid) { g[0] = clamp_unit(g[0]); g[1] = clamp_unit(g[1]); g[2] = clamp_unit(g[2]); }
    Taps<DIM, PAD> t;
    t.build(g[0], g[1], g[2], d);
#pragma unroll
    for (int c = 0; c < C; ++c) gmax = fmaxf(gmax, live ? fabsf(gon[(int64_t)c * V + s]) : 0.f);
    if (live) {
      const AxisTap* ax[3] = {&t.x, &t.y, &t.z};
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (ax[a]->v0 || ax[a]->v1) {
          lo[a] = min(lo[a], ax[a]->i0 + (ax[a]->v0 ? 0 : 1));
          hi[a] = max(hi[a], ax[a]->i0 + (ax[a]->v1 ? 1 : 0));
        }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
    }
    gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    red[6][wave] = __float_as_int(gmax);
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], red[a][w]); hi[a] = max(hi[a], red[3 + a][w]); }
    gmax = fmaxf(gmax, __int_as_float(red[6][w]));
  }
  // window = box capped to the LDS budget (keeps the low corner; what falls outside uses global atomics)
  constexpr int cells_per_ch = kWin3Cells / CP;
  int ww = min(max(hi[0] - lo[0] + 1, 0), 64), wh = max(hi[1] - lo[1] + 1, 0), wd = max(hi[2] - lo[2] + 1, 0);
  if (ww > 0 && wh > cells_per_ch / ww) wh = cells_per_ch / ww;
  if (ww * wh > 0 && wd > cells_per_ch / (ww * wh)) wd = cells_per_ch / (ww * wh);
  const int plane = ww * wh, cells = plane * wd;
  const float scale = gmax > 0.f ? 1048576.f / gmax : 0.f;   // 2^20

  // ---- 2. deposits.  Grid and grad_out of the four samples are requested up front (unconditionally: a dead sample
  // reads voxel 0), so the loop below starts with its operands in flight instead of one memory round trip per sample.
  float gpre[kWin3Z][3], gopre[kWin3Z][C];
#pragma unroll
  for (int j = 0; j < kWin3Z; ++j) {
    const int sz = tz * kWin3Z + j;
    const int s = (col_live && sz < d.s0) ? (sz * d.s1 + sy) * d.s2 + sx : 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) gpre[j][a] = gn[(int64_t)a * V + s];
#pragma unroll
    for (int c = 0; c < C; ++c) gopre[j][c] = gon[(int64_t)c * V + s];
  }
#pragma unroll
  for (int c0 = 0; c0 < C; c0 += CP) {
  for (int i = threadIdx.x; i < CP * cells; i += kBlock) win[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kWin3Z; ++j) {
    const int sz = tz * kWin3Z + j;
    if (!(col_live && sz < d.s0)) continue;
    const int s = (sz * d.s1 + sy) * d.s2 + sx;
    float g[3] = {gpre[j][0], gpre[j][1], gpre[j][2]};
    bool pass[3] = {true, true, true};
    if (clamp_grid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { pass[a] = g[a] >= -1.f && g[a] <= 1.f; g[a] = clamp_unit(g[a]); }
    }
    Taps<DIM, PAD> t;
    t.build(g[0], g[1], g[2], d);
    float go[C];
#pragma unroll
    for (int c = 0; c < C; ++c) go[c] = gopre[j][c];
    // the corner values of the coordinate path (every channel) are requested here and used after the deposits
    constexpr bool CP_PATH = SELF || GG;
    float cv[CP_PATH ? C : 1][8];
    if (CP_PATH && c0 == 0) {
      const CornerOffsets<DIM, PAD> co(t, d);
#pragma unroll
      for (int c = 0; c < C; ++c) co.load(inn + (int64_t)c * V, cv[c]);
    }
    // window cell / voxel of corner (0,0,0) once (24-bit multiplies: a 32-bit v_mul_lo costs four VALU slots); the
    // other corners are +1, +ww, +plane away
    const int wx0 = t.x.i0 - lo[0], wy0 = t.y.i0 - lo[1], wz0 = t.z.i0 - lo[2];
    const int cell0 = __mul24(wz0, plane) + __mul24(wy0, ww) + wx0;
    const int vox0 = __mul24(__mul24(t.z.i0, d.s1) + t.y.i0, d.s2) + t.x.i0;
    const int rowv = d.s2, planev = __mul24(d.s1, d.s2);
    const bool inx[2] = {wx0 >= 0 && wx0 < ww, wx0 + 1 >= 0 && wx0 + 1 < ww};
    const bool iny[2] = {wy0 >= 0 && wy0 < wh, wy0 + 1 >= 0 && wy0 + 1 < wh};
    const bool inz[2] = {wz0 >= 0 && wz0 < wd, wz0 + 1 >= 0 && wz0 + 1 < wd};
#if ADVCHAIN_WINDOW_FLAT
    {
      // branch-light deposits (round 6, as in the owner-computes scatters): every corner adds into the window with a masked
      // weight -- a corner that is invalid or lies outside a capped window adds zero at a cell of the lane's own -- and only a
      // wave that really has a valid corner outside its window (rare on smooth fields) takes the global-atomic branch.
      // 3D only: -1..-2 % per launch, cfg-5 82.56 -> 82.24 ms; the 2D kernel measured no different and keeps its branches
      const int own_cell = cells > 0 ? (int)threadIdx.x % cells : 0;
      bool outside = false;
#pragma unroll
      for (int cz = 0; cz < 2; ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            const bool okc = t.ok(cz, cy, cx), inw = inx[cx] && iny[cy] && inz[cz] && cells > 0;
            outside = outside || (okc && !inw);
            const float ws = (okc && inw) ? t.w(cz, cy, cx) * scale : 0.f;
            int* cell = win + ((okc && inw) ? cell0 + (cz ? plane : 0) + (cy ? ww : 0) + cx : own_cell);
#pragma unroll
            for (int c = 0; c < CP; ++c) atomicAdd(cell + c * cells, fix_round(ws * go[c0 + c]));
          }
      if (__ballot(outside) != 0) {
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
              if (!t.ok(cz, cy, cx) || (inx[cx] && iny[cy] && inz[cz] && cells > 0)) continue;
              const float w = t.w(cz, cy, cx);
              const int64_t dst = vox0 + (cz ? planev : 0) + (cy ? rowv : 0) + cx + (int64_t)c0 * V;
#pragma unroll
              for (int c = 0; c < CP; ++c) win_global_add<DET>(ginn, accn, dst + (int64_t)c * V, w * go[c0 + c], sdet);
            }
      }
    }
#else
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          if (!t.ok(cz, cy, cx)) continue;
          const float w = t.w(cz, cy, cx);
          if (inx[cx] && iny[cy] && inz[cz]) {
            const float ws = w * scale;
            int* cell = win + cell0 + (cz ? plane : 0) + (cy ? ww : 0) + cx;
#pragma unroll
            for (int c = 0; c < CP; ++c) atomicAdd(cell + c * cells, fix_round(ws * go[c0 + c]));
          } else {
            const int64_t dst = vox0 + (cz ? planev : 0) + (cy ? rowv : 0) + cx + (int64_t)c0 * V;
#pragma unroll
            for (int c = 0; c < CP; ++c) win_global_add<DET>(ginn, accn, dst + (int64_t)c * V, w * go[c0 + c], sdet);
          }
        }
#endif
    if ((SELF || GG) && c0 == 0) {
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) sample_linear_bwd_values<DIM, PAD>(cv[c], go[c], t, ax, ay, az);
      const float ggx = pass[0] ? t.x.mult * ax : 0.f, ggy = pass[1] ? t.y.mult * ay : 0.f,
                  ggz = pass[2] ? t.z.mult * az : 0.f;
      if (SELF) {
        if (ggx != 0.f) win_global_add<DET>(ginn, accn, s, ggx, sdet);
        if (ggy != 0.f) win_global_add<DET>(ginn, accn, (int64_t)V + s, ggy, sdet);
        if (ggz != 0.f) win_global_add<DET>(ginn, accn, 2 * (int64_t)V + s, ggz, sdet);
      } else {
        float* gg = ggrid + (int64_t)n * DIM * V + s;
        gg[0] = ggx;
        gg[V] = ggy;
        gg[2 * (int64_t)V] = ggz;
      }
    }
  }
  __syncthreads();

  // ---- 3. flush: thread <-> cell of the window in memory order (runs of consecutive addresses; a wave per window row left
  // the lanes beyond the row's width idle, 18-40 of 64: 13-21 % of the kernel at 8 x . x 160 x 160 x 80)
  const float inv = gmax * (1.f / 1048576.f);
  {
    const float inv_ww = 1.f / (float)max(ww, 1), inv_wh = 1.f / (float)max(wh, 1), inv_wd = 1.f / (float)max(wd, 1);
    for (int i = threadIdx.x; i < CP * cells; i += kBlock) {
      const int a = win[i];
      if (a == 0) continue;
      const int r = (int)(((float)i + 0.5f) * inv_ww), wx = i - r * ww;          // (exact: i < 2^15)
      const int rz = (int)(((float)r + 0.5f) * inv_wh), wy = r - rz * wh;
      const int c = (int)(((float)rz + 0.5f) * inv_wd), wz = rz - c * wd;
      win_global_add<DET>(ginn, accn, (int64_t)(c0 + c) * V + ((lo[2] + wz) * d.s1 + (lo[1] + wy)) * d.s2 + (lo[0] + wx),
                          (float)a * inv, sdet);
    }
  }
  if (c0 + CP < C) __syncthreads();          // the window is cleared for the next channel pair
  }
}

}  // namespace advchain

using namespace advchain;

// grad_in (and, for SELF, the same tensor) is zero-filled here.  2D: every displacement the gather form does not take.
// 3D: from the displacement hint |halo| >= 2 (everything above the gather form;
// since the corner loads are issued unconditionally it beats the owner-computes tiles from one voxel up).
// A chained workspace is told (by the kernel itself) that this launch left no max|result| behind (header [3] = -1:
// see scatter_tiled.hip).
// det_ws (deterministic mode): the caller's advchain_scatter_workspace buffer; its tail holds the int64 image of grad_in and
// the per-entry maxima -- three more launches (clear, maxima, convert), no float atomic between tiles.
// Returns ADVCHAIN_ERR_UNSUPPORTED for what the kernels do not cover (the caller keeps the owner-computes tiles).
extern "C" int advchain_get_deterministic(void);
int advchain_scatter_window_launch(bool self, const float* gout, const float* in, const float* grid, float* gin,
                                   float* ggrid, int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                   int halo, int32_t* workspace, hipStream_t st, int32_t* det_ws) {
  static const bool off = getenv("ADVCHAIN_NO_WINDOW_SCATTER") != nullptr;   // A/B knob
  static const int min3 = 2;   // measured optimum (was a tuning knob until round 4)
  if (off || padding == PAD_REFLECTION) return ADVCHAIN_ERR_UNSUPPORTED;
  if (d.s2 >= (1 << 23) || (int64_t)d.s0 * d.s1 >= (1 << 23)) return ADVCHAIN_ERR_UNSUPPORTED;   // 24-bit index products
  if (self ? C != ndim : (C != 1 && C != 2 && C != 4)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (ndim == 3 && (halo < 0 ? -halo : halo) < min3) return ADVCHAIN_ERR_UNSUPPORTED;
  const bool det = advchain_get_deterministic() != 0 && det_ws != nullptr;
  const int64_t V = d.voxels();
  unsigned long long* acc64 = nullptr;
  float* maxn = nullptr;
  if (det) {
    // [header 4][2 N V][int64 image: N x C x V (room for 4 channels)][max per entry: N]
    acc64 = reinterpret_cast<unsigned long long*>(det_ws + 4 + 2 * N * V);
    maxn = reinterpret_cast<float*>(det_ws + 4 + 2 * N * V + 8 * N * V);
    advchain_zero_async(acc64, sizeof(unsigned long long) * N * C * V, st);
    advchain_zero_async(maxn, sizeof(float) * N, st);
    const int64_t per_n = C * V;
    int64_t nb = (per_n + kBlock * 16 - 1) / (kBlock * 16);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_det_absmax, dim3((unsigned)nb, (unsigned)N), dim3(kBlock), 0, st, gout, maxn, per_n);
  } else {
    advchain_zero_async(gin, sizeof(float) * N * C * V, st);
  }
  const bool gg = ggrid != nullptr;
  dim3 b(kBlock);
#define GO_PAD(C_, SELF_, GG_) \
  do { if (padding == PAD_BORDER) GO(PAD_BORDER, C_, SELF_, GG_); else GO(PAD_ZEROS, C_, SELF_, GG_); } while (0)
#define GO_ALL(CS_) \
  do { \
    if (self) GO(PAD_BORDER, CS_, true, false); \
    else if (C == 1) { if (gg) GO_PAD(1, false, true); else GO_PAD(1, false, false); } \
    else if (C == 2) { if (gg) GO_PAD(2, false, true); else GO_PAD(2, false, false); } \
    else { if (gg) GO_PAD(4, false, true); else GO_PAD(4, false, false); } \
  } while (0)
  if (ndim == 2) {
    const int n2 = (d.s2 + kWinT - 1) / kWinT, n1 = (d.s1 + kWinT - 1) / kWinT;
    dim3 g((unsigned)(n1 * n2), (unsigned)N);
#define GO(PAD_, C_, SELF_, GG_) \
  do { \
    if (det) hipLaunchKernelGGL((k_scatter_window2d<PAD_, C_, SELF_, GG_, true>), g, b, 0, st, gout, in, grid, gin, ggrid, d, n2, clamp_grid, workspace, acc64, maxn); \
    else hipLaunchKernelGGL((k_scatter_window2d<PAD_, C_, SELF_, GG_, false>), g, b, 0, st, gout, in, grid, gin, ggrid, d, n2, clamp_grid, workspace, acc64, maxn); \
  } while (0)
    GO_ALL(2);
#undef GO
  } else {
    // tile of samples: 32 x 8 x 4, or 16 x 16 x 4 for rows that 32 does not divide (cfg-5's rows of 80: a third of the x tiles
    // would be half empty; 88.2 -> 86.9 ms per cfg-5 call from the flat flush, -> 86.1 with this shape; 16 x 16 x 8 the same
    // on the solver's fields and worse on rough ones, whose windows outgrow the LDS budget).  Rows of 64: 32 x 8 x 4 stays
    // (cfg-3 14.41 against 14.58 ms)
    static const int shape = getenv("ADVCHAIN_WIN3_SHAPE") ? atoi(getenv("ADVCHAIN_WIN3_SHAPE")) : -1;   // A/B knob: 0 | 1
    const int sh = shape >= 0 ? (shape != 0) : ((d.s2 % 32 != 0 && d.s2 % 16 == 0) ? 1 : 0);
    const int TXs = sh == 0 ? 32 : 16, TYs = kBlock / TXs, TZs = 4;
    const int n2 = (d.s2 + TXs - 1) / TXs, n1 = (d.s1 + TYs - 1) / TYs, n0 = (d.s0 + TZs - 1) / TZs;
    dim3 g((unsigned)(n0 * n1 * n2), (unsigned)N);
#define GO3(PAD_, C_, SELF_, GG_, TX_, DET_) \
  hipLaunchKernelGGL((k_scatter_window3d<PAD_, C_, SELF_, GG_, TX_, 4, DET_>), g, b, 0, st, gout, in, grid, gin, ggrid, d, n1, n2, clamp_grid, workspace, acc64, maxn)
#define GO(PAD_, C_, SELF_, GG_) \
  do { \
    if (sh == 0) { if (det) GO3(PAD_, C_, SELF_, GG_, 32, true); else GO3(PAD_, C_, SELF_, GG_, 32, false); } \
    else { if (det) GO3(PAD_, C_, SELF_, GG_, 16, true); else GO3(PAD_, C_, SELF_, GG_, 16, false); } \
  } while (0)
    GO_ALL(3);
#undef GO
#undef GO3
  }
#undef GO_ALL
#undef GO_PAD
  if (det) {
    const int64_t per_n = C * V;
    int64_t nb = (per_n + kBlock * 8 - 1) / (kBlock * 8);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_det_convert, dim3((unsigned)nb, (unsigned)N), dim3(kBlock), 0, st, reinterpret_cast<const long long*>(acc64),
                       maxn, gin, per_n);
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
