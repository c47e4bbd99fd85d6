// LDS-staged forward sampler (gfx950): advchain_compose_self_fwd and advchain_grid_sample_fwd for near-identity
// warps at equal input/output size.
//
// Measured rule on MI355X (profiles/): a vector-memory instruction costs a CU ~26 clk (dword) to
// ~47 clk (dwordx4) whatever it carries, so a kernel that gathers 2^d corners x C channels per voxel from global
// memory is bound by its instruction count (compose_self_fwd: 30 per voxel-wave = 1.3 TB/s), not by bytes.
// Here a workgroup stages the input tile plus a halo into LDS with 16-byte loads (each line is fetched once per
// workgroup), then every thread takes its taps from LDS (ds_read, ~100 clk latency instead of ~2 us) and writes
// 4 consecutive voxels with one 16-byte store.  Corners outside the staged region (displacement beyond the halo)
// fall back to global gathers, so results do not depend on the halo size.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

constexpr int kMaxStageRows = 256;   // rows of the staged region covered by the per-workgroup offset table

struct GTile {
  int t0, t1, t2;   // output tile (z, y, x); t2 % 4 == 0
  int h0, h1, h2;   // halo; h2 % 4 == 0
  int n0, n1, n2;   // tiles per axis
  int rw;           // staged row pitch (floats), multiple of 4
};

template <int DIM, int PAD, int C, bool SELF>
__global__ void __launch_bounds__(kBlock)
k_sample_tiled(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out,
               const float* __restrict__ phi0, Dims d, GTile tc, int clamp_grid, int final_mode,
               float* __restrict__ disp_out, int unaligned) {
  extern __shared__ float lds[];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  int b = blockIdx.x;
  const int tx = b % tc.n2; b /= tc.n2;
  const int ty = b % tc.n1;
  const int tz = b / tc.n1;
  const int x0 = tx * tc.t2, y0 = ty * tc.t1, z0 = tz * tc.t0;
  // staged region, clipped to the volume (rx0 is a multiple of 4; so is rx1 unless S2 is not: partial last quad)
  const int rx0 = max(x0 - tc.h2, 0), rx1 = min(x0 + tc.t2 + tc.h2, d.s2);
  const int ry0 = max(y0 - tc.h1, 0), ry1 = min(y0 + tc.t1 + tc.h1, d.s1);
  const int rz0 = max(z0 - tc.h0, 0), rz1 = min(z0 + tc.t0 + tc.h0, d.s0);
  const int rw = tc.rw, rh = ry1 - ry0, rd = rz1 - rz0;
  const int rw4 = (rx1 - rx0 + 3) >> 2;
  const int plane = rw * rh * rd;       // floats per staged channel
  const int rhw = rh * rw;
  const float* inn = in + (int64_t)n * C * V;
  // ---- stage: C channels x rd x rh rows of (rx1-rx0) floats, 16 bytes per lane
  // ---- stage: C channels x rd x rh rows of (rx1-rx0) floats, 16 bytes per lane.  Staged row R = (c * rd + lz) * rh + ly
  // sits at lds + R * rw; its global offset comes from a table built once per workgroup, and a thread keeps its quad
  // column: no integer division per staged quad (three of them were 31 % of this kernel's VALU instructions)
  const int rows = C * rd * rh;
  __shared__ long long s_row[kMaxStageRows];
  const bool table = rows <= kMaxStageRows;
  if (table) {
    for (int R = threadIdx.x; R < rows; R += kBlock) {
      const int r2 = R / rh, ly = R - r2 * rh;
      const int c = r2 / rd, lz = r2 - c * rd;
      s_row[R] = (long long)c * V + ((rz0 + lz) * d.s1 + (ry0 + ly)) * d.s2 + rx0;
    }
    __syncthreads();
    const int rsub = threadIdx.x / rw4, q = threadIdx.x - rsub * rw4;
    const int rstep = kBlock / rw4;
    if (rsub < rstep) {
      if (!unaligned) {
        for (int R = rsub; R < rows; R += rstep)
          *reinterpret_cast<float4*>(lds + R * rw + 4 * q) = *reinterpret_cast<const float4*>(inn + s_row[R] + 4 * q);
      } else {
        // rows not 16-byte aligned (S2 % 4 != 0 or a misaligned base): dword loads, partial last quad padded with zeros
        const int left = rx1 - rx0 - 4 * q;
        for (int R = rsub; R < rows; R += rstep) {
          const float* src = inn + s_row[R] + 4 * q;
          float4 v;
          v.x = src[0];
          v.y = left > 1 ? src[1] : 0.f;
          v.z = left > 2 ? src[2] : 0.f;
          v.w = left > 3 ? src[3] : 0.f;
          *reinterpret_cast<float4*>(lds + R * rw + 4 * q) = v;
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < rows * rw4; e += kBlock) {
      const int q = e % rw4;
      const int R = e / rw4;
      const int r2 = R / rh, ly = R - r2 * rh;
      const int c = r2 / rd, lz = r2 - c * rd;
      const float* src = inn + (int64_t)c * V + ((rz0 + lz) * d.s1 + (ry0 + ly)) * d.s2 + rx0 + 4 * q;
      const int left = unaligned ? rx1 - rx0 - 4 * q : 4;
      float4 v;
      if (!unaligned) {
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        v.y = left > 1 ? src[1] : 0.f;
        v.z = left > 2 ? src[2] : 0.f;
        v.w = left > 3 ? src[3] : 0.f;
      }
      *reinterpret_cast<float4*>(lds + R * rw + 4 * q) = v;
    }
  }
  __syncthreads();
  // ---- compute: lanes run along x and wrap to the next row of the tile (neighbouring lanes read neighbouring LDS
  // words -- with 4 consecutive x per thread the taps of a wave sat 4 words apart and every LDS read was a 4-way
  // bank conflict; with one wave per row a row of 80 voxels left 48 of 128 lanes idle)
  const float* gn = SELF ? nullptr : grid + (int64_t)n * DIM * V;
  float* on = out + (int64_t)n * C * V;
  float dmax = 0.f;
  const int tw = min(tc.t2, d.s2 - x0), th = min(tc.t1, d.s1 - y0), td = min(tc.t0, d.s0 - z0);   // clipped tile
  const int items = tw * th * td;
  const int step_r = kBlock / tw, step_x = kBlock - step_r * tw;
  int lx, ly, lz;
  {
    const int r = threadIdx.x / tw;
    lx = threadIdx.x - r * tw;
    lz = r / th;
    ly = r - lz * th;
  }
  for (int i = threadIdx.x; i < items; i += kBlock) {
    {
      const int sx = x0 + lx, sy = y0 + ly, sz = z0 + lz;
      const int s = __mul24(__mul24(sz, d.s1) + sy, d.s2) + sx;     // launcher: rows and row length below 2^23
      float gx, gy, gz = 0.f;
      if (SELF) {
        const int lo = __mul24(sz - rz0, rhw) + __mul24(sy - ry0, rw) + (sx - rx0);
        gx = lds[lo];
        gy = lds[plane + lo];
        if (DIM == 3) gz = lds[2 * plane + lo];
      } else {
        gx = gn[s];
        gy = gn[(int64_t)V + s];
        if (DIM == 3) gz = gn[2 * (int64_t)V + s];
      }
      if (clamp_grid) { gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz); }
      // final mode: phi0 of all channels requested here, together (one memory round trip, overlapped with the taps;
      // loaded where they are used they were three serial ones)
      float p0v[C];
      if (SELF && final_mode == 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) p0v[c] = phi0[((int64_t)n * DIM + c) * V + s];
      }
      Taps<DIM, PAD> t;
      t.build(gx, gy, gz, d);
      // corner box clamped to the volume; inside the staged region?
      const int cx0 = max(t.x.i0, 0), cx1 = min(t.x.i0 + 1, d.s2 - 1);
      const int cy0 = max(t.y.i0, 0), cy1 = min(t.y.i0 + 1, d.s1 - 1);
      const int cz0 = DIM == 3 ? max(t.z.i0, 0) : 0, cz1 = DIM == 3 ? min(t.z.i0 + 1, d.s0 - 1) : 0;
      const bool staged = (cx0 >= rx0) && (cx1 < rx1) && (cy0 >= ry0) && (cy1 < ry1) && (cz0 >= rz0) && (cz1 < rz1) &&
                          (cx0 <= cx1) && (cy0 <= cy1) && (cz0 <= cz1);
      float res[C];
      if (staged) {
        // tile-local offsets: 24-bit multiplies (full rate; a 32-bit v_mul_lo costs four VALU slots)
        const int ox[2] = {cx0 - rx0, cx1 - rx0};
        const int oy[2] = {__mul24(cy0 - ry0, rw), __mul24(cy1 - ry0, rw)};
        const int oz[2] = {__mul24(cz0 - rz0, rhw), __mul24(cz1 - rz0, rhw)};
        float w[8];
#pragma unroll
        for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx) {
              // border padding: the coordinate is clipped into the volume, so a corner beyond it carries weight 0
              // already (its index is clamped above): no validity select
              const float wt = t.w(cz, cy, cx);
              w[(cz * 2 + cy) * 2 + cx] = (PAD == PAD_BORDER || t.ok(cz, cy, cx)) ? wt : 0.f;
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* p = lds + c * plane;
          float acc = 0.f;
#pragma unroll
          for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
              for (int cx = 0; cx < 2; ++cx)     // see tap_acc() (sampler_common.h): the same rounding in every forward kernel
                acc = tap_acc<DIM>(acc, p[oz[cz] + oy[cy] + ox[cx]], w[(cz * 2 + cy) * 2 + cx]);
          res[c] = acc;
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) res[c] = sample_linear<DIM, PAD>(inn + (int64_t)c * V, t, d);
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float o = res[c];
        if (SELF && final_mode == 1) {
          // (sample - phi0) + identity   (adv_morph.py:143,176 + 474,483)
          const float p0 = p0v[c];
          o = (o - p0) + (c == 0 ? lin_coord(sx, d.s2) : (c == 1 ? lin_coord(sy, d.s1) : lin_coord(sz, d.s0)));
        }
        on[(int64_t)c * V + s] = o;
        if (SELF && disp_out) {   // displacement of the composed field (the next squaring's input), for its backward
          dmax = fmaxf(dmax, voxel_displacement(o, c == 0 ? d.s2 : (c == 1 ? d.s1 : d.s0), c == 0 ? sx : (c == 1 ? sy : sz)));
        }
      }
    }
    lx += step_x;
    ly += step_r;
    if (lx >= tw) { lx -= tw; ++ly; }
    while (ly >= th) { ly -= th; ++lz; }
  }
  if (SELF && disp_out) wave_max_to_slots(dmax, disp_out);
}

}  // namespace advchain

using namespace advchain;

static bool choose_gtile(int ndim, const Dims& d, int C, int halo_hint, GTile& tc) {
  static const int h3 = 1;   // measured optimum (was a tuning knob until round 4)
  static const int h2 = 8;
  if (d.s2 < 8 || d.s2 >= (1 << 23) || (int64_t)d.s0 * d.s1 >= (1 << 23)) return false;   // 24-bit index products
  static const bool tiles_2d = false;
  if (ndim == 2 && !tiles_2d) return false;   // measured: in 2D (4 corners) the direct 4-chain gather kernels are faster
  int h = ndim == 3 ? h3 : h2;
  if (halo_hint > 0) h = halo_hint;
  static const int t0_3 = 2;   // measured: small tiles (more resident workgroups) beat low halo amplification
  static const int t1_3 = 8;
  static const int t1_2 = 16;
  if (ndim == 3) { tc.t0 = t0_3; tc.t1 = t1_3; tc.h0 = tc.h1 = h; }
  else { tc.t0 = 1; tc.t1 = t1_2; tc.h0 = 0; tc.h1 = h; }
  if (d.s2 <= 96) { tc.t2 = d.s2; tc.h2 = 0; }     // whole rows: no x halo
  else { tc.t2 = 64; tc.h2 = (h + 3) / 4 * 4; }
  if (tc.t1 > d.s1) tc.t1 = d.s1;
  if (tc.t0 > d.s0) tc.t0 = d.s0;
  auto bytes = [&]() {
    return (int64_t)C * (tc.t0 + 2 * tc.h0) * (tc.t1 + 2 * tc.h1) * ((tc.t2 + 2 * tc.h2 + 3) / 4 * 4) * 4;
  };
  while (bytes() > 65536) {
    if (tc.t0 > 1) tc.t0 = (tc.t0 + 1) / 2;
    else if (tc.t1 > 2) tc.t1 = (tc.t1 + 1) / 2;
    else return false;
  }
  tc.rw = (tc.t2 + 2 * tc.h2 + 3) / 4 * 4;
  tc.n2 = (d.s2 + tc.t2 - 1) / tc.t2;
  tc.n1 = (d.s1 + tc.t1 - 1) / tc.t1;
  tc.n0 = (d.s0 + tc.t0 - 1) / tc.t0;
  return true;
}

template <int DIM, int PAD, bool SELF>
static bool launch_sample_c(int C, dim3 g, size_t lds, hipStream_t st, const float* in, const float* grid, float* out,
                            const float* phi0, Dims d, GTile tc, int clamp_grid, int final_mode, float* disp_out, int unaligned) {
#define LAUNCH(C_) hipLaunchKernelGGL((k_sample_tiled<DIM, PAD, C_, SELF>), g, dim3(kBlock), lds, st, in, grid, out, phi0, d, tc, clamp_grid, final_mode, disp_out, unaligned)
  switch (C) {
    case 1: if constexpr (!SELF) { LAUNCH(1); return true; } return false;
    case 2: if constexpr (!SELF || DIM == 2) { LAUNCH(2); return true; } return false;
    case 3: if constexpr (!SELF || DIM == 3) { LAUNCH(3); return true; } return false;
    case 4: if constexpr (!SELF) { LAUNCH(4); return true; } return false;
    default: return false;
  }
#undef LAUNCH
}

// sample_march.hip: z-marching form (3D, rows of at most 64 voxels)
int advchain_sample_march_launch(bool self, const float* in, const float* grid, float* out, const float* phi0, int64_t N,
                                 int64_t C, Dims d, int padding, int clamp_grid, int final_mode, float* disp_out,
                                 hipStream_t st);

// sample_ring.hip: z-marching form with a ring of 2H+2 planes (3D image warps, C == 1, hints of 2 .. 4 voxels)
int advchain_sample_ring_launch(const float* in, const float* grid, float* out, int64_t N, Dims d, int padding, int clamp_grid,
                                int hint, hipStream_t st);

// Returns ADVCHAIN_ERR_UNSUPPORTED when the shape does not qualify (caller uses the direct-gather kernels).
// The z-marching kernel keeps three planes of its rows in LDS and sends a lane whose taps leave them to global gathers:
// unbeatable below one voxel (self-composition 29 against 42 us at 4 x 3 x 128 x 128 x 64), equal at 1-2 voxels, slower
// beyond (97 against 63 us at 4 voxels, 117 against 83 at 7).  Rows longer than 64 voxels run as two or more partly
// empty x segments: at 160 x 160 x 80 the tile kernel wins at every displacement for C = 3 / C = 1 (83 against 87 us
// below a voxel, 116 against 465 at 4 voxels).  `hint` = the caller's displacement estimate in voxels, rounded up
// (0 = unknown): a performance hint, results do not depend on it.
static bool march_forward_pays(bool self, int64_t C, const Dims& d, int hint) {
  static const bool always = false;   // measured optimum (was a tuning knob until round 4): the round-2 policy before the hint
  if (always) return true;
  static const bool no_flat = getenv("ADVCHAIN_NO_FLAT_FWD") != nullptr;   // A/B knob
  static const int flat_hint_max = 1;   // measured optimum (was a tuning knob until round 4)
  if (d.s2 > 64 && d.s2 <= 128 && !no_flat) return hint <= flat_hint_max;   // lane <-> flat voxel (k_sample_march_flat): no idle lanes
  if (d.s2 > 64) return !self && C == 4 && hint <= 1;
  return hint <= 1;
}

int advchain_sample_tiled_launch(bool self, const float* in, const float* grid, float* out, const float* phi0,
                                 int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid, int final_mode,
                                 int halo, float* disp_out, hipStream_t st, int disp_hint) {
  static const bool off = getenv("ADVCHAIN_NO_GATHER_TILES") != nullptr;   // A/B knob
  if (off || C < 1 || C > 4) return ADVCHAIN_ERR_UNSUPPORTED;
  if (ndim == 3 && march_forward_pays(self, C, d, disp_hint)) {
    const int rc = advchain_sample_march_launch(self, in, grid, out, phi0, N, C, d, padding, clamp_grid, final_mode, disp_out, st);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
  if (ndim == 3 && !self && C == 1 && disp_hint >= 2) {   // 2 .. 4 voxels: the ring of 2H+2 planes (sample_ring.hip)
    const int rc = advchain_sample_ring_launch(in, grid, out, N, d, padding, clamp_grid, disp_hint, st);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
  const int unaligned = (reinterpret_cast<uintptr_t>(in) & 15) != 0 || (d.s2 & 3) != 0;   // only `in` is read 16 bytes at a time
  GTile tc;
  if (!choose_gtile(ndim, d, (int)C, halo, tc)) return ADVCHAIN_ERR_UNSUPPORTED;
  const size_t lds = (size_t)C * (tc.t0 + 2 * tc.h0) * (tc.t1 + 2 * tc.h1) * tc.rw * sizeof(float);
  dim3 g((unsigned)(tc.n0 * tc.n1 * tc.n2), (unsigned)N);
  bool ok = false;
  if (self) {
    ok = ndim == 3 ? launch_sample_c<3, PAD_BORDER, true>((int)C, g, lds, st, in, grid, out, phi0, d, tc, 0, final_mode, disp_out, unaligned)
                   : launch_sample_c<2, PAD_BORDER, true>((int)C, g, lds, st, in, grid, out, phi0, d, tc, 0, final_mode, disp_out, unaligned);
  } else if (ndim == 3) {
    switch (padding) {
      case PAD_ZEROS: ok = launch_sample_c<3, PAD_ZEROS, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
      case PAD_BORDER: ok = launch_sample_c<3, PAD_BORDER, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
      default: ok = launch_sample_c<3, PAD_REFLECTION, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
    }
  } else {
    switch (padding) {
      case PAD_ZEROS: ok = launch_sample_c<2, PAD_ZEROS, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
      case PAD_BORDER: ok = launch_sample_c<2, PAD_BORDER, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
      default: ok = launch_sample_c<2, PAD_REFLECTION, false>((int)C, g, lds, st, in, grid, out, phi0, d, tc, clamp_grid, 0, nullptr, unaligned); break;
    }
  }
  if (!ok) return ADVCHAIN_ERR_UNSUPPORTED;
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
