// Gather-form adjoint of the self-composition  phi -> phi o phi  for sub-H-voxel displacements (gfx950).
//
// The backward of grid_sample is a scatter-add: sample s deposits w * grad_out[s] on the 2^d corners of its sampling
// position p_s.  When |p_s - s| < H voxels on every axis (the early squarings of adv_morph.py:132-135,165-168 compose
// fields whose displacement is 2^-k of the total) every corner of s lies within s +- H, so the same sum can be
// written from the receiving side with the (bi/tri)linear weight in its tent form:
//
//     grad_phi_c[u] = sum_{s in u +- H} grad_out_c[s] * prod_a max(0, 1 - |p_s,a - u_a|)  +  (coordinate path at s = u)
//
// No atomics, no fixed point, fixed summation order.  A workgroup stages phi and grad_out of its tile plus an H-halo
// in LDS (16 bytes per lane), then
//   stage    phi is stored as the offset o = unnormalize(phi) - s of the sampling position from the sample's own voxel
//            (phi itself is recoverable from it); samples whose clipped offset leaves -H <= f < H ("irregular": large
//            displacement, NaN) get grad_out = 0 in LDS, and their owner appends them to the overflow list;
//   phase A  every owned sample takes the 2^d corner values of the field from LDS and forms the coordinate-path
//            gradient (the part ATen calls grad_grid) in registers;
//   phase B  lane <-> x: for every owned output row the wave walks the (2H+1)^(d-1) neighbouring sample rows; each
//            lane keeps 2H+1 partial sums (its sample's deposits on x-H..x+H), which 2H whole-wave DPP shifts fold
//            into the owning lanes at the end.
// The overflow list (normally empty) is drained by a second launch with global atomics, all corners of a sample.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

constexpr int kXPad = 4;   // x halo of a 64-lane row when the row is wider than one wave (multiple of 4, >= H)

__global__ void k_gather_prepare(int32_t* ws, int chain) {
  ws[2] = chain ? ws[3] : 0;
  ws[0] = 0;
  ws[1] = 0;
  ws[3] = 0;
}

// weight of a sample at offset f (= p - s) for the output at s + E
template <int E>
__device__ __forceinline__ float tent(float f) { return fmaxf(0.f, 1.f - fabsf(f - (float)E)); }

template <int H, int K = -H>
struct XSpread {
  // acc[c][K + H] += a[c] * tent<K>(fx) for K = -H..H
  template <int C>
  static __device__ __forceinline__ void run(float (&acc)[C][2 * H + 1], const float (&a)[C], float fx) {
    const float w = tent<K>(fx);
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c][K + H] = fmaf(a[c], w, acc[c][K + H]);
    if constexpr (K < H) XSpread<H, K + 1>::template run<C>(acc, a, fx);
  }
};

// flags
constexpr int kClip = 1;        // sampling positions are clipped to [0, S-1] (border padding and/or clamp_grid)
constexpr int kBorder = 2;      // border padding: zero coordinate gradient AT and beyond the border (else: beyond only)
constexpr int kClampGrid = 4;   // the caller asked for clamp(grid, -1, 1) (only the fallback path needs to know)
constexpr int kUnaligned = 8;   // rows are not 16-byte aligned (S2 % 4 != 0 or a misaligned base): stage with dword loads

template <int DIM, int C, int H, bool SELF, bool GG, int TZ, int TY, int NT>
struct GatherCfg {
  static constexpr int RZ = DIM == 3 ? TZ + 2 * H : 1;
  static constexpr int RY = TY + 2 * H;
  static constexpr int ROWS = RZ * RY;
  static constexpr int NW = NT / 64;
  static constexpr int OWNED = (DIM == 3 ? TZ : 1) * TY;
  static constexpr int RPW = OWNED / NW;
  static constexpr bool STAGE_IN = GG && !SELF;
  static constexpr int CH = DIM + C + (STAGE_IN ? C : 0);
  static constexpr size_t LDS = (size_t)CH * ROWS * 64 * sizeof(float);
  static_assert(OWNED % NW == 0, "owned rows must divide evenly among the waves");
  static_assert(!SELF || C == DIM, "self-composition carries DIM channels");
};

// unnormalised source coordinate of a normalised grid value (GridSampler.h grid_sampler_unnormalize, align_corners)
__device__ __forceinline__ float unnormalize(float g, int S) { return ((g + 1.f) * 0.5f) * (float)(S - 1); }

// coordinate-path gradient of one sample straight from global memory (irregular samples only)
template <int DIM, int PAD, int C>
__device__ __forceinline__ void coord_grad_global(const float* __restrict__ inn, const float* __restrict__ gn,
                                                  const float* __restrict__ gon, int s, int V, const Dims& d,
                                                  int clamp_grid, float (&gg)[3]) {
  float gx = gn[s], gy = gn[V + s], gz = DIM == 3 ? gn[2 * V + s] : 0.f;
  bool px = true, py = true, pz = true;
  if (clamp_grid) {
    px = gx >= -1.f && gx <= 1.f; py = gy >= -1.f && gy <= 1.f; pz = gz >= -1.f && gz <= 1.f;
    gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz);
  }
  Taps<DIM, PAD> t;
  t.build(gx, gy, gz, d);
  float ax = 0.f, ay = 0.f, az = 0.f, dummy = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c)
    sample_linear_bwd<DIM, PAD, false, true>(inn + (int64_t)c * V, nullptr, gon[(int64_t)c * V + s], t, d, ax, ay,
                                             DIM == 3 ? az : dummy);
  gg[0] = px ? t.x.mult * ax : 0.f;
  gg[1] = py ? t.y.mult * ay : 0.f;
  gg[2] = (DIM == 3 && pz) ? t.z.mult * az : 0.f;
}

// SELF : in == grid == phi (C == DIM); gin receives value path + coordinate path      (advchain_compose_self_bwd)
// !SELF: gin <- value path; GG: ggrid <- coordinate path (needs `in`, staged as well)  (advchain_grid_sample_bwd)
// (3D: two 8-wave workgroups per CU = 4 waves per SIMD = at most 128 VGPRs)
template <int DIM, int C, int H, bool SELF, bool GG, int TZ, int TY, int NT>
__global__ void __launch_bounds__(NT, (DIM == 3 && NT == 512) ? 4 : 1)
k_adjoint_gather(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                 float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int n2, int wide, int flags,
                 float* __restrict__ absmax_out, int* __restrict__ ovf_count, int2* __restrict__ ovf_list, int ovf_cap,
                 int32_t* __restrict__ untracked) {
  using G = GatherCfg<DIM, C, H, SELF, GG, TZ, TY, NT>;
  // strict launches of a chain do not track max|result|: say so in the workspace header (see scatter_tiled.hip)
  if (untracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) untracked[3] = -1;
  constexpr int RY = G::RY, ROWS = G::ROWS, NW = G::NW, RPW = G::RPW;
  constexpr int OG = DIM, OI = DIM + C;   // first grad_out / input channel in LDS
  // [CH][ROWS][64]: channels 0..DIM-1 hold o = unnormalize(grid) - s (unclipped offset of the sampling position from
  // the sample's own voxel, in voxels), then C channels grad_out (0 for irregular samples), then (STAGE_IN) C of `in`
  extern __shared__ float lds[];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b = blockIdx.x;
  const int tx = b % n2; b /= n2;
  const int ty = b % n1;
  const int tz = b / n1;
  const int xown = wide ? 64 - 2 * kXPad : 64;
  const int x0 = tx * xown, y0 = ty * TY, z0 = DIM == 3 ? tz * TZ : 0;
  const int rx0 = wide ? x0 - kXPad : 0;          // x of lane 0 (multiple of 4)
  const int ry0 = y0 - H, rz0 = DIM == 3 ? z0 - H : 0;
  const float* gn = grid + (int64_t)n * DIM * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  const int S[3] = {d.s2, d.s1, d.s0};
  const float fH = (float)H;
  // the self-composition always runs clip + border: constants there, run-time flags for the image warps
  const bool clip = SELF ? true : (flags & kClip) != 0, border = SELF ? true : (flags & kBorder) != 0;

  // ---- stage: one item = 4 consecutive x of one region row, all channels (rows / quads outside the volume: zeros)
  for (int e = threadIdx.x; e < ROWS * 16; e += NT) {
    const int q = e & 15;
    const int r = e >> 4;
    const int sy = ry0 + r % RY, sz = rz0 + r / RY;
    const int x = rx0 + 4 * q;
    float o[DIM][4], g[C][4], vin[G::STAGE_IN ? C : 1][4];
    const bool inside = sy >= 0 && sy < d.s1 && sz >= 0 && sz < d.s0 && x >= 0 && x < d.s2;
    if (inside) {
      const int s = (sz * d.s1 + sy) * d.s2 + x;
      if (!(flags & kUnaligned)) {
#pragma unroll
        for (int a = 0; a < DIM; ++a) load_vec<4>(gn + (int64_t)a * V + s, o[a]);
#pragma unroll
        for (int c = 0; c < C; ++c) load_vec<4>(gon + (int64_t)c * V + s, g[c]);
        if (G::STAGE_IN) {
#pragma unroll
          for (int c = 0; c < C; ++c) load_vec<4>(inn + (int64_t)c * V + s, vin[c]);
        }
      } else {
        // the last quad of a row may be partial: what lies beyond the row is the next row's, staged as zeros
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool in_row = x + k < d.s2;
          const int sk = in_row ? s + k : s;
#pragma unroll
          for (int a = 0; a < DIM; ++a) o[a][k] = gn[(int64_t)a * V + sk];
#pragma unroll
          for (int c = 0; c < C; ++c) g[c][k] = in_row ? gon[(int64_t)c * V + sk] : 0.f;
          if (G::STAGE_IN) {
#pragma unroll
            for (int c = 0; c < C; ++c) vin[c][k] = in_row ? inn[(int64_t)c * V + sk] : 0.f;
          }
        }
      }
      const int sc[3] = {x, sy, sz};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool regular = true;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
          const float xs = unnormalize(o[a][k], S[a]);
          const float sa = (float)(sc[a] + (a == 0 ? k : 0));
          const float p = clip ? fminf(fmaxf(xs, 0.f), (float)(S[a] - 1)) : xs;
          const float f = p - sa;
          regular = regular && (f >= -fH) && (f < fH);                        // false for NaN
          o[a][k] = (xs > -1.0e9f && xs < 1.0e9f) ? xs - sa : 0.f;
        }
        if (!regular) {
#pragma unroll
          for (int c = 0; c < C; ++c) g[c][k] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int a = 0; a < DIM; ++a) o[a][k] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) g[c][k] = 0.f;
        if (G::STAGE_IN) {
#pragma unroll
          for (int c = 0; c < C; ++c) vin[c][k] = 0.f;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a) store_vec<4>(lds + (a * ROWS + r) * 64 + 4 * q, o[a]);
#pragma unroll
    for (int c = 0; c < C; ++c) store_vec<4>(lds + ((OG + c) * ROWS + r) * 64 + 4 * q, g[c]);
    if (G::STAGE_IN) {
#pragma unroll
      for (int c = 0; c < C; ++c) store_vec<4>(lds + ((OI + c) * ROWS + r) * 64 + 4 * q, vin[c]);
    }
  }
  __syncthreads();

  const int sx = rx0 + lane;
  const bool xowned = wide ? (lane >= kXPad && lane < 64 - kXPad && sx < d.s2) : (sx < d.s2);
  const float xlo = -(float)sx, xhi = (float)(d.s2 - 1 - sx);   // clip bounds of a sample in this lane, as offsets
  float* ginn = gin + (int64_t)n * C * V;
  float m = 0.f;
  // 3D, H = 1: a wave owns two rows ADJACENT in y and runs phase B for both at once -- they share 6 of their 9
  // neighbour rows each, so a staged row is read, clipped and given its x / z tents once for the pair
  constexpr bool PAIR = DIM == 3 && H == 1 && RPW == 2 && TY % 2 == 0;
  float ggs[PAIR ? 2 : 1][3];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int o = PAIR ? RPW * wave + j : wave + j * NW;
    const int ly = o % TY, lz = o / TY;
    const int uy = y0 + ly, uz = z0 + lz;
    if (uy >= d.s1 || uz >= d.s0) continue;   // wave-uniform
    const int rc = (DIM == 3 ? (lz + H) * RY : 0) + ly + H;
    const int s = (uz * d.s1 + uy) * d.s2 + sx;

    // ---- phase A: coordinate-path gradient of the sample at this output position
    float gg[3] = {0.f, 0.f, 0.f};
    if ((SELF || GG) && xowned) {
      const int sc[3] = {sx, uy, uz};
      float w1[3] = {0.f, 0.f, 0.f}, mult[3] = {0.f, 0.f, 0.f};
      int i0[3] = {0, 0, 0};
      bool regular = true;
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        float xs = lds[(a * ROWS + rc) * 64 + lane] + (float)sc[a];
        const float top = (float)(S[a] - 1);
        mult[a] = 0.5f * top;
        if (clip) {
          if (border ? xs <= 0.f : xs < 0.f) mult[a] = 0.f;
          if (border ? xs >= top : xs > top) mult[a] = 0.f;
          xs = fminf(fmaxf(xs, 0.f), top);
        }
        const float fl = floorf(xs);
        i0[a] = (int)fl;
        w1[a] = xs - fl;
        const float f = xs - (float)sc[a];
        regular = regular && (f >= -fH) && (f < fH);
      }
      if (regular) {
        float go[C];
#pragma unroll
        for (int c = 0; c < C; ++c) go[c] = lds[((OG + c) * ROWS + rc) * 64 + lane];
        // a corner outside the volume reads as 0 (zeros padding skips it; under border padding / clamp it carries
        // weight 0 or multiplier 0).  Rows outside the volume are staged as zeros; only x needs the select.
        const bool okx0 = i0[0] >= 0, okx1 = i0[0] + 1 < d.s2;
        const int lx0 = max(i0[0], 0) - rx0, lx1 = min(i0[0] + 1, d.s2 - 1) - rx0;
        const int r00 = (DIM == 3 ? (i0[2] - rz0) * RY : 0) + (i0[1] - ry0);
        const float wx1 = w1[0], wx0 = 1.f - wx1, wy1 = w1[1], wy0 = 1.f - wy1, wz1 = w1[2], wz0 = 1.f - wz1;
        float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* p = lds + (SELF ? c : OI + c) * ROWS * 64;
          float v[2][2][2];
#pragma unroll
          for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              const int r = r00 + cz * RY + cy;
              v[cz][cy][0] = okx0 ? p[r * 64 + lx0] : 0.f;
              v[cz][cy][1] = okx1 ? p[r * 64 + lx1] : 0.f;
            }
          // SELF: phi_c(u) = (o_c(u) + u_c) * 2/(S_c-1) - 1: only differences along an axis enter; the identity part
          // of a difference along axis c is the index step (1)
          const float ux = (SELF && c == 0) ? 1.f : 0.f, uyy = (SELF && c == 1) ? 1.f : 0.f, uzz = (SELF && c == 2) ? 1.f : 0.f;
          float dx, dy, dz = 0.f;
          if (DIM == 3) {
            dx = ((v[0][0][1] - v[0][0][0] + ux) * wy0 + (v[0][1][1] - v[0][1][0] + ux) * wy1) * wz0 +
                 ((v[1][0][1] - v[1][0][0] + ux) * wy0 + (v[1][1][1] - v[1][1][0] + ux) * wy1) * wz1;
            dy = ((v[0][1][0] - v[0][0][0] + uyy) * wx0 + (v[0][1][1] - v[0][0][1] + uyy) * wx1) * wz0 +
                 ((v[1][1][0] - v[1][0][0] + uyy) * wx0 + (v[1][1][1] - v[1][0][1] + uyy) * wx1) * wz1;
            dz = ((v[1][0][0] - v[0][0][0] + uzz) * wx0 + (v[1][0][1] - v[0][0][1] + uzz) * wx1) * wy0 +
                 ((v[1][1][0] - v[0][1][0] + uzz) * wx0 + (v[1][1][1] - v[0][1][1] + uzz) * wx1) * wy1;
          } else {
            dx = (v[0][0][1] - v[0][0][0] + ux) * wy0 + (v[0][1][1] - v[0][1][0] + ux) * wy1;
            dy = (v[0][1][0] - v[0][0][0] + uyy) * wx0 + (v[0][1][1] - v[0][0][1] + uyy) * wx1;
          }
          const float kc = SELF ? go[c] * (2.f / (float)(S[c] - 1)) : go[c];
          acc3[0] = fmaf(dx, kc, acc3[0]); acc3[1] = fmaf(dy, kc, acc3[1]); acc3[2] = fmaf(dz, kc, acc3[2]);
        }
#pragma unroll
        for (int a = 0; a < DIM; ++a) gg[a] = mult[a] * acc3[a];
      } else {
        // irregular sample: taps and corner values from global memory, deposits through the overflow list
        if (border) coord_grad_global<DIM, PAD_BORDER, C>(inn, gn, gon, s, V, d, flags & kClampGrid, gg);
        else coord_grad_global<DIM, PAD_ZEROS, C>(inn, gn, gon, s, V, d, flags & kClampGrid, gg);
      }
    }
    if (xowned && ovf_count) {
      // the owner lists its irregular samples (same test as at staging time: grad_out was zeroed there)
      const int sc[3] = {sx, uy, uz};
      bool regular = true;
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        float xs = lds[(a * ROWS + rc) * 64 + lane] + (float)sc[a];
        if (clip) xs = fminf(fmaxf(xs, 0.f), (float)(S[a] - 1));
        const float f = xs - (float)sc[a];
        regular = regular && (f >= -fH) && (f < fH);
      }
      if (!regular) {
        const int slot = atomicAdd(ovf_count, 1);
        if (slot < ovf_cap) ovf_list[slot] = make_int2(n, s);
      }
    }

    if constexpr (PAIR) {
#pragma unroll
      for (int a = 0; a < 3; ++a) ggs[j][a] = gg[a];
    } else {
    // ---- phase B: gather the deposits of the (2H+1)^d neighbouring samples
    float acc[C][2 * H + 1];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int k = 0; k < 2 * H + 1; ++k) acc[c][k] = 0.f;
#pragma unroll
    for (int dz = (DIM == 3 ? -H : 0); dz <= (DIM == 3 ? H : 0); ++dz)
#pragma unroll
      for (int dy = -H; dy <= H; ++dy) {
        // sample row (uz + dz, uy + dy): this output sits at offset (-dz, -dy) from it
        const int r = rc + (DIM == 3 ? dz * RY : 0) + dy;
        float fx = lds[(0 * ROWS + r) * 64 + lane];
        float fy = lds[(1 * ROWS + r) * 64 + lane];
        float fz = DIM == 3 ? lds[(2 * ROWS + r) * 64 + lane] : 0.f;
        if (clip) {
          fx = __builtin_amdgcn_fmed3f(fx, xlo, xhi);
          fy = __builtin_amdgcn_fmed3f(fy, -(float)(uy + dy), (float)(d.s1 - 1 - uy - dy));     // wave-uniform bounds
          if (DIM == 3) fz = __builtin_amdgcn_fmed3f(fz, -(float)(uz + dz), (float)(d.s0 - 1 - uz - dz));
        }
        float w = fmaxf(0.f, 1.f - fabsf(fy + (float)dy));
        if (DIM == 3) w *= fmaxf(0.f, 1.f - fabsf(fz + (float)dz));
        float a[C];
#pragma unroll
        for (int c = 0; c < C; ++c) a[c] = lds[((OG + c) * ROWS + r) * 64 + lane] * w;
        XSpread<H>::template run<C>(acc, a, fx);
      }
    // fold the x partial sums into the owning lanes: out(x) = sum_k acc[k](lane x - k)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float up = acc[c][2 * H];               // deposits on x + H
#pragma unroll
      for (int k = H - 1; k >= 0; --k) up = lane_prev_f(up) + acc[c][k + H];
      float dn = acc[c][0];                   // deposits on x - H
#pragma unroll
      for (int k = -H + 1; k <= -1; ++k) dn = lane_next_f(dn) + acc[c][k + H];
      dn = lane_next_f(dn);
      float v = up + dn;
      if (SELF) v += gg[c < 3 ? c : 0];
      if (xowned) {
        ginn[(int64_t)c * V + s] = v;
        m = fmaxf(m, fabsf(v));
      }
    }
    if (!SELF && GG && xowned) {
      float* gq = ggrid + (int64_t)n * DIM * V + s;
#pragma unroll
      for (int a = 0; a < DIM; ++a) gq[(int64_t)a * V] = gg[a];
    }
    }   // !PAIR
  }
  if constexpr (PAIR) {
    const int o0 = RPW * wave;
    const int ly0 = o0 % TY, lz = o0 / TY;
    const int uy0 = y0 + ly0, uz = z0 + lz;
    if (uy0 < d.s1 && uz < d.s0) {   // wave-uniform
      const int rc0 = (lz + H) * RY + ly0 + H;
      float acc[2][C][3];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[o][c][k] = 0.f;
#pragma unroll
      for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int yy = -1; yy <= 2; ++yy) {
          // sample row (uz + dz, uy0 + yy): output row uy0 + o sits at offset (-dz, o - yy) from it
          const int r = rc0 + dz * RY + yy;
          float fx = lds[(0 * ROWS + r) * 64 + lane];
          float fy = lds[(1 * ROWS + r) * 64 + lane];
          float fz = lds[(2 * ROWS + r) * 64 + lane];
          if (clip) {
            fx = __builtin_amdgcn_fmed3f(fx, xlo, xhi);
            fy = __builtin_amdgcn_fmed3f(fy, -(float)(uy0 + yy), (float)(d.s1 - 1 - uy0 - yy));   // wave-uniform bounds
            fz = __builtin_amdgcn_fmed3f(fz, -(float)(uz + dz), (float)(d.s0 - 1 - uz - dz));
          }
          const float wz = fmaxf(0.f, 1.f - fabsf(fz + (float)dz));
          const float t[3] = {tent<-1>(fx), tent<0>(fx), tent<1>(fx)};
          float g[C];
#pragma unroll
          for (int c = 0; c < C; ++c) g[c] = lds[((OG + c) * ROWS + r) * 64 + lane] * wz;
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int dy = yy - o;
            if (dy < -1 || dy > 1) continue;   // compile time
            const float wy = fmaxf(0.f, 1.f - fabsf(fy + (float)dy));
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const float a = g[c] * wy;
#pragma unroll
              for (int k = 0; k < 3; ++k) acc[o][c][k] = fmaf(a, t[k], acc[o][c][k]);
            }
          }
        }
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        if (uy0 + o >= d.s1) continue;   // wave-uniform
        const int s = (uz * d.s1 + uy0 + o) * d.s2 + sx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float up = lane_prev_f(acc[o][c][2]) + acc[o][c][1];   // deposits on x + 1, folded into the owner
          float v = up + lane_next_f(acc[o][c][0]);                   // deposits on x - 1
          if (SELF) v += ggs[o][c < 3 ? c : 0];
          if (xowned) {
            ginn[(int64_t)c * V + s] = v;
            m = fmaxf(m, fabsf(v));
          }
        }
        if (!SELF && GG && xowned) {
          float* gq = ggrid + (int64_t)n * DIM * V + s;
#pragma unroll
          for (int a = 0; a < DIM; ++a) gq[(int64_t)a * V] = ggs[o][a];
        }
      }
    }
  }
  if (SELF && absmax_out) {
    __shared__ float smem[NT / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) smem[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, smem[w]);
      if (m > __builtin_nontemporal_load(absmax_out)) atomicMax(reinterpret_cast<unsigned int*>(absmax_out), __float_as_uint(m));
    }
  }
}

// Drains the overflow list: every corner of a listed sample, global atomics (runs after the tiles were stored).
template <int DIM, int PAD>
__global__ void __launch_bounds__(kBlock)
k_gather_overflow(const float* __restrict__ gout, const float* __restrict__ grid, float* __restrict__ gin, int C, Dims d,
                  int clamp_grid, const int* __restrict__ ovf_count, const int2* __restrict__ ovf_list, int ovf_cap,
                  float* __restrict__ absmax_out) {
  const int V = (int)d.voxels();
  const int count = min(*ovf_count, ovf_cap);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    const int2 e = ovf_list[i];
    const int n = e.x, s = e.y;
    const float* pn = grid + (int64_t)n * DIM * V;
    float gx = pn[s], gy = pn[V + s], gz = DIM == 3 ? pn[2 * V + s] : 0.f;
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
          const int o = t.off(cz, cy, cx, d);
          const float wgt = t.w(cz, cy, cx);
          for (int c = 0; c < C; ++c) {
            const float v = wgt * gout[((int64_t)n * C + c) * V + s];
            const float old = atomicAdd(gin + ((int64_t)n * C + c) * V + o, v);
            const float nv = fabsf(old + v);
            if (absmax_out && nv > __builtin_nontemporal_load(absmax_out))
              atomicMax(reinterpret_cast<unsigned int*>(absmax_out), __float_as_uint(nv));
          }
        }
  }
}

// max over samples and axes of |unnormalize(phi_a) - s_a|: the displacement bound (in voxels) that selects the halo
template <int VEC>
__global__ void __launch_bounds__(kBlock) k_max_displacement(const float* __restrict__ phi, float* __restrict__ out, Dims d, int ndim) {
  const int V = (int)d.voxels();
  const int plane = blockIdx.y;              // n * ndim + a
  const int a = plane % ndim;
  const int Sa = a == 0 ? d.s2 : (a == 1 ? d.s1 : d.s0);
  const float* p = phi + (int64_t)plane * V;
  float m = 0.f;
  for (int i = (blockIdx.x * kBlock + threadIdx.x) * VEC; i < V; i += gridDim.x * kBlock * VEC) {
    float v[VEC];
    load_vec<VEC>(p + i, v);
    const int x = i % d.s2, q = i / d.s2;
    const int y = q % d.s1, z = q / d.s1;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int sa = a == 0 ? x + k : (a == 1 ? y : z);
      const float dv = fabsf(unnormalize(v[k], Sa) - (float)sa);
      m = fmaxf(m, dv < 1.0e9f ? dv : 1.0e9f);   // NaN -> ignored by fmaxf; inf -> capped
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float smem[kBlock / 64];
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, smem[w]);
    if (m > __builtin_nontemporal_load(out)) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}

// out[r] = max_j slots[r][j]  (NaN propagates, as torch.max does): one workgroup per row of displacement slots.
// NT threads, VEC floats per request: a row of 4096 slots is ONE 16-byte request per thread of a 1024-thread workgroup (round 6;
// 256 threads walking 16 dependent 4-byte trips took 9 us a launch, six launches per cfg-2 call)
template <int NT, int VEC>
__global__ void __launch_bounds__(NT) k_slot_rows_max(float* __restrict__ slots, float* __restrict__ out, int cols, int reset) {
  float* p = slots + (int64_t)blockIdx.x * cols;
  float m = -3.4e38f;
  bool nan = false;
  if (VEC == 4) {
    for (int j = threadIdx.x * 4; j < cols; j += NT * 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + j);
      nan = nan || !(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      if (reset) *reinterpret_cast<float4*>(p + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int j = threadIdx.x; j < cols; j += NT) {
      const float v = p[j];
      nan = nan || !(v == v);
      m = fmaxf(m, v);
      if (reset) p[j] = 0.f;      // the accumulator is the caller's persistent buffer: ready for the next chain
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    const int other = __shfl_xor((int)nan, o, 64);   // not inside `nan || ...`: a lane that short-circuits leaves the shuffle
    nan = nan || other != 0;
  }
  __shared__ float sm[NT / 64];
  __shared__ int sn[NT / 64];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; sn[threadIdx.x >> 6] = nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NT / 64; ++w) { m = fmaxf(m, sm[w]); nan = nan || sn[w]; }
    out[blockIdx.x] = nan ? __int_as_float(0x7fc00000) : m;
  }
}

// flag[0] |= 1 when some v[i] is outside [lo[i], hi[i]) (a NaN is outside): the premise check of a replayed launch plan
__global__ void __launch_bounds__(64) k_bounds_check(const float* __restrict__ v, const float* __restrict__ lo,
                                                     const float* __restrict__ hi, int n, int* __restrict__ flag) {
  bool bad = false;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float x = v[i];
    bad = bad || !(x >= lo[i] && x < hi[i]);
  }
  if (__any(bad) && threadIdx.x == 0) atomicOr(flag, 1);
}

}  // namespace advchain

using namespace advchain;

template <int DIM, int C, int H, bool SELF, bool GG, int TZ, int TY, int NT>
static void launch_gather(const float* gout, const float* in, const float* grid, float* gin, float* ggrid, int64_t N,
                          Dims d, int padding, int clamp_grid, int32_t* ws, int chain, bool strict, hipStream_t st) {
  using G = GatherCfg<DIM, C, H, SELF, GG, TZ, TY, NT>;
  const int wide = d.s2 > 64;
  const int n2 = wide ? (d.s2 + (64 - 2 * kXPad) - 1) / (64 - 2 * kXPad) : 1;
  const int n1 = (d.s1 + TY - 1) / TY;
  const int n0 = DIM == 3 ? (d.s0 + TZ - 1) / TZ : 1;
  int* cnt = ws;
  float* amax_out = SELF ? reinterpret_cast<float*>(ws + 3) : nullptr;
  int2* list = reinterpret_cast<int2*>(ws + 4);
  const int64_t cap64 = N * d.voxels();
  const int cap = cap64 > 0x7fffffff ? 0x7fffffff : (int)cap64;
  const bool border = padding == PAD_BORDER;
  const uintptr_t al = reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(grid) |
                       (GatherCfg<DIM, C, H, SELF, GG, TZ, TY, NT>::STAGE_IN ? reinterpret_cast<uintptr_t>(in) : 0);
  const bool unaligned = (d.s2 & 3) != 0 || (al & 15) != 0;
  const int flags = ((border || clamp_grid) ? kClip : 0) | (border ? kBorder : 0) | (clamp_grid ? kClampGrid : 0) |
                    (unaligned ? kUnaligned : 0);
  auto kern = k_adjoint_gather<DIM, C, H, SELF, GG, TZ, TY, NT>;
  static bool attr_set = false;
  if (G::LDS > 65536 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    attr_set = true;
  }
  if (strict) {
    // the caller guarantees the displacement bound (measured): no sample can be irregular, so there is no overflow
    // list to reset or drain and the whole step is this one launch
    hipLaunchKernelGGL(kern, dim3((unsigned)(n0 * n1 * n2), (unsigned)N), dim3(NT), G::LDS, st, gout, in, grid, gin, ggrid,
                       d, n1, n2, wide, flags, (float*)nullptr, (int*)nullptr, (int2*)nullptr, 0, SELF ? ws : (int32_t*)nullptr);
    return;
  }
  hipLaunchKernelGGL(k_gather_prepare, dim3(1), dim3(1), 0, st, ws, chain);
  hipLaunchKernelGGL(kern, dim3((unsigned)(n0 * n1 * n2), (unsigned)N), dim3(NT), G::LDS, st, gout, in, grid, gin, ggrid, d,
                     n1, n2, wide, flags, amax_out, cnt, list, cap, (int32_t*)nullptr);
  if (border) hipLaunchKernelGGL((k_gather_overflow<DIM, PAD_BORDER>), dim3(16), dim3(kBlock), 0, st, gout, grid, gin, C, d,
                                 clamp_grid, cnt, list, cap, amax_out);
  else hipLaunchKernelGGL((k_gather_overflow<DIM, PAD_ZEROS>), dim3(16), dim3(kBlock), 0, st, gout, grid, gin, C, d,
                          clamp_grid, cnt, list, cap, amax_out);
}

// rows that are not 16-byte aligned are staged with dword loads (launch_gather: kUnaligned)
static bool gather_shape_ok(const Dims& d) { return d.s2 >= 8; }

// adjoint_march.hip: z-marching form for exact sub-voxel bounds in 3D
int advchain_self_adjoint_march_launch(const float* gout, const float* phi, float* gphi, int64_t N, Dims d,
                                       int32_t* workspace, hipStream_t st);
int advchain_warp_adjoint_march_launch(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                       int64_t N, int64_t C, Dims d, int padding, int clamp_grid, hipStream_t st);

// Self-composition backward in gather form.  `halo` is the caller's displacement bound in voxels; shapes or bounds the
// gather form does not cover return ADVCHAIN_ERR_UNSUPPORTED (the caller uses the LDS-tiled scatter).
// Workspace protocol identical to advchain_scatter_tiled_launch (header [0] overflow count, [3] max|result|).
int advchain_self_adjoint_gather_launch(const float* gout, const float* phi, float* gphi, int64_t N, int ndim, Dims d,
                                        int32_t* workspace, int chain, int halo, hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_ADJOINT_GATHER") != nullptr;   // A/B knob
  const bool strict = halo < 0;     // negative: exact bound |halo|, guaranteed by the caller
  if (strict) halo = -halo;
  if (off || !workspace || halo < 1 || !gather_shape_ok(d)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (strict && ndim == 3 && halo == 1) {
    const int rc = advchain_self_adjoint_march_launch(gout, phi, gphi, N, d, workspace, st);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
#define SELF_GO(DIM_, H_, TZ_, TY_, NT_) \
  launch_gather<DIM_, DIM_, H_, true, false, TZ_, TY_, NT_>(gout, phi, phi, gphi, nullptr, N, d, PAD_BORDER, 0, workspace, chain, strict, st)
  if (ndim == 3) {
    if (halo != 1) return ADVCHAIN_ERR_UNSUPPORTED;
    SELF_GO(3, 1, 4, 4, 512);
  } else {
    if (halo == 1) SELF_GO(2, 1, 1, 16, 256);
    else if (halo == 2) SELF_GO(2, 2, 1, 16, 256);
    else if (halo <= 4) SELF_GO(2, 4, 1, 16, 256);
    else return ADVCHAIN_ERR_UNSUPPORTED;
  }
#undef SELF_GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// grid_sample backward (grad_in [+ grad_grid]) in gather form: same contract; C in {1, 4}, zeros / border padding.
int advchain_warp_adjoint_gather_launch(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                        int64_t N, int64_t C, int ndim, Dims d, int padding, int clamp_grid,
                                        int32_t* workspace, int halo, hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_ADJOINT_GATHER") != nullptr;   // A/B knob
  const bool strict = halo < 0;     // negative: exact bound |halo|, guaranteed by the caller
  if (strict) halo = -halo;
  if (off || !workspace || halo < 1 || !gin || padding == PAD_REFLECTION) return ADVCHAIN_ERR_UNSUPPORTED;
  if (!gather_shape_ok(d)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (C != 1 && C != 4) return ADVCHAIN_ERR_UNSUPPORTED;
  if (strict && ndim == 3 && halo == 1) {
    const int rc = advchain_warp_adjoint_march_launch(gout, in, grid, gin, ggrid, N, C, d, padding, clamp_grid, st);
    if (rc != ADVCHAIN_ERR_UNSUPPORTED) return rc;
  }
#define WARP_GO(DIM_, C_, H_, GG_, TZ_, TY_, NT_) \
  launch_gather<DIM_, C_, H_, false, GG_, TZ_, TY_, NT_>(gout, in, grid, gin, ggrid, N, d, padding, clamp_grid, workspace, 0, strict, st)
  const bool gg = ggrid != nullptr;
  if (ndim == 3) {
    if (halo != 1) return ADVCHAIN_ERR_UNSUPPORTED;
    if (C == 1) { if (gg) WARP_GO(3, 1, 1, true, 4, 4, 512); else WARP_GO(3, 1, 1, false, 4, 4, 512); }
    else { if (gg) WARP_GO(3, 4, 1, true, 2, 4, 512); else WARP_GO(3, 4, 1, false, 4, 4, 512); }
  } else {
    if (halo <= 2) {
      if (C == 1) { if (gg) WARP_GO(2, 1, 2, true, 1, 16, 256); else WARP_GO(2, 1, 2, false, 1, 16, 256); }
      else { if (gg) WARP_GO(2, 4, 2, true, 1, 16, 256); else WARP_GO(2, 4, 2, false, 1, 16, 256); }
    } else if (halo <= 4) {
      if (C == 1) { if (gg) WARP_GO(2, 1, 4, true, 1, 16, 256); else WARP_GO(2, 1, 4, false, 1, 16, 256); }
      else { if (gg) WARP_GO(2, 4, 4, true, 1, 16, 256); else WARP_GO(2, 4, 4, false, 1, 16, 256); }
    } else return ADVCHAIN_ERR_UNSUPPORTED;
  }
#undef WARP_GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

extern "C" int advchain_max_displacement(const float* phi, float* out, int64_t N, int ndim, const int64_t* dims,
                                         void* stream) {
  ADVCHAIN_CHECK_ARG(phi && out && dims, "max_displacement: null pointer");
  ADVCHAIN_CHECK_ARG(ndim == 2 || ndim == 3, "max_displacement: ndim");
  ADVCHAIN_CHECK_ARG(N >= 0 && N * ndim < 65536, "max_displacement: bad N");
  if (N == 0) return ADVCHAIN_OK;
  Dims d;
  d.s0 = ndim == 3 ? (int)dims[0] : 1;
  d.s1 = (int)dims[ndim - 2];
  d.s2 = (int)dims[ndim - 1];
  ADVCHAIN_CHECK_ARG(d.s0 > 0 && d.s1 > 0 && d.s2 > 0 && d.voxels() < (1ll << 31), "max_displacement: bad dims");
  const bool vec = d.s2 % 4 == 0 && (reinterpret_cast<uintptr_t>(phi) & 15) == 0;
  // ~512 workgroups in all: every workgroup ends in one same-address atomic (~10 ns each when they collide)
  int blocks = advchain_blocks(d.voxels(), kBlock * (vec ? 4 : 1));
  const int cap = (int)(512 / (N * ndim)) > 1 ? (int)(512 / (N * ndim)) : 1;
  if (blocks > cap) blocks = cap;
  dim3 g((unsigned)blocks, (unsigned)(N * ndim));
  if (vec) hipLaunchKernelGGL((k_max_displacement<4>), g, dim3(kBlock), 0, (hipStream_t)stream, phi, out, d, ndim);
  else hipLaunchKernelGGL((k_max_displacement<1>), g, dim3(kBlock), 0, (hipStream_t)stream, phi, out, d, ndim);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

extern "C" int advchain_slot_rows_max(float* slots, float* out, int64_t rows, int64_t cols, int reset, void* stream) {
  ADVCHAIN_CHECK_ARG(slots && out, "slot_rows_max: null pointer");
  ADVCHAIN_CHECK_ARG(rows >= 0 && rows < 65536 && cols >= 1 && cols < (1ll << 31), "slot_rows_max: bad shape");
  if (rows == 0) return ADVCHAIN_OK;
  if ((cols & 3) == 0 && (reinterpret_cast<uintptr_t>(slots) & 15) == 0 && cols >= 2048)
    hipLaunchKernelGGL((k_slot_rows_max<1024, 4>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, slots, out, (int)cols, reset);
  else
    hipLaunchKernelGGL((k_slot_rows_max<kBlock, 1>), dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, slots, out, (int)cols, reset);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

extern "C" int advchain_bounds_check(const float* values, const float* lo, const float* hi, int64_t n, int32_t* flag, void* stream) {
  ADVCHAIN_CHECK_ARG(values && lo && hi && flag, "bounds_check: null pointer");
  ADVCHAIN_CHECK_ARG(n >= 0 && n < (1ll << 20), "bounds_check: bad n");
  if (n == 0) return ADVCHAIN_OK;
  hipLaunchKernelGGL(k_bounds_check, dim3(1), dim3(64), 0, (hipStream_t)stream, values, lo, hi, (int)n, flag);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
