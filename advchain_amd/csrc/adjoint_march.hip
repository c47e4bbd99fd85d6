// z-marching gather-form adjoint for 3D warps with an EXACT sub-voxel displacement bound (gfx950).
//
// Same mathematics as adjoint_gather.hip (H = 1):  grad_in[u] = sum_{s in u +- 1} grad_out[s] * prod_a tent(p_s,a - u_a)
// plus the coordinate path at s = u -- but organised around the SAMPLES instead of the outputs, and marched along z:
//
//   * a workgroup owns TY output rows (whole x rows, lane <-> x) and walks ZC planes.  Each sample plane is staged
//     ONCE per workgroup (16-byte loads, offsets o = unnormalize(grid) - s computed at staging time) and each staged
//     sample row is read ONCE per wave that needs it: its clip, its three x tents and its three z tents are computed
//     once and deposited on the 3 (z) x 3 (x) x RPW (y) outputs it touches.  The output-centric kernel re-reads and
//     re-derives every sample row for each of the 9 output rows around it (x2.4 the VALU work, x3 the LDS reads).
//   * a wave keeps 3 planes of partial sums per owned row in registers; the plane that has seen all three of its
//     sample planes is folded over x with two whole-wave DPP shifts and stored, the others shift down one slot.
//   * the next plane is loaded into registers at the top of a step and written to LDS at its end (one barrier per
//     step): the loads of step k+1 are in flight under the arithmetic of step k.  The image / field planes the
//     coordinate path needs (z-1, z, z+1 around a sample) live in a 4-slot ring, staged two steps ahead.
//   * with |p - s| < 1 guaranteed the tents need no general max(0, 1 - |f - k|): t(-1) = max(0, -f), t(+1) = max(0, f),
//     t(0) = 1 - t(-1) - t(+1).
//
// Contract: the caller guarantees |unnormalize(grid) - s| < 1 voxel for every sample (ops.squaring_halo / ops.warp_halo
// measure it in the forward).  Shapes outside the fast form (rows longer than 64 voxels, rows not a multiple of 4,
// misaligned bases) return ADVCHAIN_ERR_UNSUPPORTED and the caller keeps the tile kernel of adjoint_gather.hip.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

constexpr int kMarchClip = 1;     // sampling positions are clipped to [0, S-1] (border padding and/or clamp_grid)
constexpr int kMarchBorder = 2;   // border padding: zero coordinate gradient AT and beyond the border (else: beyond only)

template <int C, bool SELF, bool GG, int NW, int RPW>
struct MarchCfg {
  static constexpr int TY = NW * RPW;
  static constexpr int R = TY + 2;                              // staged rows per plane (one halo row each side)
  static constexpr int NT = NW * 64;
  static constexpr bool HAS_IMG = GG && !SELF;
  static constexpr int RING_CH = SELF ? 3 : (HAS_IMG ? C : 0);  // planes z-1..z+1 are needed: 4-slot ring
  static constexpr int LATE_CH = SELF ? 3 : 3 + C;              // only the current plane is needed: 2 slots
  static constexpr int RING_FLOATS = 4 * RING_CH * R * 64;
  static constexpr size_t LDS = (size_t)(RING_FLOATS + 2 * LATE_CH * R * 64) * sizeof(float);
  static_assert(R * 16 <= NT, "one staging item (4 voxels of one row, all channels) per thread");
  static_assert(!SELF || C == 3, "the self-composition carries 3 channels");
};

__device__ __forceinline__ float march_unnormalize(float g, int S) { return ((g + 1.f) * 0.5f) * (float)(S - 1); }

template <int C, bool SELF, bool GG, int NW, int RPW>
__global__ void __launch_bounds__(NW * 64)
k_adjoint_march(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ grid,
                float* __restrict__ gin, float* __restrict__ ggrid, Dims d, int n1, int zc, int flags,
                int32_t* __restrict__ untracked) {
  using G = MarchCfg<C, SELF, GG, NW, RPW>;
  constexpr int R = G::R, TY = G::TY, RC = G::RING_CH, LC = G::LATE_CH;
  if (untracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) untracked[3] = -1;   // see adjoint_gather.hip
  extern __shared__ float lds[];
  float* const ring = lds;                       // [slot 4][RC][R][64]
  float* const late = lds + G::RING_FLOATS;      // [slot 2][LC][R][64]
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ty = blockIdx.x % n1, tz = blockIdx.x / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* gn = grid + (int64_t)n * 3 * V;
  const float* gon = gout + (int64_t)n * C * V;
  const float* inn = in + (int64_t)n * C * V;
  float* ginn = gin + (int64_t)n * C * V;
  const int S[3] = {d.s2, d.s1, d.s0};
  const bool clip = SELF ? true : (flags & kMarchClip) != 0, border = SELF ? true : (flags & kMarchBorder) != 0;

  // ---- staging item of this thread: 4 consecutive x of staged row r_st, every channel
  const bool has_item = threadIdx.x < R * 16;
  const int r_st = threadIdx.x >> 4, q_st = threadIdx.x & 15;
  const int sy_st = y0 - 1 + r_st, x_st = 4 * q_st;
  const bool row_ok = has_item && sy_st >= 0 && sy_st < d.s1 && x_st < d.s2;
  const int row_off = sy_st * d.s2 + x_st;
  const int lds_item = r_st * 64 + x_st;

  // field plane p -> offsets o = unnormalize(field) - own voxel (what the tents and the corner search work on)
  auto load_field = [&](int p, float (&v)[3][4]) {
    if (row_ok && p >= 0 && p < d.s0) {
      const int s = p * d.s1 * d.s2 + row_off;
#pragma unroll
      for (int a = 0; a < 3; ++a) load_vec<4>(gn + (int64_t)a * V + s, v[a]);
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[a][k] = __int_as_float(0x7fc00000);   // marker: outside the volume
    }
  };
  auto field_to_offsets = [&](int p, float (&v)[3][4]) {
    const int sc[3] = {x_st, sy_st, p};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xs = march_unnormalize(v[a][k], S[a]);
        const float sa = (float)(sc[a] + (a == 0 ? k : 0));
        v[a][k] = (xs > -1.0e9f && xs < 1.0e9f) ? xs - sa : 0.f;            // NaN (outside the volume) -> 0
      }
  };
  auto load_plain = [&](const float* base, int nch, int p, float (*v)[4]) {
    const bool ok = row_ok && p >= 0 && p < d.s0;
    const int s = p * d.s1 * d.s2 + row_off;
    for (int c = 0; c < nch; ++c) {
      if (ok) {
        const float4 t = *reinterpret_cast<const float4*>(base + (int64_t)c * V + s);
        v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
      } else {
        v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
      }
    }
  };
  auto store_lds = [&](float* base, int ch, int nch_total, int slot, const float (&v)[4]) {
    *reinterpret_cast<float4*>(base + ((slot * nch_total + ch) * R) * 64 + lds_item) = make_float4(v[0], v[1], v[2], v[3]);
  };
  // ring plane p: SELF -> field offsets; warp with grad_grid -> the image.  late plane p: SELF -> grad_out; warp ->
  // field offsets + grad_out
  float pr[RC > 0 ? RC : 1][4], pl[LC][4];
  auto fetch_ring = [&](int p) {
    if constexpr (SELF) load_field(p, pr);
    else if constexpr (G::HAS_IMG) load_plain(inn, C, p, pr);
  };
  auto commit_ring = [&](int p) {
    if constexpr (RC > 0) {
      if constexpr (SELF) field_to_offsets(p, pr);
      if (has_item) {
#pragma unroll
        for (int c = 0; c < RC; ++c) store_lds(ring, c, RC, p & 3, pr[c]);
      }
    }
  };
  auto fetch_late = [&](int p) {
    if constexpr (SELF) load_plain(gon, 3, p, pl);
    else {
      float f[3][4];
      load_field(p, f);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) pl[a][k] = f[a][k];
      load_plain(gon, C, p, pl + 3);
    }
  };
  auto commit_late = [&](int p) {
    if constexpr (!SELF) {
      float f[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) f[a][k] = pl[a][k];
      field_to_offsets(p, f);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) pl[a][k] = f[a][k];
    }
    if (has_item) {
#pragma unroll
      for (int c = 0; c < LC; ++c) store_lds(late, c, LC, p & 1, pl[c]);
    }
  };

  // ---- prologue: ring planes za-1, za; late plane za-1
  fetch_ring(za - 1);
  commit_ring(za - 1);
  fetch_ring(za);
  commit_ring(za);
  fetch_late(za - 1);
  commit_late(za - 1);
  __syncthreads();

  // partial sums: [owned row][target plane zp-1, zp, zp+1][channel][deposit on x-1, x, x+1]
  float acc[RPW][3][C][3];
#pragma unroll
  for (int o = 0; o < RPW; ++o)
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[o][kz][c][kx] = 0.f;
  float gg_hold[RPW][3];     // SELF: coordinate-path gradient of the previous plane's samples (added when that plane is stored)
#pragma unroll
  for (int o = 0; o < RPW; ++o) gg_hold[o][0] = gg_hold[o][1] = gg_hold[o][2] = 0.f;

  const float xlo = clip ? -(float)lane : -3.0e38f, xhi = clip ? (float)(d.s2 - 1 - lane) : 3.0e38f;
  const bool xowned = lane < d.s2;

  for (int zp = za - 1; zp <= zb; ++zp) {
    // ---- loads of the next planes go out first: they land while this plane is being worked on
    const bool more_ring = zp + 2 <= zb, more_late = zp + 1 <= zb;
    if (more_ring) fetch_ring(zp + 2);
    if (more_late) fetch_late(zp + 1);

    const int rs = zp & 3, ls = zp & 1;
    const float* fbase = SELF ? ring + (rs * RC) * R * 64 : late + (ls * LC) * R * 64;              // field offsets, 3 ch
    const float* gobase = SELF ? late + (ls * LC) * R * 64 : late + (ls * LC + 3) * R * 64;         // grad_out, C ch

    // ---- phase B: deposits of sample plane zp on the target planes zp-1, zp, zp+1
    if (zp >= 0 && zp < d.s0) {
      const float zlo = clip ? -(float)zp : -3.0e38f, zhi = clip ? (float)(d.s0 - 1 - zp) : 3.0e38f;
#pragma unroll
      for (int i = 0; i < RPW + 2; ++i) {
        const int r = wave * RPW + i;           // staged row of the sample; sample y = y0 - 1 + r
        const int ys = y0 - 1 + r;
        float fx = fbase[(0 * R + r) * 64 + lane];
        float fy = fbase[(1 * R + r) * 64 + lane];
        float fz = fbase[(2 * R + r) * 64 + lane];
        float go[C];
#pragma unroll
        for (int c = 0; c < C; ++c) go[c] = gobase[(c * R + r) * 64 + lane];
        const float ylo = clip ? -(float)ys : -3.0e38f, yhi = clip ? (float)(d.s1 - 1 - ys) : 3.0e38f;
        fx = __builtin_amdgcn_fmed3f(fx, xlo, xhi);
        fy = __builtin_amdgcn_fmed3f(fy, ylo, yhi);
        fz = __builtin_amdgcn_fmed3f(fz, zlo, zhi);
        float tx[3], tyv[3], tzv[3];
        tx[0] = fmaxf(0.f, -fx); tx[2] = fmaxf(0.f, fx); tx[1] = (1.f - tx[0]) - tx[2];
        tyv[0] = fmaxf(0.f, -fy); tyv[2] = fmaxf(0.f, fy); tyv[1] = (1.f - tyv[0]) - tyv[2];
        tzv[0] = fmaxf(0.f, -fz); tzv[2] = fmaxf(0.f, fz); tzv[1] = (1.f - tzv[0]) - tzv[2];
#pragma unroll
        for (int o = 0; o < RPW; ++o) {
          const int k = o + 1 - i;              // output row minus sample row (compile time)
          if (k < -1 || k > 1) continue;
          const float wy = tyv[k + 1];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float a = go[c] * wy;
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) {
              const float b = a * tzv[kz];
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) acc[o][kz][c][kx] = fmaf(b, tx[kx], acc[o][kz][c][kx]);
            }
          }
        }
      }
    }

    // ---- target plane zp-1 has now seen its three sample planes: fold over x and store
    const int zt = zp - 1;
    if (zt >= za && zt < zb) {
#pragma unroll
      for (int o = 0; o < RPW; ++o) {
        const int uy = y0 + wave * RPW + o;
        if (uy >= d.s1) continue;               // wave-uniform
        const int s = (zt * d.s1 + uy) * d.s2 + lane;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v = lane_prev_f(acc[o][0][c][2]) + acc[o][0][c][1] + lane_next_f(acc[o][0][c][0]);
          if (SELF) v += gg_hold[o][c < 3 ? c : 0];
          if (xowned) ginn[(int64_t)c * V + s] = v;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < RPW; ++o)
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          acc[o][0][c][kx] = acc[o][1][c][kx];
          acc[o][1][c][kx] = acc[o][2][c][kx];
          acc[o][2][c][kx] = 0.f;
        }

    // ---- phase A: coordinate-path gradient of the owned samples of plane zp (corner values from the ring)
    if ((SELF || GG) && zp >= za && zp < zb) {
#pragma unroll
      for (int o = 0; o < RPW; ++o) {
        const int r = wave * RPW + o + 1;
        const int uy = y0 - 1 + r;
        if (uy >= d.s1) continue;               // wave-uniform
        const int sc[3] = {lane, uy, zp};
        float w1[3], mult[3];
        int i0[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float xs = fbase[(a * R + r) * 64 + lane] + (float)sc[a];
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
        }
        float go[C];
#pragma unroll
        for (int c = 0; c < C; ++c) go[c] = gobase[(c * R + r) * 64 + lane];
        // a corner outside the volume reads as 0: rows and planes outside it are staged as zeros; only x needs a select
        const bool okx0 = i0[0] >= 0, okx1 = i0[0] + 1 < d.s2;
        const int lx0 = max(i0[0], 0), lx1 = min(i0[0] + 1, d.s2 - 1);
        const int ry = i0[1] - (y0 - 1);        // staged row of the lower y corner (0 .. R-2)
        const int p0 = (i0[2] & 3) * RC * R * 64, p1 = ((i0[2] + 1) & 3) * RC * R * 64;
        const float wx1 = w1[0], wx0 = 1.f - wx1, wy1 = w1[1], wy0 = 1.f - wy1, wz1 = w1[2], wz0 = 1.f - wz1;
        float acc3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v[2][2][2];
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
              const float* p = ring + (cz ? p1 : p0) + (c * R + ry + cy) * 64;
              const float a0 = p[lx0], a1 = p[lx1];
              v[cz][cy][0] = okx0 ? a0 : 0.f;
              v[cz][cy][1] = okx1 ? a1 : 0.f;
            }
          // SELF: phi_c(u) = (o_c(u) + u_c) * 2/(S_c-1) - 1: only differences along an axis enter; the identity part of a
          // difference along axis c is the index step (1)
          const float ux = (SELF && c == 0) ? 1.f : 0.f, uyy = (SELF && c == 1) ? 1.f : 0.f, uzz = (SELF && c == 2) ? 1.f : 0.f;
          const float dx = ((v[0][0][1] - v[0][0][0] + ux) * wy0 + (v[0][1][1] - v[0][1][0] + ux) * wy1) * wz0 +
                           ((v[1][0][1] - v[1][0][0] + ux) * wy0 + (v[1][1][1] - v[1][1][0] + ux) * wy1) * wz1;
          const float dy = ((v[0][1][0] - v[0][0][0] + uyy) * wx0 + (v[0][1][1] - v[0][0][1] + uyy) * wx1) * wz0 +
                           ((v[1][1][0] - v[1][0][0] + uyy) * wx0 + (v[1][1][1] - v[1][0][1] + uyy) * wx1) * wz1;
          const float dz = ((v[1][0][0] - v[0][0][0] + uzz) * wx0 + (v[1][0][1] - v[0][0][1] + uzz) * wx1) * wy0 +
                           ((v[1][1][0] - v[0][1][0] + uzz) * wx0 + (v[1][1][1] - v[0][1][1] + uzz) * wx1) * wy1;
          const float kc = SELF ? go[c] * (2.f / (float)(S[c < 3 ? c : 0] - 1)) : go[c];
          acc3[0] = fmaf(dx, kc, acc3[0]); acc3[1] = fmaf(dy, kc, acc3[1]); acc3[2] = fmaf(dz, kc, acc3[2]);
        }
        if constexpr (SELF) {
#pragma unroll
          for (int a = 0; a < 3; ++a) gg_hold[o][a] = mult[a] * acc3[a];
        } else {
          if (xowned) {
            float* gq = ggrid + (int64_t)n * 3 * V + (zp * d.s1 + uy) * d.s2 + lane;
#pragma unroll
            for (int a = 0; a < 3; ++a) gq[(int64_t)a * V] = mult[a] * acc3[a];
          }
        }
      }
    }

    // ---- the prefetched planes go to LDS; nobody reads these slots in this step (ring: zp+2 = zp-2 mod 4, last read
    // in step zp-1; late: zp+1 = zp-1 mod 2, last read in step zp-1)
    if (more_ring) commit_ring(zp + 2);
    if (more_late) commit_late(zp + 1);
    __syncthreads();
  }
}

}  // namespace advchain

using namespace advchain;

static int march_zc(const Dims& d, int64_t N, int ty) {
  static const int forced = getenv("ADVCHAIN_MARCH_ZC") ? atoi(getenv("ADVCHAIN_MARCH_ZC")) : 0;   // tuning knob
  if (forced > 0) return forced;
  // enough workgroups to fill 256 CUs four deep, but chunks no shorter than 8 planes (2 of ZC + 2 steps are halo work)
  const int64_t cols = N * ((d.s1 + ty - 1) / ty);
  int zc = d.s0;
  while (zc > 8 && cols * ((d.s0 + zc - 1) / zc) < 1024) zc = (zc + 1) / 2;
  return zc;
}

template <int C, bool SELF, bool GG, int NW, int RPW>
static void launch_march(const float* gout, const float* in, const float* grid, float* gin, float* ggrid, int64_t N,
                         Dims d, int flags, int32_t* ws, hipStream_t st) {
  using G = MarchCfg<C, SELF, GG, NW, RPW>;
  auto kern = k_adjoint_march<C, SELF, GG, NW, RPW>;
  static bool attr_set = false;
  if (G::LDS > 65536 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
    attr_set = true;
  }
  const int n1 = (d.s1 + G::TY - 1) / G::TY;
  const int zc = march_zc(d, N, G::TY);
  const int n0 = (d.s0 + zc - 1) / zc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n1 * n0), (unsigned)N), dim3(G::NT), G::LDS, st, gout, in, grid, gin, ggrid, d, n1,
                     zc, flags, SELF ? ws : (int32_t*)nullptr);
}

static bool march_shape_ok(const Dims& d, const void* a, const void* b, const void* c) {
  static const bool off = getenv("ADVCHAIN_NO_MARCH_ADJOINT") != nullptr;   // A/B knob
  if (off) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
  return d.s2 >= 8 && d.s2 <= 64 && (d.s2 & 3) == 0 && (al & 15) == 0 && d.s0 >= 2;
}

// Exact-bound (|displacement| < 1 voxel) self-composition backward, 3D.  ADVCHAIN_ERR_UNSUPPORTED: use the tile kernel.
int advchain_self_adjoint_march_launch(const float* gout, const float* phi, float* gphi, int64_t N, Dims d,
                                       int32_t* workspace, hipStream_t st) {
  if (!march_shape_ok(d, gout, phi, nullptr)) return ADVCHAIN_ERR_UNSUPPORTED;
  static const int rpw = getenv("ADVCHAIN_MARCH_SELF_RPW") ? atoi(getenv("ADVCHAIN_MARCH_SELF_RPW")) : 1;   // tuning knob
  if (rpw == 2) launch_march<3, true, false, 4, 2>(gout, phi, phi, gphi, nullptr, N, d, 0, workspace, st);
  else launch_march<3, true, false, 4, 1>(gout, phi, phi, gphi, nullptr, N, d, 0, workspace, st);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// Exact-bound grid_sample backward (grad_in [+ grad_grid]), 3D, C in {1, 4}, zeros / border padding.
int advchain_warp_adjoint_march_launch(const float* gout, const float* in, const float* grid, float* gin, float* ggrid,
                                       int64_t N, int64_t C, Dims d, int padding, int clamp_grid, hipStream_t st) {
  if (padding == PAD_REFLECTION || (C != 1 && C != 4) || !gin) return ADVCHAIN_ERR_UNSUPPORTED;
  if (!march_shape_ok(d, gout, grid, ggrid ? in : nullptr)) return ADVCHAIN_ERR_UNSUPPORTED;
  const bool border = padding == PAD_BORDER;
  const int flags = ((border || clamp_grid) ? kMarchClip : 0) | (border ? kMarchBorder : 0);
  static const int rpw1 = getenv("ADVCHAIN_MARCH_C1_RPW") ? atoi(getenv("ADVCHAIN_MARCH_C1_RPW")) : 2;   // tuning knob
  if (C == 1) {
    if (ggrid) {
      if (rpw1 == 2) launch_march<1, false, true, 4, 2>(gout, in, grid, gin, ggrid, N, d, flags, nullptr, st);
      else launch_march<1, false, true, 4, 1>(gout, in, grid, gin, ggrid, N, d, flags, nullptr, st);
    } else {
      launch_march<1, false, false, 4, 2>(gout, in, grid, gin, ggrid, N, d, flags, nullptr, st);
    }
  } else {
    if (ggrid) launch_march<4, false, true, 4, 1>(gout, in, grid, gin, ggrid, N, d, flags, nullptr, st);
    else launch_march<4, false, false, 4, 1>(gout, in, grid, gin, ggrid, N, d, flags, nullptr, st);
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
