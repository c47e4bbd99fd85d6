// z-marching forward sampler for 3D image warps whose field moves a sample by 2 .. 4 voxels (gfx950): one channel,
// rows of at most 64 voxels -- advchain_grid_sample_fwd on the AdvMorph field the ascent steps end with
// (adv_morph.py:546-557).  sample_march.hip keeps three planes of its rows and sends a lane whose corners leave them to
// global gathers: beyond a voxel that is every lane, and the tile kernel (gather_tiled.hip) that takes over stages a
// (2 + 2h) x (8 + 2h) block per 2 x 8 outputs.  This is the forward half of k_scatter_march3d_wide (scatter_march.hip):
//   * a workgroup owns TY = 8 rows (lane <-> x, wave <-> row) and walks a chunk of planes; the planes z-H .. z+H of
//     rows y0-H .. y0+TY+H-1 of `in` live in an LDS ring of 2H+2 slots, every plane staged ONCE per workgroup with
//     16-byte loads, zero outside the volume and with zero columns either side (zeros padding is data);
//   * the grid values of the own rows come through a 16-byte LDS stage as well; requests run two steps ahead in two
//     register sets that swap roles; one barrier per step; results leave 16 bytes per lane through an LDS stage;
//   * 60 KiB of LDS at H = 4: two workgroups a CU fill each other's barrier bubbles.
// H comes from the caller's displacement hint and is a performance parameter only: a lane whose corners leave the ring
// takes sample_linear's gathers, and the ring path uses the same arithmetic (weights as products, tap_acc order), so the
// result does not depend on H or on which kernel ran.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

template <int H>
struct RingFwdCfg {
  static constexpr int TY = 8, NWV = 8, NT = NWV * 64, XS = 68;
  static constexpr int NP = 2 * H + 2, R = TY + 2 * H;
  static constexpr int RING = NP * R * XS + 4, STAGE = 3 * TY * 64, OBUF = TY * 64;
  static constexpr int KR = (R * 16 + NT - 1) / NT;           // ring items (16 bytes) per thread and plane
  static constexpr size_t LDS = (size_t)(RING + 2 * STAGE + 2 * OBUF) * sizeof(float);
  static_assert(3 * TY * 16 <= NT, "one grid item per thread");
};

template <int PAD, int H>
__global__ void __launch_bounds__(RingFwdCfg<H>::NT)
k_sample_ring(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out, Dims d, int n1, int zc,
              int clamp_grid) {
  using G = RingFwdCfg<H>;
  constexpr int TY = G::TY, NT = G::NT, XS = G::XS, NP = G::NP, R = G::R, KR = G::KR;
  extern __shared__ float lds[];
  float* const ring = lds;                                    // [NP][R][XS]: data at columns 4 .. 67
  float* const stage = ring + G::RING;                        // [2][3][TY][64]: grid x, y, z of the own rows
  float* const obuf = stage + 2 * G::STAGE;                   // [2][TY][64]
  const int V = (int)d.voxels();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // blocks are dealt to the 8 XCDs round-robin: a contiguous run of tiles (y fastest, then z, then the batch) per XCD
  const int nb = gridDim.x, ntile = n1 * ((d.s0 + zc - 1) / zc);
  const int tl = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
  const int n = tl / ntile, trem = tl - n * ntile;
  const int ty = trem % n1, tz = trem / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* inn = in + (int64_t)n * V;
  const float* gn = grid + (int64_t)n * 3 * V;
  float* outn = out + (int64_t)n * V;
  const bool xin = lane < d.s2;

  auto ring_slot = [&](int z) { return ((z % NP) + NP) % NP; };
  // requests are unconditional, from addresses clamped into the volume; what lies outside becomes zero on the way to LDS
  const int qx = min(4 * (tid & 15), max(d.s2 - 4, 0));
  auto fetch_in = [&](int z, float4 (&v)[KR]) {
    const int zq = min(max(z, 0), d.s0 - 1);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int r = min((tid + k * NT) >> 4, R - 1);
      const int y = min(max(y0 - H + r, 0), d.s1 - 1);
      v[k] = *reinterpret_cast<const float4*>(inn + (unsigned)((zq * d.s1 + y) * d.s2 + qx));
    }
  };
  auto commit_in = [&](int z, const float4 (&v)[KR]) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int i = tid + k * NT;
      if (i >= R * 16) continue;
      const int r = i >> 4, q = i & 15, y = y0 - H + r;
      const bool ok = z >= 0 && z < d.s0 && y >= 0 && y < d.s1 && 4 * q < d.s2;
      *reinterpret_cast<float4*>(ring + (ring_slot(z) * R + r) * XS + 4 + 4 * q) = ok ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto fetch_grid = [&](int z, float4& v) {                    // threads 0 .. 3 TY 16 - 1: (channel, own row, 4 x)
    const int i = min(tid, 3 * TY * 16 - 1);
    const int ch = i / (TY * 16), r = (i >> 4) % TY;
    const int y = min(y0 + r, d.s1 - 1), zq = min(max(z, 0), d.s0 - 1);
    v = *reinterpret_cast<const float4*>(gn + (int64_t)ch * V + (unsigned)((zq * d.s1 + y) * d.s2 + qx));
  };
  auto commit_grid = [&](float* st, const float4& v) {
    if (tid < 3 * TY * 16) *reinterpret_cast<float4*>(st + tid * 4) = v;
  };

  // ---- prologue: ring planes za-H .. za+H, the first plane's grid rows, the requests of step za+1; zero columns
  float4 inA[KR], inB[KR], gA, gB;
  {
    float4 pin[2 * H + 1][KR];
#pragma unroll
    for (int p = 0; p <= 2 * H; ++p) fetch_in(za - H + p, pin[p]);
    fetch_grid(za, gB);
    fetch_in(za + H + 1, inA);
    fetch_grid(za + 1, gA);
    for (int i = tid; i < G::RING; i += NT) ring[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int p = 0; p <= 2 * H; ++p) commit_in(za - H + p, pin[p]);
    commit_grid(stage, gB);
  }
  __syncthreads();

  int cur = 0;
  // ic / gc: requested a step ago for step z+1, committed at the end of this one; il / gl: requested now for step z+2
  auto step = [&](int z, float4 (&ic)[KR], float4& gc, float4 (&il)[KR], float4& gl) {
    const float* st = stage + cur * G::STAGE;
    float* ob = obuf + cur * G::OBUF;
    fetch_in(z + 2 + H, il);
    fetch_grid(z + 2, gl);
    const int uy = y0 + wave;
    if (uy < d.s1) {                                           // wave-uniform
      float g[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) g[a] = st[(a * TY + wave) * 64 + lane];
      if (clamp_grid) { g[0] = clamp_unit(g[0]); g[1] = clamp_unit(g[1]); g[2] = clamp_unit(g[2]); }
      Taps<3, PAD> t;
      t.build(g[0], g[1], g[2], d);
      // all eight corners inside the ring (x: columns -1 .. S2 are there as zeros); NaN coordinates sit at -16: outside
      const int rx = t.x.i0 + 4, ry = t.y.i0 - (y0 - H), rz = t.z.i0 - (z - H);
      const bool staged = rx >= 3 && rx <= 67 && ry >= 0 && ry < R - 1 && rz >= 0 && rz < 2 * H;
      float res = 0.f;
      if (staged) {
        const float* p0 = ring + (ring_slot(t.z.i0) * R + ry) * XS + rx;
        const float* p1 = ring + (ring_slot(t.z.i0 + 1) * R + ry) * XS + rx;
        float acc = 0.f;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cx = 0; cx < 2; ++cx)     // product then sum, never contracted: see tap_acc() (sampler_common.h)
              acc = tap_acc<3>(acc, t.ok(cz, cy, cx) ? ((cz ? p1 : p0) + cy * XS)[cx] : 0.f, t.w(cz, cy, cx));
        res = acc;
      } else if (xin) {
        res = sample_linear<3, PAD>(inn, t, d);
      }
      ob[wave * 64 + lane] = res;
    }
    if (z + 1 < zb) {
      commit_in(z + 1 + H, ic);
      commit_grid(stage + (cur ^ 1) * G::STAGE, gc);
    }
    __syncthreads();
    if (tid < TY * 16) {                                       // the plane leaves 16 bytes per lane
      const int row = tid >> 4, q = tid & 15, oy = y0 + row;
      if (oy < d.s1 && 4 * q < d.s2)
        *reinterpret_cast<float4*>(outn + (unsigned)((z * d.s1 + oy) * d.s2 + 4 * q)) = *reinterpret_cast<const float4*>(ob + row * 64 + 4 * q);
    }
    cur ^= 1;
  };
  for (int z = za; z < zb; z += 2) {
    step(z, inA, gA, inB, gB);
    if (z + 1 < zb) step(z + 1, inB, gB, inA, gA);
  }
}

}  // namespace advchain

using namespace advchain;

// 3D image warp forward, C == 1, rows of at most 64 voxels, displacement hint of 2 .. 4 voxels.
// ADVCHAIN_ERR_UNSUPPORTED: the caller uses the tile kernel / the direct gathers.
int advchain_sample_ring_launch(const float* in, const float* grid, float* out, int64_t N, Dims d, int padding, int clamp_grid,
                                int hint, hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_RING_FWD") != nullptr;   // A/B knob
  static const int h_forced = 0;   // measured optimum (was a tuning knob until round 4)
  if (off || padding == PAD_REFLECTION) return ADVCHAIN_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(grid);
  if (d.s2 < 8 || d.s2 > 64 || (d.s2 & 3) != 0 || (al & 15) != 0 || d.s0 < 2 || d.voxels() * 4 >= (1ll << 31))
    return ADVCHAIN_ERR_UNSUPPORTED;
  int H = h_forced > 0 ? h_forced : hint;
  if (H < 2 || H > 4) return ADVCHAIN_ERR_UNSUPPORTED;
  const int n1 = (int)((d.s1 + 7) / 8);
  static const int zc_forced = 0;
  int zc = (int)d.s0;
  while (zc > 16 && N * n1 * ((d.s0 + zc - 1) / zc) < 1024) zc = (zc + 1) / 2;
  if (zc_forced > 0) zc = zc_forced;
  const int n0 = (int)((d.s0 + zc - 1) / zc);
  dim3 g((unsigned)(n1 * n0 * N));
#define GO(PAD_, H_) do { \
    auto kern = k_sample_ring<PAD_, H_>; \
    static bool attr_set = false; \
    if (RingFwdCfg<H_>::LDS > 65536 && !attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RingFwdCfg<H_>::LDS); attr_set = true; } \
    hipLaunchKernelGGL(kern, g, dim3(RingFwdCfg<H_>::NT), RingFwdCfg<H_>::LDS, st, in, grid, out, d, n1, zc, clamp_grid); } while (0)
#define GO_H(PAD_) do { if (H == 2) GO(PAD_, 2); else if (H == 3) GO(PAD_, 3); else GO(PAD_, 4); } while (0)
  if (padding == PAD_BORDER) GO_H(PAD_BORDER); else GO_H(PAD_ZEROS);
#undef GO_H
#undef GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
