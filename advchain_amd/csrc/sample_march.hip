// z-marching forward sampler for 3D warps at equal input / output size (gfx950): advchain_grid_sample_fwd and
// advchain_compose_self_fwd for fields that move a sample by less than a voxel -- the image warps of a freshly
// initialised AdvMorph and the early squarings of its scaling-and-squaring chain (adv_morph.py:132-135,165-168).
//
// A workgroup owns TY output rows (whole x rows, lane <-> x) and walks ZC planes.  The input planes z-1, z, z+1 of its
// rows (plus one halo row either side) live in a 4-slot LDS ring: every plane is staged ONCE per workgroup with 16-byte
// loads -- the tile kernel of gather_tiled.hip stages a (2+2) x (8+2) block for 2 x 8 outputs, 2.5x its tile -- and
// the plane two steps ahead is loaded into registers at the top of a step and written to LDS at its end, so that its
// memory round trip runs under the taps of the current plane (one barrier per step).  Staged rows carry 4 zero floats
// either side and rows / planes outside the volume are staged as zeros: a corner outside the volume reads the 0 zeros
// padding asks for without a select.  A wave whose lanes all find their 8 corners in the ring takes them from LDS;
// otherwise (displacement of a voxel or more somewhere in the row) the row falls back to global gathers, so results do
// not depend on what is staged.  Results leave through LDS, 16 bytes per lane (see adjoint_march.hip).
//
// The arithmetic of a tap (weights as products, accumulation order) is the one of gather_tiled.hip / sample_linear:
// the squaring chain amplifies rounding differences 2^8-fold and its parity tolerance was set with that order.
#include <stdlib.h>
#include "sampler_common.h"

namespace advchain {

constexpr int kFwdSegOwn = 56;   // owned lanes of an x segment (rows longer than 64 voxels)
enum { kFwdFree = 0, kFwdClamp = 1, kFwdBorder = 2 };   // zeros padding | zeros padding + clamp(grid, -1, 1) | border padding

template <int C, bool SELF, int NW, int RPW>
struct FwdMarchCfg {
  static constexpr int TY = NW * RPW;
  static constexpr int R = TY + 2;
  static constexpr int NT = NW * 64;
  static constexpr int PITCH = 72;                          // 4 zeros | 64 voxels | 4 zeros
  static constexpr int PS = C * R * PITCH;                  // floats per ring slot
  static constexpr int NA_ROUND = C > 2 ? 2 : C;            // output channels transposed per round
  static constexpr int TRW = NA_ROUND * RPW * 64;
  static constexpr size_t LDS = (size_t)(4 * PS + NW * TRW) * sizeof(float);
  static_assert(R * 16 <= NT, "one staging item (4 voxels of one row, all channels) per thread");
  static_assert(!SELF || C == 3, "the self-composition carries 3 channels");
};

template <int C, bool SELF, int MODE, int NW, int RPW>
__global__ void __launch_bounds__(NW * 64)
k_sample_march(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out,
               const float* __restrict__ phi0, Dims d, int n1, int zc, int final_mode, float* __restrict__ disp_out, int nseg) {
  using G = FwdMarchCfg<C, SELF, NW, RPW>;
  constexpr int R = G::R, TY = G::TY, P = G::PITCH, PS = G::PS;
  constexpr int PAD = MODE == kFwdBorder ? PAD_BORDER : PAD_ZEROS;
  extern __shared__ float lds[];
  float* const ring = lds;                  // [slot 4][C][R][P]
  float* const trbuf = lds + 4 * PS;        // [wave][TRW]
  const int V = (int)d.voxels();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup L runs on XCD L % 8 (observed dispatch order; speed only): give every XCD a contiguous run of tiles, so
  // that the halo rows / planes two neighbouring workgroups both stage are served by the same L2 (nseg < 0: off)
  int tile = blockIdx.x + gridDim.x * blockIdx.y;
  const int tiles = gridDim.x * gridDim.y;
  if (nseg > 0 && (tiles & 7) == 0) tile = (tile & 7) * (tiles >> 3) + (tile >> 3);
  nseg = nseg < 0 ? -nseg : nseg;
  const int n = tile / (int)gridDim.x;
  // rows longer than 64 voxels: x segments of 56 owned lanes with 4 halo lanes either side (16-byte aligned staging)
  int rem = tile - n * (int)gridDim.x;
  const int seg = rem % nseg;
  rem /= nseg;
  const int xbase = nseg > 1 ? seg * kFwdSegOwn - 4 : 0;      // x of lane 0
  const int xl = xbase + lane;                                // x of this lane
  const bool xowned = nseg > 1 ? (lane >= 4 && lane < 60 && xl < d.s2) : lane < d.s2;
  const int ty = rem % n1, tz = rem / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* inn = in + (int64_t)n * C * V;
  const float* gn = SELF ? nullptr : grid + (int64_t)n * 3 * V;
  const float* p0n = (SELF && final_mode == 1) ? phi0 + (int64_t)n * 3 * V : nullptr;
  float* outn = out + (int64_t)n * C * V;
  const int plane_stride = d.s1 * d.s2;

  // ---- the zero columns of every staged row (never written again)
  for (int e = threadIdx.x; e < 4 * C * R * 2; e += G::NT) {
    const int row = e >> 1, side = e & 1;
    *reinterpret_cast<float4*>(lds + row * P + (side ? 68 : 0)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- staging item of this thread: 4 consecutive x of staged row r_st, every channel
  const bool has_item = threadIdx.x < R * 16;
  const int r_st = threadIdx.x >> 4, q_st = threadIdx.x & 15;
  const int sy_st = y0 - 1 + r_st, x_st = xbase + 4 * q_st;
  const bool row_ok = has_item && sy_st >= 0 && sy_st < d.s1 && x_st >= 0 && x_st < d.s2;
  // loads are unconditional, from addresses clamped into the volume; what lies outside is zeroed on the way to LDS (a
  // load inside `if (inside)` gets its own exec-mask block and the compiler serialises such blocks with full waits)
  const int row_off = min(max(sy_st, 0), d.s1 - 1) * d.s2 + ((x_st >= 0 && x_st < d.s2) ? x_st : 0);
  const int lds_item = r_st * P + 4 + 4 * q_st;
  auto fetch = [&](int p, float (*v)[4]) {
    const uint32_t s = (uint32_t)(min(max(p, 0), d.s0 - 1) * plane_stride + row_off);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(inn + (size_t)c * V + s);
      v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
    }
  };
  auto commit = [&](int p, float (*v)[4]) {
    const bool ok = row_ok && p >= 0 && p < d.s0;
    float* slot = ring + (p & 3) * PS + lds_item;
#pragma unroll
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<float4*>(slot + c * R * P) = make_float4(ok ? v[c][0] : 0.f, ok ? v[c][1] : 0.f, ok ? v[c][2] : 0.f,
                                                                 ok ? v[c][3] : 0.f);
  };

  // ---- prologue: planes za-1, za, za+1 (all loads issued before the first LDS write)
  float pr[C][4];
  if (has_item) {
    float pa[C][4], pb[C][4];
    fetch(za - 1, pa);
    fetch(za, pb);
    fetch(za + 1, pr);
    commit(za - 1, pa);
    commit(za, pb);
    commit(za + 1, pr);
  }
  __syncthreads();

  float* const tr = trbuf + wave * G::TRW;
  const int own_row0 = wave * RPW;
  float dmax = 0.f;
  const float topx = (float)(d.s2 - 1), topy = (float)(d.s1 - 1), topz = (float)(d.s0 - 1);

  for (int z = za; z < zb; ++z) {
    const bool more = z + 2 <= zb;          // plane zb (= z+1 of the last step) is the last one needed
    if (has_item) fetch(z + 2, pr);      // (beyond the chunk: a clamped address, never committed)
    // grid values (and phi0 in final mode) of the owned rows: requested now, used below
    float g[RPW][3], p0v[RPW][3];
    if constexpr (!SELF) {
#pragma unroll
      for (int o = 0; o < RPW; ++o) {
        const int uy = min(y0 + own_row0 + o, d.s1 - 1);
        const uint32_t s = (uint32_t)((z * d.s1 + uy) * d.s2 + min(max(xl, 0), d.s2 - 1));
#pragma unroll
        for (int a = 0; a < 3; ++a) g[o][a] = gn[(size_t)a * V + s];
      }
    } else if (p0n) {
#pragma unroll
      for (int o = 0; o < RPW; ++o) {
        const int uy = min(y0 + own_row0 + o, d.s1 - 1);
        const uint32_t s = (uint32_t)((z * d.s1 + uy) * d.s2 + min(max(xl, 0), d.s2 - 1));
#pragma unroll
        for (int a = 0; a < 3; ++a) p0v[o][a] = p0n[(size_t)a * V + s];
      }
    }

    float res[C][RPW];
#pragma unroll
    for (int o = 0; o < RPW; ++o) {
      const int r = own_row0 + o + 1;       // staged row of this output row
      const int uy = y0 - 1 + r;
      if (uy >= d.s1) {                     // wave-uniform: row beyond the volume (partial last tile)
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][o] = 0.f;
        continue;
      }
      float gx, gy, gz;
      if constexpr (SELF) {
        const float* c0 = ring + (z & 3) * PS + r * P + 4 + lane;
        gx = c0[0]; gy = c0[R * P]; gz = c0[2 * R * P];
      } else {
        gx = g[o][0]; gy = g[o][1]; gz = g[o][2];
        if (MODE == kFwdClamp) { gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz); }
      }
      // unnormalised source coordinates (grid_sampler_unnormalize, align_corners); border padding clips them
      float xs = ((gx + 1.f) * 0.5f) * topx, ys = ((gy + 1.f) * 0.5f) * topy, zs = ((gz + 1.f) * 0.5f) * topz;
      if (PAD == PAD_BORDER) {
        xs = fminf(fmaxf(xs, 0.f), topx); ys = fminf(fmaxf(ys, 0.f), topy); zs = fminf(fmaxf(zs, 0.f), topz);
      }
      // NaN / huge: every corner out of range (med3 returns the smallest operand when one is NaN)
      xs = __builtin_amdgcn_fmed3f(xs, -16.f, 1.0e9f);
      ys = __builtin_amdgcn_fmed3f(ys, -16.f, 1.0e9f);
      zs = __builtin_amdgcn_fmed3f(zs, -16.f, 1.0e9f);
      const float fx = floorf(xs), fy = floorf(ys), fz = floorf(zs);
      const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
      const float wx1 = xs - fx, wx0 = (fx + 1.f) - xs;
      const float wy1 = ys - fy, wy0 = (fy + 1.f) - ys;
      const float wz1 = zs - fz, wz0 = (fz + 1.f) - zs;
      // both x corners in the staged 64 columns (a single segment also has the zero columns at x = -1 and x = S2)
      const bool xin = nseg > 1 ? (unsigned)(ix - xbase) <= 62u : (unsigned)(ix + 1) <= (unsigned)d.s2;
      const bool staged = (unsigned)(iz - z + 1) <= 1u && (unsigned)(iy - uy + 1) <= 1u && xin;
      // lanes whose 8 corners are in the ring take them from LDS; the others (displacement of a voxel or more) gather
      // from global memory with the same arithmetic.  Smooth fields make whole rows one or the other, and the empty
      // side of the branch is skipped (execz).
      if (staged) {
        float w[8];
        w[0] = (wx0 * wy0) * wz0; w[1] = (wx1 * wy0) * wz0; w[2] = (wx0 * wy1) * wz0; w[3] = (wx1 * wy1) * wz0;
        w[4] = (wx0 * wy0) * wz1; w[5] = (wx1 * wy0) * wz1; w[6] = (wx0 * wy1) * wz1; w[7] = (wx1 * wy1) * wz1;
        const int oxy = (r + (iy - uy)) * P + 4 + (ix - xbase);
        const float* q0 = ring + (iz & 3) * PS + oxy;
        const float* q1 = ring + ((iz + 1) & 3) * PS + oxy;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float acc = 0.f;
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
              for (int cx = 0; cx < 2; ++cx)     // product then sum, never contracted: see tap_acc() (sampler_common.h)
                acc = tap_acc<3>(acc, ((cz ? q1 : q0) + (c * R + cy) * P)[cx], w[(cz * 2 + cy) * 2 + cx]);
          res[c][o] = acc;
        }
      } else if (xowned) {
        Taps<3, PAD> t;
        t.build(gx, gy, gz, d);
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][o] = sample_linear<3, PAD, false>(inn + (size_t)c * V, t, d);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][o] = 0.f;
      }
      if constexpr (SELF) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v = res[c][o];
          const int sc = c == 0 ? xl : (c == 1 ? uy : z), Sc = c == 0 ? d.s2 : (c == 1 ? d.s1 : d.s0);
          if (final_mode == 1) v = (v - p0v[o][c]) + lin_coord(sc, Sc);   // (sample - phi0) + identity (adv_morph.py:143,176 + 474,483)
          res[c][o] = v;
          if (disp_out && xowned && uy < d.s1) dmax = fmaxf(dmax, voxel_displacement(v, Sc, sc));
        }
      }
    }

    // ---- results leave 4 voxels per lane through the wave's LDS scratch
    constexpr int NR = G::NA_ROUND;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += NR) {
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int o = 0; o < RPW; ++o)
          if (c0 + a < C) tr[(a * RPW + o) * 64 + lane] = res[c0 + a][o];
      lds_order();
      constexpr int ITEMS = NR * RPW * 16;
#pragma unroll
      for (int i0 = 0; i0 < ITEMS; i0 += 64) {
        const int j = i0 + lane;
        const int a = j / (RPW * 16), o = (j / 16) % RPW, q = j & 15;
        const float4 v4 = *reinterpret_cast<const float4*>(tr + (a * RPW + o) * 64 + 4 * q);
        const int xq = xbase + 4 * q;
        const bool valid = j < ITEMS && c0 + a < C && xq < d.s2 && (nseg == 1 || (q >= 1 && q <= 14)) && (y0 + own_row0 + o) < d.s1;
        if (valid)
          *reinterpret_cast<float4*>(outn + (size_t)(c0 + a) * V + (uint32_t)((z * d.s1 + y0 + own_row0 + o) * d.s2 + xq)) = v4;
      }
      lds_order();
    }

    if (has_item && more) commit(z + 2, pr);
    __syncthreads();
  }
  if (SELF && disp_out) wave_max_to_slots(dmax, disp_out);
}

// ---------------------------------------------------------------------------------------------
// Rows of 68 .. PW voxels (cfg-5: 160 x 160 x 80): the same march with lane <-> FLAT voxel of the tile's TY x W plane.
// The taps of a lane read LDS at addresses it computes itself, so nothing ties a lane to an x: a workgroup owns TY = 8
// whole rows, its TY * W voxels of a plane are cut into items of 64 consecutive voxels (whole rows are contiguous in
// memory: the grid loads and the 16-byte result stores of an item are one coalesced run) and wave w takes items
// w * IPW .. w * IPW + IPW - 1 -- every lane of every wave has a voxel (80 = 1.25 x 64: the x segments of the kernel above
// leave 48 of 128 lanes idle and the tile kernel, 164 us per squaring at 8 x 3 x 160 x 160 x 80, was the faster choice).
// Staged rows are W voxels between 4 zero floats; arithmetic, fallback and results are those of k_sample_march.
// ---------------------------------------------------------------------------------------------
template <int C, bool SELF, int PW, int IPW>
struct FlatMarchCfg {
  static constexpr int TY = 8;
  static constexpr int R = TY + 2;
  static constexpr int P = PW + 8;                           // 4 zeros | up to PW voxels | 4 zeros
  static constexpr int PS = C * R * P;
  static constexpr int NWMAX = (TY * PW + 64 * IPW - 1) / (64 * IPW);
  static constexpr int NA_ROUND = C > 2 ? 2 : C;
  static constexpr int TRW = NA_ROUND * IPW * 64;
  static size_t lds_bytes(int nw) { return (size_t)(4 * PS + nw * TRW) * sizeof(float); }
  static_assert(!SELF || C == 3, "the self-composition carries 3 channels");
};

template <int C, bool SELF, int MODE, int PW, int IPW>
__global__ void __launch_bounds__((FlatMarchCfg<C, SELF, PW, IPW>::NWMAX) * 64)
k_sample_march_flat(const float* __restrict__ in, const float* __restrict__ grid, float* __restrict__ out,
                    const float* __restrict__ phi0, Dims d, int n1, int zc, int final_mode, float* __restrict__ disp_out,
                    int xcd) {
  using G = FlatMarchCfg<C, SELF, PW, IPW>;
  constexpr int R = G::R, TY = G::TY, P = G::P, PS = G::PS;
  constexpr int PAD = MODE == kFwdBorder ? PAD_BORDER : PAD_ZEROS;
  constexpr bool LATE = SELF || C == 4;     // order of commit / request / stores at the end of a step (see there)
  extern __shared__ float lds[];
  float* const ring = lds;                  // [slot 4][C][R][P]
  float* const trbuf = lds + 4 * PS;        // [wave][TRW]
  const int V = (int)d.voxels();
  const int W = d.s2, W4 = d.s2 >> 2;
  const int NT = (int)blockDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tile = blockIdx.x + gridDim.x * blockIdx.y;
  const int tiles = gridDim.x * gridDim.y;
  if ((xcd & 1) && (tiles & 7) == 0) tile = (tile & 7) * (tiles >> 3) + (tile >> 3);   // XCD-contiguous tiles (see above)
  const int dbg = xcd >> 4;   // timing experiments only (ADVCHAIN_FWD_MARCH_DEBUG, results are wrong): 1 no taps, 2 no stores
  const int n = tile / (int)gridDim.x;
  const int rem = tile - n * (int)gridDim.x;
  const int ty = rem % n1, tz = rem / n1;
  const int y0 = ty * TY;
  const int za = tz * zc, zb = min(za + zc, d.s0);
  const float* inn = in + (int64_t)n * C * V;
  const float* gn = SELF ? nullptr : grid + (int64_t)n * 3 * V;
  const float* p0n = (SELF && final_mode == 1) ? phi0 + (int64_t)n * 3 * V : nullptr;
  float* outn = out + (int64_t)n * C * V;
  const int plane_stride = d.s1 * d.s2;
  const int flat_n = min(TY, d.s1 - y0) * W;       // voxels of this tile in a plane

  // ---- the zero columns of every staged row (never written again)
  for (int e = threadIdx.x; e < 4 * C * R * 2; e += NT) {
    const int row = e >> 1, side = e & 1;
    *reinterpret_cast<float4*>(lds + row * P + (side ? 4 + W : 0)) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- staging item of this thread: 4 consecutive x of staged row r_st, every channel
  const bool has_item = (int)threadIdx.x < R * W4;
  const int r_st = (int)threadIdx.x / W4, q_st = (int)threadIdx.x - r_st * W4;
  const int sy_st = y0 - 1 + r_st;
  const bool row_ok = has_item && sy_st >= 0 && sy_st < d.s1;
  const int row_off = min(max(sy_st, 0), d.s1 - 1) * W + (has_item ? 4 * q_st : 0);
  const int lds_item = r_st * P + 4 + 4 * q_st;
  auto fetch = [&](int p, float (*v)[4]) {
    const uint32_t s = (uint32_t)(min(max(p, 0), d.s0 - 1) * plane_stride + row_off);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(inn + (size_t)c * V + s);
      v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
    }
  };
  auto commit = [&](int p, float (*v)[4]) {
    const bool ok = row_ok && p >= 0 && p < d.s0;
    float* slot = ring + (p & 3) * PS + lds_item;
#pragma unroll
    for (int c = 0; c < C; ++c)
      *reinterpret_cast<float4*>(slot + c * R * P) = make_float4(ok ? v[c][0] : 0.f, ok ? v[c][1] : 0.f, ok ? v[c][2] : 0.f,
                                                                 ok ? v[c][3] : 0.f);
  };

  float pr[C][4];
  if (has_item) {
    float pa[C][4], pb[C][4];
    fetch(za - 1, pa);
    fetch(za, pb);
    fetch(za + 1, pr);
    commit(za - 1, pa);
    commit(za, pb);
    commit(za + 1, pr);
    if constexpr (LATE) fetch(za + 2, pr);
  }
  __syncthreads();

  // ---- the voxels of this lane: item i of the wave -> flat voxel fc[i] = staged row orow[i] + 1, column ox[i]
  int fc[IPW], orow[IPW], ox[IPW];
  bool valid[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int f = (wave * IPW + i) * 64 + lane;
    valid[i] = f < flat_n;
    fc[i] = min(f, flat_n - 1);
    orow[i] = fc[i] / W;
    ox[i] = fc[i] - orow[i] * W;
  }

  float* const tr = trbuf + wave * G::TRW;
  float dmax = 0.f;
  const float topx = (float)(d.s2 - 1), topy = (float)(d.s1 - 1), topz = (float)(d.s0 - 1);

  for (int z = za; z < zb; ++z) {
    const bool more = z + 2 <= zb;
    if constexpr (!LATE) {
      if (has_item) fetch(z + 2, pr);
    }
    const uint32_t tile_off = (uint32_t)((z * d.s1 + y0) * W);
    float g[IPW][3], p0v[IPW][3];
    if constexpr (!SELF) {
#pragma unroll
      for (int i = 0; i < IPW; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) g[i][a] = gn[(size_t)a * V + tile_off + fc[i]];
    } else if (p0n) {
#pragma unroll
      for (int i = 0; i < IPW; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) p0v[i][a] = p0n[(size_t)a * V + tile_off + fc[i]];
    }

    float res[C][IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int r = orow[i] + 1;            // staged row of this voxel
      const int uy = y0 + orow[i];
      const int xl = ox[i];
      float gx, gy, gz;
      if constexpr (SELF) {
        const float* c0 = ring + (z & 3) * PS + r * P + 4 + xl;
        gx = c0[0]; gy = c0[R * P]; gz = c0[2 * R * P];
      } else {
        gx = g[i][0]; gy = g[i][1]; gz = g[i][2];
        if (MODE == kFwdClamp) { gx = clamp_unit(gx); gy = clamp_unit(gy); gz = clamp_unit(gz); }
      }
      float xs = ((gx + 1.f) * 0.5f) * topx, ys = ((gy + 1.f) * 0.5f) * topy, zs = ((gz + 1.f) * 0.5f) * topz;
      if (PAD == PAD_BORDER) {
        xs = fminf(fmaxf(xs, 0.f), topx); ys = fminf(fmaxf(ys, 0.f), topy); zs = fminf(fmaxf(zs, 0.f), topz);
      }
      xs = __builtin_amdgcn_fmed3f(xs, -16.f, 1.0e9f);
      ys = __builtin_amdgcn_fmed3f(ys, -16.f, 1.0e9f);
      zs = __builtin_amdgcn_fmed3f(zs, -16.f, 1.0e9f);
      const float fx = floorf(xs), fy = floorf(ys), fz = floorf(zs);
      const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
      const float wx1 = xs - fx, wx0 = (fx + 1.f) - xs;
      const float wy1 = ys - fy, wy0 = (fy + 1.f) - ys;
      const float wz1 = zs - fz, wz0 = (fz + 1.f) - zs;
      const bool staged = (unsigned)(iz - z + 1) <= 1u && (unsigned)(iy - uy + 1) <= 1u && (unsigned)(ix + 1) <= (unsigned)W;
      if (dbg & 1) {
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][i] = wx1 + wy1 + wz1;
      } else if (staged) {
        float w[8];
        w[0] = (wx0 * wy0) * wz0; w[1] = (wx1 * wy0) * wz0; w[2] = (wx0 * wy1) * wz0; w[3] = (wx1 * wy1) * wz0;
        w[4] = (wx0 * wy0) * wz1; w[5] = (wx1 * wy0) * wz1; w[6] = (wx0 * wy1) * wz1; w[7] = (wx1 * wy1) * wz1;
        const int oxy = (r + (iy - uy)) * P + 4 + ix;
        const float* q0 = ring + (iz & 3) * PS + oxy;
        const float* q1 = ring + ((iz + 1) & 3) * PS + oxy;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float acc = 0.f;
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
              for (int cx = 0; cx < 2; ++cx)     // product then sum, never contracted: see tap_acc() (sampler_common.h)
                acc = tap_acc<3>(acc, ((cz ? q1 : q0) + (c * R + cy) * P)[cx], w[(cz * 2 + cy) * 2 + cx]);
          res[c][i] = acc;
        }
      } else if (valid[i]) {
        Taps<3, PAD> t;
        t.build(gx, gy, gz, d);
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][i] = sample_linear<3, PAD, false>(inn + (size_t)c * V, t, d);
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) res[c][i] = 0.f;
      }
      if constexpr (SELF) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float v = res[c][i];
          const int sc = c == 0 ? xl : (c == 1 ? uy : z), Sc = c == 0 ? d.s2 : (c == 1 ? d.s1 : d.s0);
          if (final_mode == 1) v = (v - p0v[i][c]) + lin_coord(sc, Sc);
          res[c][i] = v;
          if (disp_out && valid[i]) dmax = fmaxf(dmax, voxel_displacement(v, Sc, sc));
        }
      }
    }

    // LATE: the plane requested a step ago goes to LDS (its slot held plane z-2: nobody reads it in this step) and the next
    // one is requested BEFORE this step's stores: a wait for loads that are older than conditional stores has to be
    // vmcnt(0) and would wait for the stores' acknowledgements every step (adjoint_march.hip).  Same-box A/B at
    // 8 x . x 160 x 160 x 80: self-composition 124.8 -> 117.6 us, C = 4 184.7 -> 177 us, C = 1 unchanged (it keeps the
    // old order; so does k_sample_march, where the late order measured 3-8 % slower).
    if constexpr (LATE) {
      if (has_item) { commit(z + 2, pr); fetch(z + 3, pr); }
    }
    // ---- results leave 4 voxels per lane through the wave's LDS scratch (an item is 64 consecutive voxels in memory)
    constexpr int NR = G::NA_ROUND;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += NR) {
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int i = 0; i < IPW; ++i)
          if (c0 + a < C) tr[(a * IPW + i) * 64 + lane] = res[c0 + a][i];
      lds_order();
      constexpr int ITEMS = NR * IPW * 16;
#pragma unroll
      for (int i0 = 0; i0 < ITEMS; i0 += 64) {
        const int j = i0 + lane;
        const int a = j / (IPW * 16), i = (j / 16) % IPW, q = j & 15;
        const float4 v4 = *reinterpret_cast<const float4*>(tr + (a * IPW + i) * 64 + 4 * q);
        const int f4 = (wave * IPW + i) * 64 + 4 * q;
        if (j < ITEMS && c0 + a < C && f4 < flat_n && !(dbg & 2))
          *reinterpret_cast<float4*>(outn + (size_t)(c0 + a) * V + tile_off + (uint32_t)f4) = v4;
      }
      lds_order();
    }

    if constexpr (!LATE) {
      if (has_item && more) commit(z + 2, pr);
    }
    __syncthreads();
  }
  if (SELF && disp_out) wave_max_to_slots(dmax, disp_out);
}

// (Round 3 kept an experiment here -- two 3D squarings per launch through a second LDS ring, k_compose2_march behind
// ADVCHAIN_FUSE2: 3-17 % slower than one launch per squaring because the second ring halved the resident workgroups of a
// kernel that is 47 % parked.  Measurements and counters: profiles/r03/f1/; the code was removed in round 4.)

}  // namespace advchain

using namespace advchain;

static int fwd_march_zc(const Dims& d, int64_t N, int ty, int C) {
  static const int forced = getenv("ADVCHAIN_FWD_MARCH_ZC") ? atoi(getenv("ADVCHAIN_FWD_MARCH_ZC")) : 0;   // A/B knob (0: the rule below)
  if (forced > 0) return forced;
  // measured at 4 x C x 128 x 128 x 64: the one-channel warp is latency-bound and wants 8 workgroups per CU even at
  // chunks of 4 planes (15.4 us against 17.7 at 8 and 25 at 16 planes); 3-4 channels want 4 per CU and chunks >= 8
  // (a measured dead end: handing the grid values to their lanes through LDS with 16-byte loads -- 18.9 us)
  const int64_t cols = N * ((d.s1 + ty - 1) / ty);
  const int want = C == 1 ? 2048 : 1024, floor_zc = C == 1 ? 4 : 8;
  int zc = d.s0;
  while (zc > floor_zc && cols * ((d.s0 + zc - 1) / zc) < want) zc = (zc + 1) / 2;
  return zc;
}

template <int C, bool SELF, int MODE, int NW, int RPW>
static void launch_fwd_march(const float* in, const float* grid, float* out, const float* phi0, int64_t N, Dims d,
                             int final_mode, float* disp_out, hipStream_t st) {
  using G = FwdMarchCfg<C, SELF, NW, RPW>;
  auto kern = k_sample_march<C, SELF, MODE, NW, RPW>;
  // occupancy experiment (round 6, profiles/r06/f1_3d/): ADVCHAIN_MARCH_LDS_PAD = extra dynamic LDS in KiB that nobody touches --
  // the kernel then runs with the workgroups-per-CU a fused two-level kernel's LDS footprint would leave it (results unchanged)
  static const size_t lds_pad = getenv("ADVCHAIN_MARCH_LDS_PAD") ? (size_t)atoi(getenv("ADVCHAIN_MARCH_LDS_PAD")) * 1024 : 0;
  static bool attr_set = false;
  if (G::LDS + lds_pad > 65536 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(G::LDS + lds_pad));
    attr_set = true;
  }
  const int n1 = (d.s1 + G::TY - 1) / G::TY;
  const int nseg = d.s2 <= 64 ? 1 : (d.s2 + kFwdSegOwn - 1) / kFwdSegOwn;
  const int zc = fwd_march_zc(d, N * nseg, G::TY, C);
  const int n0 = (d.s0 + zc - 1) / zc;
  static const bool no_xcd = false;   // measured optimum (was a tuning knob until round 4)
  hipLaunchKernelGGL(kern, dim3((unsigned)(nseg * n1 * n0), (unsigned)N), dim3(G::NT), G::LDS + lds_pad, st, in, grid, out, phi0, d, n1, zc,
                     final_mode, disp_out, no_xcd ? -nseg : nseg);
}

// rows of 68 .. 80 (PW = 80) and 84 .. 128 (PW = 128) voxels: lane <-> flat voxel (k_sample_march_flat)
constexpr int kFlatPW = 80, kFlatPW2 = 128;
template <int C, bool SELF, int MODE, int PW = kFlatPW>
static void launch_fwd_flat(const float* in, const float* grid, float* out, const float* phi0, int64_t N, Dims d,
                            int final_mode, float* disp_out, hipStream_t st) {
  constexpr int IPW = 2;
  if constexpr (PW == kFlatPW) {
    if (d.s2 > kFlatPW) return launch_fwd_flat<C, SELF, MODE, kFlatPW2>(in, grid, out, phi0, N, d, final_mode, disp_out, st);
  }
  using G = FlatMarchCfg<C, SELF, PW, IPW>;
  auto kern = k_sample_march_flat<C, SELF, MODE, PW, IPW>;
  const int nw = (G::TY * d.s2 + 64 * IPW - 1) / (64 * IPW);
  const size_t lds = G::lds_bytes(G::NWMAX);
  static bool attr_set = false;
  if (lds > 65536 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int n1 = (d.s1 + G::TY - 1) / G::TY;
  const int zc = fwd_march_zc(d, N, G::TY, C);
  const int n0 = (d.s0 + zc - 1) / zc;
  static const bool no_xcd = false;   // measured optimum (was a tuning knob until round 4)
  static const int dbg = getenv("ADVCHAIN_FWD_MARCH_DEBUG") ? atoi(getenv("ADVCHAIN_FWD_MARCH_DEBUG")) : 0;   // timing experiments
  hipLaunchKernelGGL(kern, dim3((unsigned)(n1 * n0), (unsigned)N), dim3(nw * 64), G::lds_bytes(nw), st, in, grid, out, phi0, d, n1,
                     zc, final_mode, disp_out, (no_xcd ? 0 : 1) | (dbg << 4));
}

template <int C>
static void launch_fwd_flat_mode(int mode, const float* in, const float* grid, float* out, int64_t N, Dims d, hipStream_t st) {
  if (mode == kFwdBorder) launch_fwd_flat<C, false, kFwdBorder>(in, grid, out, nullptr, N, d, 0, nullptr, st);
  else if (mode == kFwdClamp) launch_fwd_flat<C, false, kFwdClamp>(in, grid, out, nullptr, N, d, 0, nullptr, st);
  else launch_fwd_flat<C, false, kFwdFree>(in, grid, out, nullptr, N, d, 0, nullptr, st);
}

template <int C>
static void launch_fwd_march_mode(int mode, const float* in, const float* grid, float* out, int64_t N, Dims d, hipStream_t st) {
  if (mode == kFwdBorder) launch_fwd_march<C, false, kFwdBorder, 4, 2>(in, grid, out, nullptr, N, d, 0, nullptr, st);
  else if (mode == kFwdClamp) launch_fwd_march<C, false, kFwdClamp, 4, 2>(in, grid, out, nullptr, N, d, 0, nullptr, st);
  else launch_fwd_march<C, false, kFwdFree, 4, 2>(in, grid, out, nullptr, N, d, 0, nullptr, st);
}

// 3D forward warps at equal size with rows of at most 64 voxels.  ADVCHAIN_ERR_UNSUPPORTED: use the tile kernel.
int advchain_sample_march_launch(bool self, const float* in, const float* grid, float* out, const float* phi0, int64_t N,
                                 int64_t C, Dims d, int padding, int clamp_grid, int final_mode, float* disp_out,
                                 hipStream_t st) {
  static const bool off = getenv("ADVCHAIN_NO_MARCH_FWD") != nullptr;   // A/B knob
  if (off || padding == PAD_REFLECTION) return ADVCHAIN_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out);
  if (d.s2 < 8 || (d.s2 & 3) != 0 || (al & 15) != 0 || d.s0 < 2 || d.voxels() * 4 >= (1ll << 31))
    return ADVCHAIN_ERR_UNSUPPORTED;
  static const bool no_flat = getenv("ADVCHAIN_NO_FLAT_FWD") != nullptr;   // A/B knob
  const bool flat = !no_flat && d.s2 > 64 && d.s2 <= kFlatPW2;
  if (self) {
    if (C != 3) return ADVCHAIN_ERR_UNSUPPORTED;
    if (flat) launch_fwd_flat<3, true, kFwdBorder>(in, nullptr, out, phi0, N, d, final_mode, disp_out, st);
    else launch_fwd_march<3, true, kFwdBorder, 4, 2>(in, nullptr, out, phi0, N, d, final_mode, disp_out, st);
  } else {
    const int mode = padding == PAD_BORDER ? kFwdBorder : (clamp_grid ? kFwdClamp : kFwdFree);
    // (border padding with clamp_grid: the clamp is implied by the clip of the source coordinate)
    if (C == 1) flat ? launch_fwd_flat_mode<1>(mode, in, grid, out, N, d, st) : launch_fwd_march_mode<1>(mode, in, grid, out, N, d, st);
    // (four channels of rows beyond 80 voxels: the ring of the flat form takes 87 KiB, one workgroup a CU -- the x segments win,
    // 81 against 105 us at 8 x 4 x 96 x 96 x 96)
    else if (C == 4) (flat && d.s2 <= kFlatPW) ? launch_fwd_flat_mode<4>(mode, in, grid, out, N, d, st) : launch_fwd_march_mode<4>(mode, in, grid, out, N, d, st);
    else return ADVCHAIN_ERR_UNSUPPORTED;
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}
