// Affine warp through an LDS-staged source box (gfx950).
//
//   advchain_affine_warp_fwd / _bwd  <-  F.affine_grid + F.grid_sample, reference adv_affine.py:297-313
//   (linear interpolation, zeros padding -- the only mode AdvAffine uses, Q9 -- S2 % 4 == 0, 16-byte aligned tensors;
//   everything else stays on the direct-gather kernels of sampler.hip)
//
// Why: the direct-gather kernels issue 2^d dword gathers per voxel and channel, and a CU retires one vector-memory
// wave-instruction per ~26 clk whatever it carries (DESIGN lesson 4): C = 4 forward = 36 instructions per 64 voxels
// = 100 us at 4x4x128x128x64, 75 % of the wave-cycles stalled on issue (profiles/r02/sq_issue_summary.txt) at 13 % of
// the HBM roofline.  An affine map sends an output tile to a parallelepiped of the source: a workgroup takes a
// TX x TY x TZ output tile, computes the taps of its voxels ONCE (positions with the forward's exact arithmetic),
// reduces the integer bounding box of their corners over the workgroup (exact: no assumption about rounding), and per
// channel stages that box with 16-byte loads (1 KiB per wave-instruction instead of 256 B, every line asked for once
// per workgroup) into LDS, zero-filled outside the volume -- zeros padding becomes DATA: a two-cell zero border means
// corner validity needs no select and the +1 corners sit at fixed LDS offsets (ds_read2_b32 pairs).  The 2^d corners
// of every voxel then come from LDS.  A tile whose box does not fit (strong minification, NaN theta) takes the
// direct gathers, block-uniformly.  The theta-gradient kernel is the same walk with the derivative weights.
#include <stdlib.h>
#include "sampler_common.h"
#include "affine_geo.h"

namespace advchain {

// Tile geometry.  3D: 8 planes = 8 voxels a thread, least box inflation, 52 KiB and ~150 VGPRs (3 workgroups a CU);
// 4 planes (38 KiB, ~100 VGPRs, 4 workgroups a CU, 1.4x the staged bytes) measured slower and was removed.
template <int DIM, int TZ_> struct BoxGeom;
template <> struct BoxGeom<3, 8> { static constexpr int TX = 16, TY = 16, TZ = 8, CAP = 13312, LPR = 8, WGS = 3; };
template <> struct BoxGeom<2, 1> { static constexpr int TX = 32, TY = 32, TZ = 1, CAP = 4096, LPR = 16, WGS = 4; };   // 16 KiB

template <int DIM, int TZ>
__host__ __device__ inline int box_tiles(const Dims& d) {
  using G = BoxGeom<DIM, TZ>;
  return ((d.s2 + G::TX - 1) / G::TX) * ((d.s1 + G::TY - 1) / G::TY) * ((d.s0 + G::TZ - 1) / G::TZ);
}

// block b -> tile: blocks are dealt to the XCDs round-robin (b % 8); give every XCD a CONTIGUOUS run of tiles so that
// the boxes of neighbouring tiles, which overlap, are served by one L2
__device__ __forceinline__ int xcd_contiguous(int b, int nb) {
  if (nb & 7) return b;
  return (b & 7) * (nb >> 3) + (b >> 3);
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

struct BoxDesc {
  int x0, y0, z0;   // volume coordinates of box cell (0,0,0); x0 % 4 == 0
  int ex, ey, ez;   // extents in cells; ex % 4 == 0
  bool fits;
};

// The taps of one voxel, kept for all channels: lower corner and the upper-corner weights w1 = x - i0.  The lower
// weights are 1 - w1 EXACTLY: make_tap's w0 = (i0 + 1) - x is exact (x - i0 is, and 1 - w1 is a multiple of ulp(x) below
// one), so nothing is lost by not storing them.
struct VoxTap {
  int ix, iy, iz;        // lower corner, clamped into [-2, S]
  float fx, fy, fz;
};

template <int DIM>
__device__ __forceinline__ VoxTap make_vox_tap(const Theta<DIM>& th, int ox, int oy, int oz, const Dims& d) {
  float bx, by, bz, gx, gy, gz;
  affine_position_xyz<DIM>(th, ox, oy, oz, d, bx, by, bz, gx, gy, gz);
  const AxisTap tx = make_tap<PAD_ZEROS>(gx, d.s2), ty = make_tap<PAD_ZEROS>(gy, d.s1);
  VoxTap v;
  v.ix = min(max(tx.i0, -2), d.s2);     // corners beyond the volume read the zero border: [-2,-1] and [S, S+1]
  v.iy = min(max(ty.i0, -2), d.s1);
  v.fx = tx.w1; v.fy = ty.w1;
  if (DIM == 3) {
    const AxisTap tz = make_tap<PAD_ZEROS>(gz, d.s0);
    v.iz = min(max(tz.i0, -2), d.s0);
    v.fz = tz.w1;
  } else {
    v.iz = 0; v.fz = 0.f;
  }
  return v;
}

// Exact bounding box of the corners of all voxels of the workgroup (every thread passes the min / max over its own).
template <int DIM, int TZ>
__device__ __forceinline__ BoxDesc reduce_box(int lox, int hix, int loy, int hiy, int loz, int hiz, int (*red)[6]) {
  using G = BoxGeom<DIM, TZ>;
  lox = wave_min_i(lox); hix = wave_max_i(hix);
  loy = wave_min_i(loy); hiy = wave_max_i(hiy);
  if (DIM == 3) { loz = wave_min_i(loz); hiz = wave_max_i(hiz); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = lox; red[wave][1] = hix; red[wave][2] = loy; red[wave][3] = hiy; red[wave][4] = loz; red[wave][5] = hiz;
  }
  __syncthreads();
  lox = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
  hix = max(max(red[0][1], red[1][1]), max(red[2][1], red[3][1]));
  loy = min(min(red[0][2], red[1][2]), min(red[2][2], red[3][2]));
  hiy = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
  loz = min(min(red[0][4], red[1][4]), min(red[2][4], red[3][4]));
  hiz = max(max(red[0][5], red[1][5]), max(red[2][5], red[3][5]));
  BoxDesc b;
  b.x0 = lox & ~3;                                  // 16-byte staging loads (two's complement: -2 -> -4)
  b.ex = ((hix + 1 - b.x0 + 1) + 3) & ~3;           // corners ix .. ix + 1
  b.y0 = loy; b.ey = hiy + 1 - loy + 1;
  b.z0 = DIM == 3 ? loz : 0; b.ez = DIM == 3 ? hiz + 1 - loz + 1 : 1;
  b.fits = b.ex <= 4 * G::LPR && b.ex * b.ey * b.ez <= G::CAP;
  return b;
}

// Stage the box of one channel: rows of ex / 4 float4s, LPR lanes a row, loads unconditional from clamped addresses and
// zeroed by selects on the way to LDS (a conditional load is a serial load: DESIGN lessons 15, 20).  In two halves, so that
// the loads of the NEXT channel's box are in flight under the arithmetic of the current one (a workgroup is two or three
// to a CU here: nobody else hides the round trip): box_request() asks for the first NP passes of U rows into registers,
// box_commit() writes them to LDS and stages whatever the box has beyond them the plain way.
constexpr int kBoxU = 4;     // 16-byte loads in flight per lane and pass
template <int NP>
struct BoxRegs {
  float4 v[NP][kBoxU];
  unsigned in;               // bit (p * kBoxU + u): the row lies inside the volume
};

template <int DIM, int TZ, int NP>
__device__ __forceinline__ void box_request(const float* __restrict__ src, const BoxDesc& b, const Dims& d, BoxRegs<NP>& pf) {
  using G = BoxGeom<DIM, TZ>;
  constexpr int RPP = kBlock / G::LPR;               // rows per pass
  const int qx = threadIdx.x % G::LPR;
  const int r0 = threadIdx.x / G::LPR;
  const int rows = b.ey * b.ez;
  const int gx = b.x0 + 4 * qx;
  const bool xin = gx >= 0 && gx + 4 <= d.s2;
  const int cx = min(max(gx, 0), d.s2 - 4);
  const float inv_ey = 1.f / (float)b.ey;
  pf.in = 0u;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int u = 0; u < kBoxU; ++u) {
      const int r = min(r0 + (p * kBoxU + u) * RPP, rows - 1);
      const int rz = DIM == 3 ? (int)(((float)r + 0.5f) * inv_ey) : 0;
      const int ry = r - rz * b.ey;
      const int gy = b.y0 + ry, gz = b.z0 + rz;
      if (xin && gy >= 0 && gy < d.s1 && gz >= 0 && gz < d.s0) pf.in |= 1u << (p * kBoxU + u);
      const int cy = min(max(gy, 0), d.s1 - 1), cz = min(max(gz, 0), d.s0 - 1);
      pf.v[p][u] = *reinterpret_cast<const float4*>(src + ((int64_t)cz * d.s1 + cy) * d.s2 + cx);
    }
}

template <int DIM, int TZ, int NP>
__device__ __forceinline__ void box_commit(const float* __restrict__ src, const BoxDesc& b, const Dims& d, float* box,
                                           const BoxRegs<NP>& pf) {
  using G = BoxGeom<DIM, TZ>;
  constexpr int RPP = kBlock / G::LPR;
  const int qx = threadIdx.x % G::LPR;
  const int r0 = threadIdx.x / G::LPR;
  const int rows = b.ey * b.ez;
  const bool lane_on = 4 * qx < b.ex;
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int u = 0; u < kBoxU; ++u) {
      const int r = r0 + (p * kBoxU + u) * RPP;
      const bool in = (pf.in >> (p * kBoxU + u)) & 1u;
      const float4 v = pf.v[p][u];
      if (lane_on && r < rows)     // component selects (a select between two float4 OBJECTS goes through scratch memory)
        *reinterpret_cast<float4*>(box + r * b.ex + 4 * qx) = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
    }
  // ---- rows beyond the requested passes (large boxes: strong rotation / minification)
  const int gx = b.x0 + 4 * qx;
  const bool xin = gx >= 0 && gx + 4 <= d.s2;
  const int cx = min(max(gx, 0), d.s2 - 4);
  const float inv_ey = 1.f / (float)b.ey;
  for (int rb = r0 + NP * kBoxU * RPP; rb < rows; rb += RPP * kBoxU) {
    float4 v[kBoxU];
    bool in[kBoxU];
#pragma unroll
    for (int u = 0; u < kBoxU; ++u) {
      const int r = min(rb + u * RPP, rows - 1);
      const int rz = DIM == 3 ? (int)(((float)r + 0.5f) * inv_ey) : 0;
      const int ry = r - rz * b.ey;
      const int gy = b.y0 + ry, gz = b.z0 + rz;
      in[u] = xin && gy >= 0 && gy < d.s1 && gz >= 0 && gz < d.s0;
      const int cy = min(max(gy, 0), d.s1 - 1), cz = min(max(gz, 0), d.s0 - 1);
      v[u] = *reinterpret_cast<const float4*>(src + ((int64_t)cz * d.s1 + cy) * d.s2 + cx);
    }
#pragma unroll
    for (int u = 0; u < kBoxU; ++u) {
      const int r = rb + u * RPP;
      if (lane_on && r < rows)
        *reinterpret_cast<float4*>(box + r * b.ex + 4 * qx) =
            make_float4(in[u] ? v[u].x : 0.f, in[u] ? v[u].y : 0.f, in[u] ? v[u].z : 0.f, in[u] ? v[u].w : 0.f);
    }
  }
}

template <int DIM, int TZ>
__device__ __forceinline__ void tile_origin(const Dims& d, int& x0, int& y0, int& z0) {
  using G = BoxGeom<DIM, TZ>;
  const int ntx = (d.s2 + G::TX - 1) / G::TX, nty = (d.s1 + G::TY - 1) / G::TY;
  const int t = xcd_contiguous(blockIdx.x, gridDim.x);
  const int r = t / ntx;
  x0 = (t - r * ntx) * G::TX;
  z0 = (r / nty) * G::TZ;
  y0 = (r - (r / nty) * nty) * G::TY;
}

// voxel k of thread t inside the tile
template <int DIM>
__device__ __forceinline__ void local_voxel(int k, int& lx, int& ly, int& lz) {
  if (DIM == 3) { lx = threadIdx.x & 15; ly = (threadIdx.x >> 4) & 15; lz = k; }
  else { lx = threadIdx.x & 31; ly = (threadIdx.x >> 5) + 8 * k; lz = 0; }
}

template <int DIM>
__device__ __forceinline__ Theta<DIM> load_theta(const float* __restrict__ theta, int n) {
  Theta<DIM> th;
#pragma unroll
  for (int r = 0; r < DIM; ++r)
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) th.m[r][c] = theta[(int64_t)n * DIM * (DIM + 1) + r * (DIM + 1) + c];
  return th;
}

template <int DIM>
__device__ __forceinline__ Taps<DIM, PAD_ZEROS> rebuild_taps(const Theta<DIM>& th, int ox, int oy, int oz, const Dims& d) {
  float bx, by, bz, gx, gy, gz;
  affine_position_xyz<DIM>(th, ox, oy, oz, d, bx, by, bz, gx, gy, gz);
  Taps<DIM, PAD_ZEROS> t;
  t.build(gx, gy, gz, d);
  return t;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int DIM, int TZ>
__global__ void __launch_bounds__(kBlock, (BoxGeom<DIM, TZ>::WGS))     // as many workgroups a CU as the LDS allows
k_affine_box_fwd(const float* __restrict__ in, const float* __restrict__ theta, float* __restrict__ out, int C, Dims d,
                 const float* __restrict__ ride_in, float* __restrict__ ride_out, int ride_nonzero) {
  using G = BoxGeom<DIM, TZ>;
  constexpr int VPT = G::TX * G::TY * G::TZ / kBlock;
  __shared__ __attribute__((aligned(16))) float box[G::CAP];
  __shared__ int red[4][6];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  const Theta<DIM> th = load_theta<DIM>(theta, n);
  int tx0, ty0, tz0;
  tile_origin<DIM, TZ>(d, tx0, ty0, tz0);
  // per voxel and for all channels: the LDS cell of the lower corner (box-relative) and three fractions
  int cell[VPT];
  float fx[VPT], fy[VPT], fz[VPT];
  int lox = 1 << 30, hix = -(1 << 30), loy = lox, hiy = hix, loz = lox, hiz = hix;
  {
    int ixs[VPT], iys[VPT], izs[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      // overhanging threads take a voxel of the volume: their taps exist, do not widen the box much, are not stored
      const VoxTap t = make_vox_tap<DIM>(th, min(tx0 + lx, d.s2 - 1), min(ty0 + ly, d.s1 - 1), min(tz0 + lz, d.s0 - 1), d);
      ixs[k] = t.ix; iys[k] = t.iy; izs[k] = t.iz; fx[k] = t.fx; fy[k] = t.fy; fz[k] = t.fz;
      lox = min(lox, t.ix); hix = max(hix, t.ix);
      loy = min(loy, t.iy); hiy = max(hiy, t.iy);
      loz = min(loz, t.iz); hiz = max(hiz, t.iz);
      if (k & 1) __builtin_amdgcn_sched_barrier(0);
    }
    const BoxDesc bb = reduce_box<DIM, TZ>(lox, hix, loy, hiy, loz, hiz, red);
#pragma unroll
    for (int k = 0; k < VPT; ++k) cell[k] = ((izs[k] - bb.z0) * bb.ey + (iys[k] - bb.y0)) * bb.ex + (ixs[k] - bb.x0);
    lox = bb.x0; hix = bb.ex; loy = bb.y0; hiy = bb.ey; loz = bb.z0; hiz = bb.ez;
    if (!bb.fits) hix = 0;
  }
  BoxDesc b;
  b.x0 = lox; b.ex = hix; b.y0 = loy; b.ey = hiy; b.z0 = loz; b.ez = hiz; b.fits = hix != 0;
  const float* inn = in + (int64_t)n * C * V;
  float* on = out + (int64_t)n * C * V;
  // the rider (advchain_affine_warp_fwd_ride): channel C of the walk is one channel of another tensor
  const int CT = ride_out ? C + 1 : C;
  const float* rin = ride_out ? ride_in + (int64_t)n * V : nullptr;
  float* ron = ride_out ? ride_out + (int64_t)n * V : nullptr;
  if (!b.fits) {      // block-uniform: direct gathers (the arithmetic of k_affine_warp_fwd); no register array is
#pragma unroll 1      // indexed by the run-time k here (that would move the arrays to scratch for the fast path too)
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
      if (!(ox < d.s2 && oy < d.s1 && oz < d.s0)) continue;
      const Taps<DIM, PAD_ZEROS> t = rebuild_taps<DIM>(th, ox, oy, oz, d);
      const int o = (oz * d.s1 + oy) * d.s2 + ox;
      for (int c = 0; c < C; ++c) on[(int64_t)c * V + o] = sample_linear<DIM, PAD_ZEROS>(inn + (int64_t)c * V, t, d);
      if (ride_out) {
        const float r = sample_linear<DIM, PAD_ZEROS>(rin, t, d);
        ron[o] = ride_nonzero ? (r != 0.f ? 1.f : 0.f) : r;
      }
    }
    return;
  }
  const int sy = b.ex, sz = b.ex * b.ey;
  constexpr int NP = 1;                    // passes of the next box held in registers (two would cost the third workgroup a CU)
  auto chan = [&](int c) { return c == C ? rin : inn + (int64_t)c * V; };     // (block-uniform)
  BoxRegs<NP> pf;
  box_request<DIM, TZ, NP>(chan(0), b, d, pf);
  for (int c = 0; c < CT; ++c) {
    if (c > 0) __syncthreads();            // everyone is done reading the previous channel's box
    const bool rider = c == C;
    box_commit<DIM, TZ, NP>(chan(c), b, d, box, pf);
    float* oc = rider ? ron : on + (int64_t)c * V;
    __syncthreads();
    if (c + 1 < CT) box_request<DIM, TZ, NP>(chan(c + 1), b, d, pf);   // in flight under this channel's lerps
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const float* p = box + cell[k];
      // nested lerps x -> y -> z with the taps' own weights (w1 = x - i0, w0 = 1 - w1 exactly): 17 VALU operations
      // per voxel and channel instead of 8 x (2 products + fma); agrees with the corner-sum form to rounding
      const float gx = 1.f - fx[k], gy = 1.f - fy[k];
      float a[4];
#pragma unroll
      for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
          a[cz * 2 + cy] = fmaf(p[cz * sz + cy * sy + 1], fx[k], p[cz * sz + cy * sy] * gx);
      float acc = fmaf(a[1], fy[k], a[0] * gy);
      if (DIM == 3) acc = fmaf(fmaf(a[3], fy[k], a[2] * gy), fz[k], acc * (1.f - fz[k]));
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
      if (rider && ride_nonzero) acc = acc != 0.f ? 1.f : 0.f;
      if (ox < d.s2 && oy < d.s1 && oz < d.s0) oc[(oz * d.s1 + oy) * d.s2 + ox] = acc;
      // two voxels' LDS reads in flight at a time: left alone the scheduler hoists all 8 x 2^d reads (64 live
      // registers on top of the taps) and the kernel no longer fits three workgroups a CU
      if (k & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// theta gradient: gtheta_partial (N, gridDim.x, DIM*(DIM+1)) block partial sums (k_reduce_partials of sampler.hip)
// ---------------------------------------------------------------------------------------------
template <int DIM, int TZ>
__global__ void __launch_bounds__(kBlock, (DIM == 3 ? 2 : 4))   // (2D at 3 waves a SIMD, no spills: 27.7 against 25.8 us)
k_affine_box_gtheta(const float* __restrict__ gout, const float* __restrict__ in, const float* __restrict__ theta,
                    float* __restrict__ gtheta_partial, int C, Dims d, float* __restrict__ tilemax, float* __restrict__ geo,
                    int* __restrict__ gmode) {
  using G = BoxGeom<DIM, TZ>;
  constexpr int VPT = G::TX * G::TY * G::TZ / kBlock;
  constexpr int NT = DIM * (DIM + 1);
  __shared__ __attribute__((aligned(16))) float box[G::CAP];
  __shared__ int red[4][6];
  __shared__ float smem[4 * NT];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  // the per-sample geometry the grad_in kernel behind this launch reads (k_affine_geometry's job, on the side)
  if (geo && blockIdx.x == 0 && threadIdx.x == 0) affine_geometry_one<DIM>(theta, geo, gmode, n, d);
  const Theta<DIM> th = load_theta<DIM>(theta, n);
  int tx0, ty0, tz0;
  tile_origin<DIM, TZ>(d, tx0, ty0, tz0);
  int cell[VPT];
  float fx[VPT], fy[VPT], fz[VPT];
  int lox = 1 << 30, hix = -(1 << 30), loy = lox, hiy = hix, loz = lox, hiz = hix;
  {
    int ixs[VPT], iys[VPT], izs[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const VoxTap t = make_vox_tap<DIM>(th, min(tx0 + lx, d.s2 - 1), min(ty0 + ly, d.s1 - 1), min(tz0 + lz, d.s0 - 1), d);
      ixs[k] = t.ix; iys[k] = t.iy; izs[k] = t.iz; fx[k] = t.fx; fy[k] = t.fy; fz[k] = t.fz;
      lox = min(lox, t.ix); hix = max(hix, t.ix);
      loy = min(loy, t.iy); hiy = max(hiy, t.iy);
      loz = min(loz, t.iz); hiz = max(hiz, t.iz);
      if (k & 1) __builtin_amdgcn_sched_barrier(0);
    }
    const BoxDesc bb = reduce_box<DIM, TZ>(lox, hix, loy, hiy, loz, hiz, red);
#pragma unroll
    for (int k = 0; k < VPT; ++k) cell[k] = ((izs[k] - bb.z0) * bb.ey + (iys[k] - bb.y0)) * bb.ex + (ixs[k] - bb.x0);
    lox = bb.x0; hix = bb.ex; loy = bb.y0; hiy = bb.ey; loz = bb.z0; hiz = bb.ez;
    if (!bb.fits) hix = 0;
  }
  BoxDesc b;
  b.x0 = lox; b.ex = hix; b.y0 = loy; b.ey = hiy; b.z0 = loz; b.ez = hiz; b.fits = hix != 0;
  const float* inn = in + (int64_t)n * C * V;
  const float* gon = gout + (int64_t)n * C * V;
  float acc[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) acc[q] = 0.f;
  float gm = 0.f;     // max |grad_out| over the tile: the fixed-point scale of k_affine_box_gin (tilemax); inf stays, NaN drops
  // d loss / d theta[r][c] += (d loss / d grid_r of this voxel) * base_c ; zeros padding: d(unnormalised coordinate) /
  // d(grid value) = (S - 1) / 2 everywhere
  auto add_theta = [&](int ox, int oy, int oz, float sx, float sy_, float sz_) {
    const float bx = affine_base_coord(ox, d.s2), by = affine_base_coord(oy, d.s1);
    const float bz = DIM == 3 ? affine_base_coord(oz, d.s0) : 1.f;
    const float ggx = (0.5f * (float)(d.s2 - 1)) * sx, ggy = (0.5f * (float)(d.s1 - 1)) * sy_;
    const float ggz = DIM == 3 ? (0.5f * (float)(d.s0 - 1)) * sz_ : 0.f;
    const float base[4] = {bx, by, bz, 1.f};
#pragma unroll
    for (int c = 0; c < DIM + 1; ++c) {
      acc[0 * (DIM + 1) + c] += ggx * base[c];
      acc[1 * (DIM + 1) + c] += ggy * base[c];
      if constexpr (DIM == 3) acc[2 * (DIM + 1) + c] += ggz * base[c];
    }
  };
  if (!b.fits) {      // block-uniform: direct gathers; no register array indexed by the run-time k
#pragma unroll 1
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
      if (!(ox < d.s2 && oy < d.s1 && oz < d.s0)) continue;
      const Taps<DIM, PAD_ZEROS> t = rebuild_taps<DIM>(th, ox, oy, oz, d);
      const int o = (oz * d.s1 + oy) * d.s2 + ox;
      float sx = 0.f, sy_ = 0.f, sz_ = 0.f;
      for (int c = 0; c < C; ++c) {
        const float g = gon[(int64_t)c * V + o];
        gm = fmaxf(gm, fabsf(g));
        sample_linear_bwd<DIM, PAD_ZEROS, false, true>(inn + (int64_t)c * V, nullptr, g, t, d, sx, sy_, sz_);
      }
      add_theta(ox, oy, oz, sx, sy_, sz_);
    }
  } else {
    float ax[VPT], ay[VPT], az[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) { ax[k] = 0.f; ay[k] = 0.f; az[k] = 0.f; }
    const int sy = b.ex, sz = b.ex * b.ey;
    constexpr int NP = DIM == 3 ? 2 : 1;     // (3D: two workgroups a CU whatever it holds; one pass: 75.6 against 72.2 us)
    BoxRegs<NP> pf;
    box_request<DIM, TZ, NP>(inn, b, d, pf);
    for (int c = 0; c < C; ++c) {
      // this channel's grad_out of the own voxels: requested before the box, in flight under its two staging round trips
      float gv[VPT];
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        int lx, ly, lz;
        local_voxel<DIM>(k, lx, ly, lz);
        const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
        gv[k] = gon[(int64_t)c * V + (min(oz, d.s0 - 1) * d.s1 + min(oy, d.s1 - 1)) * d.s2 + min(ox, d.s2 - 1)];
      }
      if (c > 0) __syncthreads();
      box_commit<DIM, TZ, NP>(inn + (int64_t)c * V, b, d, box, pf);
      __syncthreads();
      if (c + 1 < C) box_request<DIM, TZ, NP>(inn + (int64_t)(c + 1) * V, b, d, pf);   // in flight under this channel's taps
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        int lx, ly, lz;
        local_voxel<DIM>(k, lx, ly, lz);
        const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
        const bool ok = ox < d.s2 && oy < d.s1 && oz < d.s0;
        const float g = ok ? gv[k] : 0.f;
        gm = fmaxf(gm, fabsf(g));
        const float* p = box + cell[k];
        // d(sample)/d(x, y, z) in difference form: x-lerps a and x-differences dx of the 2^(d-1) corner pairs, then
        // lerps / differences along y and z (~35 VALU operations per voxel and channel; the corner-sum form of
        // sample_linear_bwd is ~100).  Invalid corners read the zero border.
        const float gx = 1.f - fx[k], gy = 1.f - fy[k], gz = 1.f - fz[k];
        float a[4], dx[4];
#pragma unroll
        for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
          for (int cy = 0; cy < 2; ++cy) {
            const float v0 = p[cz * sz + cy * sy], v1 = p[cz * sz + cy * sy + 1];
            a[cz * 2 + cy] = fmaf(v1, fx[k], v0 * gx);
            dx[cz * 2 + cy] = v1 - v0;
          }
        if (DIM == 3) {
          const float ddx = fmaf(fmaf(dx[3], fy[k], dx[2] * gy), fz[k], fmaf(dx[1], fy[k], dx[0] * gy) * gz);
          const float ddy = fmaf(a[3] - a[2], fz[k], (a[1] - a[0]) * gz);
          const float ddz = fmaf(a[3], fy[k], a[2] * gy) - fmaf(a[1], fy[k], a[0] * gy);
          ax[k] = fmaf(ddx, g, ax[k]); ay[k] = fmaf(ddy, g, ay[k]); az[k] = fmaf(ddz, g, az[k]);
        } else {
          ax[k] = fmaf(fmaf(dx[1], fy[k], dx[0] * gy), g, ax[k]);
          ay[k] = fmaf(a[1] - a[0], g, ay[k]);
        }
        if (k & 1) __builtin_amdgcn_sched_barrier(0);     // see k_affine_box_fwd
      }
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const int ox = tx0 + lx, oy = ty0 + ly, oz = tz0 + lz;
      if (ox < d.s2 && oy < d.s1 && oz < d.s0) add_theta(ox, oy, oz, ax[k], ay[k], az[k]);
    }
  }
  block_sum<NT>(acc, smem);
  if (threadIdx.x == 0) {
    float* dst = gtheta_partial + ((int64_t)n * gridDim.x + blockIdx.x) * NT;
#pragma unroll
    for (int q = 0; q < NT; ++q) dst[q] = acc[q];
  }
  if (tilemax) {      // (uniform) indexed by the TILE, not by the block: k_affine_box_gin looks tiles up by position
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0)
      tilemax[(int64_t)n * gridDim.x + xcd_contiguous(blockIdx.x, gridDim.x)] = fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
  }
}

// ---------------------------------------------------------------------------------------------
// grad_in as an OWNER-COMPUTES scatter (zeros padding, linear).  The lattice gather of sampler.hip enumerates, per input
// voxel u, the samples v with |p(v) - u| < 1 through slab tests: ~1500 VALU wave-instructions per 64 voxels at C = 4 and
// the kernel is VALU-bound (profiles/r03/affine/pmc_sq_box.csv: 160 us at 4x4x128x128x64).  The forward taps of a sample
// are far cheaper than that enumeration (a position, three floors): here a workgroup owns a TX x TY x TZ tile of u, walks
// the bounding box of the samples that can reach it (the image of the tile +-1 under the inverse map, from
// k_affine_geometry), builds each sample's taps with the forward's arithmetic and deposits the corners that fall in ITS
// tile into LDS accumulators -- 32-bit fixed point scaled by the max |grad_out| over its own box (LDS integer atomics run
// at LDS rate, float ones do not: DESIGN lesson 1).  No global atomics, no zero fill, integer adds commute: deterministic,
// and a sample's result does not depend on what else is in the batch.  A sample is visited by every tile whose box holds
// it (2.7-6x), which is still 2.5-4x fewer instructions than the enumeration.
//   geo[n] = { M (3x3, xyz order), t (3), Minv (3x3), ext (3) }; mode[n] != 0: the sample goes through the global-atomic
//   kernel (zero-fill here).
// ---------------------------------------------------------------------------------------------
constexpr int kGeoFloatsBox = kGeoFloats;

template <int DIM, int CMAX>
__global__ void __launch_bounds__(kBlock)
k_affine_box_gin(const float* __restrict__ gout, const float* __restrict__ theta, const float* __restrict__ geo,
                 const int* __restrict__ mode, float* __restrict__ gin, int C, Dims d, const float* __restrict__ tilemax) {
  constexpr int TZ = DIM == 3 ? 8 : 1;
  using G = BoxGeom<DIM, TZ>;
  constexpr int TILE = G::TX * G::TY * G::TZ;
  constexpr int VPT = TILE / kBlock;
  __shared__ int acc[CMAX * TILE];
  __shared__ float wmax[kBlock / 64];
  const int V = (int)d.voxels();
  const int n = blockIdx.y;
  int ux0, uy0, uz0;
  tile_origin<DIM, TZ>(d, ux0, uy0, uz0);
  float* ginn = gin + (int64_t)n * C * V;
  if (mode[n] != 0) {  // this sample goes through the atomic kernel: start from zero
#pragma unroll 1
    for (int k = 0; k < VPT; ++k) {
      int lx, ly, lz;
      local_voxel<DIM>(k, lx, ly, lz);
      const int ux = ux0 + lx, uy = uy0 + ly, uz = uz0 + lz;
      if (ux < d.s2 && uy < d.s1 && uz < d.s0)
        for (int c = 0; c < C; ++c) ginn[(int64_t)c * V + (uz * d.s1 + uy) * d.s2 + ux] = 0.f;
    }
    return;
  }
  const Theta<DIM> th = load_theta<DIM>(theta, n);
  const float* gn = geo + (int64_t)n * kGeoFloatsBox;
  // ---- the box of samples that can reach the tile: p(v) in (u0 - 1, u0 + T) on every axis, v = Minv (p - t)
  int blo[3], bhi[3];
  {
    const int T[3] = {G::TX, G::TY, G::TZ};
    const int U0[3] = {ux0, uy0, uz0};
    const int S[3] = {d.s2, d.s1, d.s0};
    float plo[3], phi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { plo[r] = (float)(U0[r] - 1) - gn[9 + r]; phi[r] = (float)(U0[r] + T[r]) - gn[9 + r]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float m = gn[12 + a * 3 + r];
        lo += m >= 0.f ? m * plo[r] : m * phi[r];
        hi += m >= 0.f ? m * phi[r] : m * plo[r];
      }
      // (positions carry ~1e-4 voxel of rounding at these sizes: 0.05 of slack; the taps below decide what is deposited)
      const bool fin = lo > -1.0e8f && hi < 1.0e8f;
      blo[a] = fin ? max(0, (int)floorf(lo - 0.05f)) : 0;
      bhi[a] = fin ? min(S[a] - 1, (int)ceilf(hi + 0.05f)) : -1;
      if (a >= DIM) { blo[a] = 0; bhi[a] = 0; }
    }
  }
  const int bex = max(bhi[0] - blo[0] + 1, 0), bey = max(bhi[1] - blo[1] + 1, 0), bez = max(bhi[2] - blo[2] + 1, 0);
  const int ncell = bex * bey * bez;
  const float inv_ex = 1.f / (float)max(bex, 1), inv_exy = 1.f / (float)max(bex * bey, 1);
  const float* gon = gout + (int64_t)n * C * V;
  auto cell_voxel = [&](int i, int& vx, int& vy, int& vz) {
    vz = DIM == 3 ? (int)(((float)i + 0.5f) * inv_exy) : 0;
    const int rem = i - vz * (bex * bey);
    vy = (int)(((float)rem + 0.5f) * inv_ex);
    vx = rem - vy * bex + blo[0];
    vy += blo[1];
    vz += blo[2];
  };
  // ---- pass 1: max |grad_out| over the box (the fixed-point scale), accumulators to zero
  for (int i = threadIdx.x; i < CMAX * TILE; i += kBlock) acc[i] = 0;
  float m = 0.f;
  if (tilemax) {      // (uniform) the maxima k_affine_box_gtheta left per output tile: the tiles that cover the box
    const int T[3] = {G::TX, G::TY, G::TZ};
    const int ntx = (d.s2 + G::TX - 1) / G::TX, nty = (d.s1 + G::TY - 1) / G::TY;
    const int t0x = blo[0] / T[0], t0y = blo[1] / T[1], t0z = blo[2] / T[2];
    const int cx = ncell > 0 ? bhi[0] / T[0] - t0x + 1 : 0, cy = bhi[1] / T[1] - t0y + 1, cz = bhi[2] / T[2] - t0z + 1;
    const float* tm = tilemax + (int64_t)n * gridDim.x;
    for (int i = threadIdx.x; i < cx * cy * cz; i += kBlock) {
      const int iz = i / (cx * cy), r = i - iz * (cx * cy), iy = r / cx, ix = r - iy * cx;
      m = fmaxf(m, tm[((t0z + iz) * nty + t0y + iy) * ntx + t0x + ix]);
    }
  }
  auto box_max = [&]() {
    float mm = 0.f;
    for (int i = threadIdx.x; i < ncell; i += kBlock) {
      int vx, vy, vz;
      cell_voxel(i, vx, vy, vz);
      const int v = (vz * d.s1 + vy) * d.s2 + vx;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) mm = fmaxf(mm, fabsf(gon[(int64_t)(c < C ? c : 0) * V + v]));
      // (a NaN gradient: fmaxf drops it here; it is re-detected below through the sum test)
    }
    return mm;
  };
  auto block_max = [&](float mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
    __syncthreads();                                   // (wmax may still be read from the previous call)
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mm;
    __syncthreads();
    return fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  };
  float gmax = block_max(tilemax ? m : box_max());
  // an inf among the covering tiles need not lie in the box: look (uniform; only then) -- a non-finite gradient poisons
  // the tiles it reaches, not their neighbours
  if (tilemax && !(gmax < 3.0e38f)) gmax = block_max(box_max());
  // a cell receives at most prod(2 ext + 1) corners of weight <= 1
  float cnt = 1.f;
#pragma unroll
  for (int a = 0; a < DIM; ++a) cnt *= floorf(2.f * gn[21 + a]) + 2.f;
  const bool finite = gmax < 3.0e38f;
  const float scale = (gmax > 0.f && finite) ? 1073741824.f / (cnt * gmax) : 0.f;
  const float inv = (gmax > 0.f && finite) ? (cnt * gmax) / 1073741824.f : 0.f;
  // ---- pass 2: deposits.  GU candidates of the box per thread and round, their grad_out requested together.  The loop is
  // VALU-bound (profiles/r04/sq_issue_summary.txt: 21 % of the wave-cycles at 4 waves a SIMD = the VALU busy 85 % of the
  // time): a position, three taps and the reach test per CANDIDATE, 2.3x as many as samples at 5 degrees.  Positions
  // cannot be stepped along a row: every tile must see the same taps for a sample, so each comes from scratch.
  bool bad = false;
  constexpr int GU = DIM == 3 ? 4 : 2;     // (2 .. 8 measured within 3 %: the loop is VALU-bound, see below)
  for (int i0 = threadIdx.x; i0 < ncell; i0 += kBlock * GU) {
    float go[GU][CMAX];
    int vxs[GU], vys[GU], vzs[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      cell_voxel(min(i0 + u * kBlock, ncell - 1), vxs[u], vys[u], vzs[u]);
      const int v = (vzs[u] * d.s1 + vys[u]) * d.s2 + vxs[u];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) go[u][c] = gon[(int64_t)(c < C ? c : 0) * V + v];
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      if (i0 + u * kBlock >= ncell) continue;
      float bx, by, bz, gx, gy, gz;
      affine_position_xyz<DIM>(th, vxs[u], vys[u], vzs[u], d, bx, by, bz, gx, gy, gz);
      const AxisTap tx = make_tap<PAD_ZEROS>(gx, d.s2), ty = make_tap<PAD_ZEROS>(gy, d.s1);
      AxisTap tz;
      if (DIM == 3) tz = make_tap<PAD_ZEROS>(gz, d.s0);
      else { tz.i0 = 0; tz.w0 = 1.f; tz.w1 = 0.f; }
      const int px = tx.i0 - ux0, py = ty.i0 - uy0, pz = DIM == 3 ? tz.i0 - uz0 : 0;
      const bool reach = px >= -1 && px < G::TX && py >= -1 && py < G::TY && (DIM == 2 || (pz >= -1 && pz < G::TZ));
      if (!reach) continue;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) bad = bad || (c < C && !(go[u][c] == go[u][c]));
      // 2^d corners x C deposits per reaching sample (a row-interval compaction of the box, which cut the visited cells
      // 2-3x, changed the time by < 10 % and was dropped).  Per-axis validity and the scaled x weights once per sample; a
      // corner is one product, a channel one product + convert + LDS add.
      const float wsx[2] = {tx.w0 * scale, tx.w1 * scale};
      const bool okx[2] = {(unsigned)px < (unsigned)G::TX, (unsigned)(px + 1) < (unsigned)G::TX};
      const bool oky[2] = {(unsigned)py < (unsigned)G::TY, (unsigned)(py + 1) < (unsigned)G::TY};
      const bool okz[2] = {(unsigned)pz < (unsigned)G::TZ, DIM == 3 && (unsigned)(pz + 1) < (unsigned)G::TZ};
      int* cell0 = acc + (pz * G::TY + py) * G::TX + px;
#pragma unroll
      for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          const float wyz = DIM == 3 ? (cy ? ty.w1 : ty.w0) * (cz ? tz.w1 : tz.w0) : (cy ? ty.w1 : ty.w0);
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            if (!(okx[cx] && oky[cy] && okz[cz])) continue;
            const float ws = wsx[cx] * wyz;
            int* cell = cell0 + (cz * G::TY + cy) * G::TX + cx;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
              if (c < C) atomicAdd(cell + c * TILE, fix_round(ws * go[u][c]));
          }
        }
    }
  }
  // non-finite gradients must not come out as finite numbers: the whole tile turns NaN
  const bool poison = __syncthreads_or((int)(bad || !finite)) != 0;
  const float nanv = __int_as_float(0x7fc00000);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    int lx, ly, lz;
    local_voxel<DIM>(k, lx, ly, lz);
    const int ux = ux0 + lx, uy = uy0 + ly, uz = uz0 + lz;
    if (!(ux < d.s2 && uy < d.s1 && uz < d.s0)) continue;
    const int cellu = (lz * G::TY + ly) * G::TX + lx;
    const int u = (uz * d.s1 + uy) * d.s2 + ux;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) ginn[(int64_t)c * V + u] = poison ? nanv : (float)acc[c * TILE + cellu] * inv;
  }
}

}  // namespace advchain

using namespace advchain;

static const bool g_no_affine_box = getenv("ADVCHAIN_NO_AFFINE_BOX") != nullptr;   // A/B knob: direct-gather kernels

static inline bool box_shape_ok(const Dims& d, const void* a, const void* b, const void* c) {
  if (g_no_affine_box) return false;
  if (d.s2 % 4 != 0 || d.s2 < 4) return false;
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// Returns true when the box kernel took the launch (linear, zeros padding).
bool advchain_affine_box_fwd_launch(const float* in, const float* theta, float* out, int64_t N, int64_t C, int ndim, Dims d,
                                    hipStream_t st, const float* ride_in, float* ride_out, int ride_nonzero) {
  if (!box_shape_ok(d, in, out, ride_in) || (reinterpret_cast<uintptr_t>(ride_out) & 15) != 0) return false;
  dim3 b(kBlock);
  if (ndim == 3) hipLaunchKernelGGL((k_affine_box_fwd<3, 8>), dim3(box_tiles<3, 8>(d), (unsigned)N), b, 0, st, in, theta, out, (int)C, d, ride_in, ride_out, ride_nonzero);
  else hipLaunchKernelGGL((k_affine_box_fwd<2, 1>), dim3(box_tiles<2, 1>(d), (unsigned)N), b, 0, st, in, theta, out, (int)C, d, ride_in, ride_out, ride_nonzero);
  return true;
}

int advchain_affine_box_tiles(int ndim, Dims d) { return ndim == 3 ? box_tiles<3, 8>(d) : box_tiles<2, 1>(d); }

// Block partial sums of grad_theta -> gpart[(n * nblocks + block) * ndim * (ndim + 1)]; returns the number of blocks per
// sample, 0 when the shape is not taken.  tilemax (optional, N x nblocks floats): max |grad_out| of every output tile, by
// tile position -- the fixed-point scale of advchain_affine_box_gin_launch when it runs AFTER this launch.
int advchain_affine_box_gtheta_launch(const float* gout, const float* in, const float* theta, float* gpart, int64_t N,
                                      int64_t C, int ndim, Dims d, int max_blocks, hipStream_t st, float* tilemax, float* geo,
                                      int* mode) {
  if (!box_shape_ok(d, in, gout, nullptr)) return 0;
  const int nb = ndim == 3 ? box_tiles<3, 8>(d) : box_tiles<2, 1>(d);     // (the tiles of k_affine_box_gin: tilemax)
  if (nb > max_blocks) return 0;
  dim3 b(kBlock), g(nb, (unsigned)N);
  if (ndim == 3) hipLaunchKernelGGL((k_affine_box_gtheta<3, 8>), g, b, 0, st, gout, in, theta, gpart, (int)C, d, tilemax, geo, mode);
  else hipLaunchKernelGGL((k_affine_box_gtheta<2, 1>), g, b, 0, st, gout, in, theta, gpart, (int)C, d, tilemax, geo, mode);
  return nb;
}


// grad_in through the owner-computes LDS scatter (after k_affine_geometry filled geo / mode); false = shape not taken.
// tilemax: the per-tile maxima of advchain_affine_box_gtheta_launch on the same grad_out, or nullptr (the kernel then reads
// its box of grad_out once more for the maximum: 15 of 112 us at 4 x 4 x 128 x 128 x 64).
bool advchain_affine_box_gin_launch(const float* gout, const float* theta, const float* geo, const int* mode, float* gin,
                                    int64_t N, int64_t C, int ndim, Dims d, hipStream_t st, const float* tilemax) {
  static const bool no_gin = getenv("ADVCHAIN_NO_AFFINE_BOX_GIN") != nullptr;   // A/B knob: the lattice gather of sampler.hip
  if (no_gin || g_no_affine_box || C > 4) return false;
  dim3 b(kBlock);
  if (ndim == 3) {
    dim3 g(box_tiles<3, 8>(d), (unsigned)N);
    if (C <= 1) hipLaunchKernelGGL((k_affine_box_gin<3, 1>), g, b, 0, st, gout, theta, geo, mode, gin, (int)C, d, tilemax);
    else hipLaunchKernelGGL((k_affine_box_gin<3, 4>), g, b, 0, st, gout, theta, geo, mode, gin, (int)C, d, tilemax);
  } else {
    dim3 g(box_tiles<2, 1>(d), (unsigned)N);
    if (C <= 1) hipLaunchKernelGGL((k_affine_box_gin<2, 1>), g, b, 0, st, gout, theta, geo, mode, gin, (int)C, d, tilemax);
    else hipLaunchKernelGGL((k_affine_box_gin<2, 4>), g, b, 0, st, gout, theta, geo, mode, gin, (int)C, d, tilemax);
  }
  return true;
}
