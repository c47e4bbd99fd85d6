// Field-construction kernels for gfx950: separable Gaussian, banded tensor-product interpolation
// (velocity upsample / B-spline bias field) and its adjoint, bias-field apply, per-sample
// normalised updates, streaming elementwise ops.  All HBM-bound; no MFMA.
//
//   advchain_gauss_axis            <- depthwise 9^d Gaussian conv, adv_morph.py:377-452 (separable, Q3)
//   advchain_tp_interp_fwd         <- F.interpolate(v, size=full, linear, align_corners=False) adv_morph.py:464
//                                     (+ 'basegrid += dxy' adv_morph.py:111, + ||.|| of adv_morph.py:160)
//   advchain_band_reduce_axis      <- autograd adjoint of the above / of conv_transpose+crop+Upsample
//   advchain_bias_field_{fwd,bwd}  <- conv_transpose{2,3}d + crop + Upsample + exp + clip + multiply,
//                                     adv_bias.py:279-356,186 (closed form of SURVEY Appendix E)
//   advchain_axpy / advchain_scale <- adv_noise.py:81-84 and gradient scaling
//   advchain_sumsq_partial / advchain_norm_axpy <- unit_normalize + ascent update,
//                                     adv_transformation_base.py:151-155, adv_noise.py:56-63 etc.
#include "sampler_common.h"

namespace advchain {

constexpr int kBandMax = 8;

struct BandAxis {
  const int* start;  // [S]   first coefficient index touched by output o
  const float* w;    // [S*B] weights
  const int* lo;     // [g]   adjoint: outputs [lo[k], hi[k]) may touch coefficient k
  const int* hi;     // [g]
  int S, g, B;
};
struct BandTables { BandAxis a[3]; };

__device__ __forceinline__ void decode3(int v, const Dims& d, int& i0, int& i1, int& i2) {
  i2 = v % d.s2;
  const int r = v / d.s2;
  i1 = r % d.s1;
  i0 = r / d.s1;
}

// sum_{a,b,c} coef[s0+a, s1+b, s2+c] * w0[a] * w1[b] * w2[c]
__device__ __forceinline__ float tp_eval(const float* __restrict__ coef, const BandTables& T, int i0, int i1, int i2) {
  const BandAxis& A0 = T.a[0];
  const BandAxis& A1 = T.a[1];
  const BandAxis& A2 = T.a[2];
  const int s0 = A0.start[i0], s1 = A1.start[i1], s2 = A2.start[i2];
  const float* w0 = A0.w + i0 * A0.B;
  const float* w1 = A1.w + i1 * A1.B;
  const float* w2 = A2.w + i2 * A2.B;
  float acc = 0.f;
  for (int a = 0; a < A0.B; ++a) {
    float acc1 = 0.f;
    for (int b = 0; b < A1.B; ++b) {
      const float* row = coef + ((int64_t)(s0 + a) * A1.g + (s1 + b)) * A2.g + s2;
      float acc2 = 0.f;
      for (int c = 0; c < A2.B; ++c) acc2 += row[c] * w2[c];
      acc1 += acc2 * w1[b];
    }
    acc += acc1 * w0[a];
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// Row-blended evaluation of the banded tensor product.  A workgroup owns (plane, i0, a chunk of i1 rows): for
// each group of ROWS rows it first blends the B0 x B1 coefficient rows that the group needs into LDS
// (blend[r][k] = sum_ab w0[a] w1[b] coef[s0+a][s1+b][k], g2 values per row -- the only work that touches the
// coefficient grid), then every thread produces its voxel from B2 LDS reads.  Per voxel: B2 FMAs + the epilogue,
// instead of B0*B1*B2 global gathers + three table look-ups (the per-voxel form ran at 140-590 GB/s).
// ---------------------------------------------------------------------------------------------
constexpr int kTpChunk = 32;     // i1 rows per workgroup
constexpr int kTpMaxLds = 2048;  // floats: ROWS * g2

struct TpGeom { int XT, ROWS; };  // threads along x, rows per pass (XT * ROWS == kBlock)

__host__ __device__ inline TpGeom tp_geom(int S2) {
  TpGeom g;
  (void)S2;
  g.XT = 64;   // one wave per row, looping over x: the blend phase is amortised over 4 rows per barrier
  g.ROWS = kBlock / g.XT;
  return g;
}

// VEC = 4 (S2 % 4 == 0, 16-byte aligned planes): a thread produces 4 consecutive x and hands them to the epilogue
// together, which then moves 16 bytes per lane (these epilogues are pure streaming: their time is their number of
// vector-memory instructions).  epi(i1, x, val) with val = float[VEC].
template <int VEC, class Epi>
__device__ __forceinline__ void tp_rows(const float* __restrict__ coef, const BandTables& T, const Dims& full, int i0,
                                        int i1_begin, int i1_end, float* lds, Epi&& epi) {
  const BandAxis& A0 = T.a[0];
  const BandAxis& A1 = T.a[1];
  const BandAxis& A2 = T.a[2];
  const int ncol = full.s2 / VEC;                          // column groups per row
  int XT = 64;                                             // threads along x: one wave per row, or fewer for short rows
  if (VEC > 1) { XT = 1; while (XT < ncol && XT < 64) XT *= 2; }
  const int ROWS = kBlock / XT;
  const int tx = threadIdx.x % XT, ry = threadIdx.x / XT;
  const int g2 = A2.g;
  // Every table value this workgroup needs is requested up front, together (round 4): the row tables of its i1 chunk go to
  // LDS, the plane's and the thread's first column group's stay in registers.  Before, the blend phase loaded start[i1] and
  // then the coefficients it addresses, and the output phase start[x] / w[x] behind the barrier -- four dependent L2 round
  // trips per 8 KB written (56 us for a 100 MB write at 8 x 3 x 128 x 128 x 64); now one for the tables, one for the
  // coefficients.  Same values, same sums.
  __shared__ int s1_s[kTpChunk];
  __shared__ float w1_s[kTpChunk * kBandMax];
  const int nrow = i1_end - i1_begin;                      // <= kTpChunk
  for (int e = threadIdx.x; e < nrow * (A1.B + 1); e += kBlock) {
    const int r = e / (A1.B + 1), b = e - r * (A1.B + 1);
    if (b == 0) s1_s[r] = A1.start[i1_begin + r];
    else w1_s[r * kBandMax + b - 1] = A1.w[(i1_begin + r) * A1.B + b - 1];
  }
  const int s0 = A0.start[i0];
  const float* w0 = A0.w + i0 * A0.B;
  int s2f[VEC];
  float w2f[VEC][kBandMax];
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int col = min(tx * VEC + q, full.s2 - 1);
    s2f[q] = A2.start[col];
#pragma unroll
    for (int c = 0; c < kBandMax; ++c) w2f[q][c] = c < A2.B ? A2.w[col * A2.B + c] : 0.f;
  }
  __syncthreads();
  // rows blended per phase: as many as the LDS buffer holds (with 4 rows per phase a workgroup paid two barriers and a
  // dependent start[] -> coef[] load chain per 4 rows: 52 us for a 50 MB write at 4x3x128x128x64)
  const int RPP = max(1, min(kTpChunk, kTpMaxLds / max(g2, 1)));
  for (int base = i1_begin; base < i1_end; base += RPP) {
    for (int e = threadIdx.x; e < RPP * g2; e += kBlock) {
      const int r = e / g2, k = e - r * g2;
      const int i1 = base + r;
      float acc = 0.f;
      if (i1 < i1_end) {
        const int s1 = s1_s[i1 - i1_begin];
        const float* w1 = w1_s + (i1 - i1_begin) * kBandMax;
        if (A0.B == 2 && A1.B == 2) {   // linear upsampling (uniform): the four terms' loads in flight together
          const float* cb = coef + ((int64_t)s0 * A1.g + s1) * g2 + k;
          const float c00 = cb[0], c01 = cb[g2], c10 = cb[(int64_t)A1.g * g2], c11 = cb[(int64_t)A1.g * g2 + g2];
          const float u0 = w1[0], u1 = w1[1], v0 = w0[0], v1 = w0[1];
          acc += v0 * (0.f + u0 * c00 + u1 * c01);
          acc += v1 * (0.f + u0 * c10 + u1 * c11);
        } else {
          for (int a = 0; a < A0.B; ++a) {
            float acc1 = 0.f;
            for (int b = 0; b < A1.B; ++b) acc1 += w1[b] * coef[((int64_t)(s0 + a) * A1.g + (s1 + b)) * g2 + k];
            acc += w0[a] * acc1;
          }
        }
      }
      lds[e] = acc;
    }
    __syncthreads();
    // x outer: the bands of the thread's output columns (start, <= 8 weights each) are loaded once and kept in
    // registers for all rows (per-row table loads were a dependent global-load chain in the inner loop)
    for (int xg = tx; xg < ncol; xg += XT) {
      int s2[VEC];
      float w2[VEC][kBandMax];
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        if (xg == tx) {           // (requested before the first barrier)
          s2[q] = s2f[q];
#pragma unroll
          for (int c = 0; c < kBandMax; ++c) w2[q][c] = w2f[q][c];
        } else {
          s2[q] = A2.start[xg * VEC + q];
#pragma unroll
          for (int c = 0; c < kBandMax; ++c) w2[q][c] = c < A2.B ? A2.w[(xg * VEC + q) * A2.B + c] : 0.f;
        }
      }
      for (int r = ry; r < RPP; r += ROWS) {
        const int i1 = base + r;
        if (i1 >= i1_end) break;
        float val[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          float v = 0.f;
#pragma unroll
          for (int c = 0; c < kBandMax; ++c)
            if (c < A2.B) v += w2[q][c] * lds[r * g2 + s2[q] + c];
          val[q] = v;
        }
        epi(i1, xg * VEC, val);
      }
    }
    __syncthreads();
  }
}

// out[plane][v] = (add_identity ? identity_coord(channel) : 0) + scale * interp ; optional sum of interp^2
// (64 slot accumulators) for the 3D step-count rule.  grid = (i1 chunks, S0, planes).
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_tp_interp_fwd(const float* __restrict__ coef, float* __restrict__ out, BandTables T, Dims full, int C,
                int add_identity, float scale, float* __restrict__ sumsq, float* __restrict__ disp_out) {
  __shared__ float lds[kTpMaxLds];
  __shared__ float smem[4];
  const int plane = blockIdx.z, i0 = blockIdx.y;
  const int i1b = blockIdx.x * kTpChunk, i1e = min(i1b + kTpChunk, full.s1);
  const int64_t G = (int64_t)T.a[0].g * T.a[1].g * T.a[2].g;
  const int V = (int)full.voxels();
  const int c = plane % C;  // channel 0 = x <-> s2, 1 = y <-> s1, 2 = z <-> s0
  float sq[1] = {0.f};
  float dmax = 0.f;
  const float to_vox = fabsf(scale) * 0.5f * (float)((c == 0 ? full.s2 : (c == 1 ? full.s1 : full.s0)) - 1);
  tp_rows<VEC>(coef + (int64_t)plane * G, T, full, i0, i1b, i1e, lds, [&](int i1, int x, const float (&val)[VEC]) {
    float o[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      sq[0] += val[q] * val[q];
      dmax = fmaxf(dmax, fabsf(val[q]));
      float base = 0.f;
      if (add_identity) base = c == 0 ? lin_coord(x + q, full.s2) : (c == 1 ? lin_coord(i1, full.s1) : lin_coord(i0, full.s0));
      o[q] = base + scale * val[q];
    }
    if (out) store_vec<VEC>(out + (int64_t)plane * V + ((int64_t)i0 * full.s1 + i1) * full.s2 + x, o);
  });
  if (disp_out) wave_max_to_slots(fminf(dmax * to_vox, 1.0e9f), disp_out);   // displacement of base + scale * val, in voxels
  if (sumsq) {
    block_sum<1>(sq, smem);
    if (threadIdx.x == 0) atomic_add_f32(sumsq + (blockIdx.x + blockIdx.y * 3u + blockIdx.z * 7u) % kSumSlots, sq[0]);
  }
}

// The same with the smoothing of the low-resolution plane in front (round 6: launch consolidation): the batch [v; -v] of a
// paired 2D field -- workgroup plane b smooths velocity plane b % P scaled by +-gscale in LDS (the arithmetic of k_gauss_small,
// mirror 1: same taps, same order, zero padding), the workgroup of the first row chunk writes it to s1, and the rows are
// blended from that LDS plane instead of a global one.  One launch instead of two per DemonsCompose pair (the small-plane
// smoothing was 6 us of kernel behind a launch boundary, eleven times per cfg-2 call); every workgroup of a plane repeats the
// 2 x 9-tap smoothing of its <= 1024 values, which is nothing next to its 32 rows of output.
constexpr int kGsFuseMax = 1024;
struct GaussW9 { float w[9]; };
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_tp_interp_fwd_gs(const float* __restrict__ vel, float* __restrict__ s1, float* __restrict__ out, BandTables T, Dims full, int C,
                   int add_identity, float scale, float* __restrict__ disp_out, GaussW9 gw, float gscale, int P) {
  __shared__ float lds[kTpMaxLds];
  __shared__ float gs[2][kGsFuseMax];
  const int plane = blockIdx.z, i0 = blockIdx.y;
  const int i1b = blockIdx.x * kTpChunk, i1e = min(i1b + kTpChunk, full.s1);
  const int g1 = T.a[1].g, g2 = T.a[2].g, Vg = g1 * g2;        // (2D: a trivial leading axis)
  const int V = (int)full.voxels();
  const int c = plane % C;
  {
    const float* src = vel + (int64_t)(plane % P) * Vg;
    const float sc = plane >= P ? -gscale : gscale;
    for (int i = threadIdx.x; i < Vg; i += kBlock) gs[0][i] = src[i] * sc;
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int Sa = pass == 0 ? g2 : g1, st = pass == 0 ? 1 : g2;      // innermost axis first, like k_gauss_small
      for (int i = threadIdx.x; i < Vg; i += kBlock) {
        const int ia = (i / st) % Sa;
        float acc = 0.f;
#pragma unroll
        for (int k = -4; k <= 4; ++k) {
          const int j = ia + k;
          if (j >= 0 && j < Sa) acc += gw.w[k + 4] * gs[cur][i + k * st];
        }
        gs[cur ^ 1][i] = acc;
      }
      __syncthreads();
      cur ^= 1;
    }
    if (s1 && blockIdx.x == 0 && blockIdx.y == 0)
      for (int i = threadIdx.x; i < Vg; i += kBlock) s1[(int64_t)plane * Vg + i] = gs[0][i];      // (two passes: back in buffer 0)
  }
  float dmax = 0.f;
  const float to_vox = fabsf(scale) * 0.5f * (float)((c == 0 ? full.s2 : (c == 1 ? full.s1 : full.s0)) - 1);
  tp_rows<VEC>(gs[0], T, full, i0, i1b, i1e, lds, [&](int i1, int x, const float (&val)[VEC]) {
    float o[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      dmax = fmaxf(dmax, fabsf(val[q]));
      float base = 0.f;
      if (add_identity) base = c == 0 ? lin_coord(x + q, full.s2) : (c == 1 ? lin_coord(i1, full.s1) : lin_coord(i0, full.s0));
      o[q] = base + scale * val[q];
    }
    store_vec<VEC>(out + (int64_t)plane * V + ((int64_t)i0 * full.s1 + i1) * full.s2 + x, o);
  });
  if (disp_out) wave_max_to_slots(fminf(dmax * to_vox, 1.0e9f), disp_out);
}

// The 3D linear upsampling (bands of 2 on every axis), ZB consecutive i0 planes per workgroup, software-pipelined (round 4).
// One plane per workgroup meant: tables, coefficients, barrier, 10 outputs per thread, barrier -- 19200 workgroups of ~4 us for
// a 196 MB write at cfg-5 (1.9 TB/s), and hoisting the table loads alone changed nothing (104 us either way: the chain is per
// WORKGROUP, and a workgroup had too little to do behind it).  Here the tables of the i1 chunk, the columns and the ZB planes
// are requested once; the four coefficients of every blend element of plane i0 + 1 are requested BEFORE the outputs of plane
// i0 are formed and blended after them, into the other half of a double LDS buffer: one barrier per plane, the coefficient
// round trip under the stores.  Same sums in the same order as tp_rows: bit-identical.
constexpr int kTpZB = 4;
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_tp_interp_fwd_planes(const float* __restrict__ coef, float* __restrict__ out, BandTables T, Dims full, int C, int add_identity,
                       float scale, float* __restrict__ sumsq, float* __restrict__ disp_out) {
  __shared__ float lds[2][kTpMaxLds];
  __shared__ int s1_s[kTpChunk], s0_s[kTpZB];
  __shared__ float w1_s[kTpChunk * 2], w0_s[kTpZB * 2];
  __shared__ float smem[4];
  const BandAxis& A0 = T.a[0];
  const BandAxis& A1 = T.a[1];
  const BandAxis& A2 = T.a[2];
  const int plane = blockIdx.z;
  const int z0 = blockIdx.y * kTpZB, z1 = min(z0 + kTpZB, full.s0);
  const int i1b = blockIdx.x * kTpChunk, i1e = min(i1b + kTpChunk, full.s1);
  const int64_t G = (int64_t)A0.g * A1.g * A2.g;
  const int V = (int)full.voxels();
  const int c = plane % C;
  const float* cf = coef + (int64_t)plane * G;
  const int g2 = A2.g, nrow = i1e - i1b, NE = nrow * g2;          // NE <= kTpMaxLds (launcher)
  constexpr int EPT = kTpMaxLds / kBlock;                          // blend elements per thread at most
  const int ncol = full.s2 / VEC;
  int XT = 64;
  if (VEC > 1) { XT = 1; while (XT < ncol && XT < 64) XT *= 2; }
  const int ROWS = kBlock / XT;
  const int tx = threadIdx.x % XT, ry = threadIdx.x / XT;
  // ---- every table value, requested together
  for (int e = threadIdx.x; e < nrow * 3; e += kBlock) {
    const int r = e / 3, b = e - r * 3;
    if (b == 0) s1_s[r] = A1.start[i1b + r];
    else w1_s[r * 2 + b - 1] = A1.w[(i1b + r) * 2 + b - 1];
  }
  if (threadIdx.x < (z1 - z0) * 3) {
    const int r = threadIdx.x / 3, b = threadIdx.x - r * 3;
    if (b == 0) s0_s[r] = A0.start[z0 + r];
    else w0_s[r * 2 + b - 1] = A0.w[(z0 + r) * 2 + b - 1];
  }
  int s2f[VEC];
  float w2f[VEC][2];
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int col = min(tx * VEC + q, full.s2 - 1);
    s2f[q] = A2.start[col];
    w2f[q][0] = A2.w[col * 2];
    w2f[q][1] = A2.w[col * 2 + 1];
  }
  __syncthreads();
  float cc[EPT][4];
  auto request = [&](int zi) {          // the four coefficients of each blend element of plane z0 + zi
    const int s0 = s0_s[zi];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      if (j * kBlock >= NE) break;                                    // block-uniform
      const int e = min(threadIdx.x + j * kBlock, NE - 1);           // (unconditional loads from a clamped element)
      const int r = e / g2, k = e - r * g2;
      const float* cb = cf + ((int64_t)s0 * A1.g + s1_s[r]) * g2 + k;
      cc[j][0] = cb[0]; cc[j][1] = cb[g2]; cc[j][2] = cb[(int64_t)A1.g * g2]; cc[j][3] = cb[(int64_t)A1.g * g2 + g2];
    }
  };
  auto blend = [&](int zi, float* dst) {
    const float v0 = w0_s[zi * 2], v1 = w0_s[zi * 2 + 1];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = threadIdx.x + j * kBlock;
      if (e >= NE) continue;
      const int r = e / g2;
      const float u0 = w1_s[r * 2], u1 = w1_s[r * 2 + 1];
      float acc = 0.f;
      acc += v0 * (0.f + u0 * cc[j][0] + u1 * cc[j][1]);
      acc += v1 * (0.f + u0 * cc[j][2] + u1 * cc[j][3]);
      dst[e] = acc;
    }
  };
  float sq[1] = {0.f};
  float dmax = 0.f;
  const bool track = sumsq != nullptr || disp_out != nullptr;
  const float to_vox = fabsf(scale) * 0.5f * (float)((c == 0 ? full.s2 : (c == 1 ? full.s1 : full.s0)) - 1);
  request(0);
  blend(0, lds[0]);
  __syncthreads();
  for (int zi = 0; zi < z1 - z0; ++zi) {
    const int i0 = z0 + zi;
    const float* cur = lds[zi & 1];
    const bool more = zi + 1 < z1 - z0;
    if (more) request(zi + 1);
    for (int xg = tx; xg < ncol; xg += XT) {
      int s2[VEC];
      float w2[VEC][2];
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        if (xg == tx) { s2[q] = s2f[q]; w2[q][0] = w2f[q][0]; w2[q][1] = w2f[q][1]; }
        else { s2[q] = A2.start[xg * VEC + q]; w2[q][0] = A2.w[(xg * VEC + q) * 2]; w2[q][1] = A2.w[(xg * VEC + q) * 2 + 1]; }
      }
      // the identity part of an output is a column constant (channel 0), a row constant (1) or a plane constant (2)
      float basex[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) basex[q] = (add_identity && c == 0) ? lin_coord(xg * VEC + q, full.s2) : 0.f;
      const float basez = (add_identity && c == 2) ? lin_coord(i0, full.s0) : 0.f;
      float* orow = out ? out + (int64_t)plane * V + ((int64_t)i0 * full.s1 + i1b) * full.s2 + xg * VEC : nullptr;
      for (int r = ry; r < nrow; r += ROWS) {
        const float basey = (add_identity && c == 1) ? lin_coord(i1b + r, full.s1) : basez;
        const float* lr = cur + r * g2;
        float o[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          float v = 0.f;
          v += w2[q][0] * lr[s2[q]];
          v += w2[q][1] * lr[s2[q] + 1];
          if (track) {
            sq[0] += v * v;
            dmax = fmaxf(dmax, fabsf(v));
          }
          o[q] = (c == 0 ? basex[q] : basey) + scale * v;
        }
        if (out) store_vec<VEC>(orow + r * full.s2, o);
      }
    }
    if (more) blend(zi + 1, lds[(zi + 1) & 1]);
    __syncthreads();
  }
  if (disp_out) wave_max_to_slots(fminf(dmax * to_vox, 1.0e9f), disp_out);
  if (sumsq) {
    block_sum<1>(sq, smem);
    if (threadIdx.x == 0) atomic_add_f32(sumsq + (blockIdx.x + blockIdx.y * 3u + blockIdx.z * 7u) % kSumSlots, sq[0]);
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint along one axis:  out[o][k][i] = sum_s val(in[o][s][i]) * w[s][k - start[s]]
//   val = (in - in2) * scale  when in2 != null, else in * scale
// ---------------------------------------------------------------------------------------------
constexpr int kBrMaxW = 4096;        // densified band table (g * WB floats)

// Densify the band of every coefficient into LDS: Wd[k][j] = weight of input lo[k] + j for coefficient k, so the
// reduction loops touch no global table (a dependent global load per term made them latency-bound).
// Returns the band width WB = max_k (hi[k] - lo[k]), or 0 when the table does not fit (g > 64 or g * WB > kBrMaxW).
__device__ __forceinline__ int stage_band(const BandAxis& A, float* Wd, int* lo_s) {
  const int lane = threadIdx.x & 63;
  if (A.g > 64) return 0;
  int WB = lane < A.g ? A.hi[lane] - A.lo[lane] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) WB = max(WB, __shfl_xor(WB, o, 64));
  if (A.g * WB > kBrMaxW) return 0;
  for (int i = threadIdx.x; i < A.g * WB; i += blockDim.x) {
    const int k = i / WB, j = i - k * WB;
    const int s2 = A.lo[k] + j;
    float w = 0.f;
    if (s2 < A.hi[k]) {
      const int b = k - A.start[s2];
      if (b >= 0 && b < A.B) w = A.w[s2 * A.B + b];
    }
    Wd[i] = w;
  }
  for (int i = threadIdx.x; i < A.g; i += blockDim.x) lo_s[i] = A.lo[i];
  __syncthreads();
  return WB;
}

__global__ void __launch_bounds__(kBlock)
k_band_reduce_axis(const float* __restrict__ in, const float* __restrict__ in2, float* __restrict__ out, int64_t outer,
                   int inner, BandAxis A, float scale) {
  __shared__ float Wd[kBrMaxW];
  __shared__ int lo_s[64];
  const int WB = stage_band(A, Wd, lo_s);
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t total = outer * A.g * inner;
  if (t >= total) return;
  const int i = (int)(t % inner);
  const int64_t r = t / inner;
  const int k = (int)(r % A.g);
  const int64_t o = r / A.g;
  const float* p = in + (o * A.S) * inner + i;
  const float* p2 = in2 ? in2 + (o * A.S) * inner + i : nullptr;
  float acc = 0.f;
  if (WB > 0) {
    const int lo = lo_s[k];
    const int len = min(WB, A.S - lo);
    const float* w = Wd + k * WB;
    for (int j = 0; j < len; ++j) {
      float x = p[(int64_t)(lo + j) * inner];
      if (p2) x -= p2[(int64_t)(lo + j) * inner];
      acc = fmaf(x, w[j], acc);
    }
  } else {
    const int lo = A.lo[k], hi = A.hi[k];
    for (int s = lo; s < hi; ++s) {
      const int b = k - A.start[s];
      if (b >= 0 && b < A.B) {
        float x = p[(int64_t)s * inner];
        if (p2) x -= p2[(int64_t)s * inner];
        acc += x * A.w[s * A.B + b];
      }
    }
  }
  out[t] = acc * scale;
}

// Outer-axis variant through an LDS slab (inner % 4 == 0): a workgroup takes one `o` and a chunk of IC inner columns,
// requests its (S x IC) slab with 16-byte loads BEFORE it densifies the band table (the table walk is three dependent
// global loads; the per-thread kernel above then walks its band with one dependent strided load per tap -- 16 us for a
// 2 MB input at cfg-2, 128 workgroups).  Same sums in the same order as k_band_reduce_axis.
constexpr int kBrSlabFloats = 12288;      // 48 KiB
__global__ void __launch_bounds__(kBlock)
k_band_reduce_axis_slab(const float* __restrict__ in, const float* __restrict__ in2, float* __restrict__ out, int inner, int IC,
                        BandAxis A, float scale) {
  extern __shared__ __attribute__((aligned(16))) float slab[];      // [S][IC + 1]
  __shared__ float Wd[kBrMaxW];
  __shared__ int lo_s[64];
  const int64_t o = blockIdx.x;
  const int c0 = blockIdx.y * IC, cols = min(IC, inner - c0);
  const int q = cols >> 2, n4 = A.S * q, pitch = IC + 1;
  const float* p = in + o * (int64_t)A.S * inner + c0;
  const float* p2 = in2 ? in2 + o * (int64_t)A.S * inner + c0 : nullptr;
  constexpr int U = 4;
  float4 v[U];
  auto fetch = [&](int idx) -> float4 {
    const int e = min(idx, n4 - 1);
    const int sr = e / q, x4 = e - sr * q;
    float4 a = *reinterpret_cast<const float4*>(p + (int64_t)sr * inner + 4 * x4);
    if (p2) {
      const float4 b = *reinterpret_cast<const float4*>(p2 + (int64_t)sr * inner + 4 * x4);
      a.x -= b.x; a.y -= b.y; a.z -= b.z; a.w -= b.w;
    }
    return a;
  };
  auto put = [&](int idx, const float4& a) {
    if (idx >= n4) return;
    const int sr = idx / q, x4 = idx - sr * q;
    float* dst = slab + sr * pitch + 4 * x4;
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
  };
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = fetch(threadIdx.x + u * kBlock);
  const int WB = stage_band(A, Wd, lo_s);       // (ends with a barrier)
#pragma unroll
  for (int u = 0; u < U; ++u) put(threadIdx.x + u * kBlock, v[u]);
  for (int i0 = threadIdx.x + U * kBlock; i0 < n4; i0 += U * kBlock) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = fetch(i0 + u * kBlock);
#pragma unroll
    for (int u = 0; u < U; ++u) put(i0 + u * kBlock, v[u]);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < A.g * cols; t += kBlock) {
    const int k = t / cols, i = t - k * cols;
    float acc = 0.f;
    if (WB > 0) {
      const int lo = lo_s[k];
      const int len = min(WB, A.S - lo);
      const float* w = Wd + k * WB;
      const float* x = slab + lo * pitch + i;
      for (int j = 0; j < len; ++j) acc = fmaf(x[j * pitch], w[j], acc);
    } else {
      const int lo = A.lo[k], hi = A.hi[k];
      for (int sr = lo; sr < hi; ++sr) {
        const int b = k - A.start[sr];
        if (b >= 0 && b < A.B) acc += slab[sr * pitch + i] * A.w[sr * A.B + b];
      }
    }
    out[(o * A.g + k) * (int64_t)inner + c0 + i] = acc * scale;
  }
}

// Innermost-axis variant (inner == 1, S % 4 == 0): the full-resolution pass, which reads the whole gradient once.
// A wave stages RW consecutive rows in its LDS slab with 16-byte loads (fusing (in - in2)); lane (r, kg) then takes the
// banded dot products of coefficients kg, kg + KG, kg + 2 KG, ... of row r from LDS.  The band of coefficient k is
// densified once per workgroup into LDS (Wd[k][j] = weight of input lo[k] + j), so the inner loop touches no global
// memory.
//
// Bank conflicts were this kernel (round 3: 31 us for 67 MB at cfg-2, 2.2 TB/s): with lanes = (4 rows) x (16 coefficients),
// rows 256 floats apart and bands starting every 16 floats, the 64 reads of one tap hit 4 banks -- a 16-way conflict, 16
// LDS passes per ds_read.  Now rows are S + 1 floats apart (bank = row + column), a wave holds RW = 64 / KG rows and the KG
// coefficients in flight at a step are NEIGHBOURS (kg + KG i): for a 16x upsampling (bands 16 apart, KG = 4) the 64 lanes of
// a step read 64 different banks.  Same sums in the same order: bit-identical to the round-3 kernel.
constexpr int kBrThreads = 128;      // 2 waves: a slab of 16 rows x 257 floats per wave is 16 KiB
template <int KG>
__global__ void __launch_bounds__(kBrThreads)
k_band_reduce_rows(const float* __restrict__ in, const float* __restrict__ in2, float* __restrict__ out, int64_t rows,
                   BandAxis A, float scale, int iters) {
  extern __shared__ __attribute__((aligned(16))) float slabs[];      // [waves][RW][S + 1]
  __shared__ float Wd[kBrMaxW];
  __shared__ int lo_s[64];
  constexpr int RW = 64 / KG, NWV = kBrThreads / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = A.S, g = A.g, B = A.B;
  const int WB = stage_band(A, Wd, lo_s);
  const bool dense = WB > 0;                  // otherwise: weights straight from the (global) band table
  const int q = S >> 2;                          // float4 per row
  const int pitch = S + 1;
  float* my = slabs + wave * RW * pitch;
  const int r = lane % RW, kg = lane / RW;
  for (int it = 0; it < iters; ++it) {
    const int64_t base = (((int64_t)blockIdx.x * iters + it) * NWV + wave) * RW;
    __syncthreads();   // previous pass consumed (first time: the tables are in place)
    for (int idx = lane; idx < RW * q; idx += 64) {
      const int rr = idx / q, x4 = idx - rr * q;
      const int64_t gr = base + rr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows) {
        v = *reinterpret_cast<const float4*>(in + gr * S + 4 * x4);
        if (in2) {
          const float4 u = *reinterpret_cast<const float4*>(in2 + gr * S + 4 * x4);
          v = make_float4(v.x - u.x, v.y - u.y, v.z - u.z, v.w - u.w);
        }
      }
      float* dst = my + rr * pitch + 4 * x4;     // (rows are S + 1 apart: dword stores)
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    const int64_t gr = base + r;
    if (gr < rows) {
      const float* x = my + r * pitch;
      if (dense) {
        // two coefficients of the lane at a time, eight taps requested before the first is used: the sums keep their
        // order (one fma chain per coefficient, ascending j) but the LDS round trips overlap.  A tap beyond the row
        // (lo + j >= S) carries weight 0 in Wd; its index is clamped into the row instead of leaving the loop early.
        for (int k0 = kg; k0 < g; k0 += 2 * KG) {
          const int k1 = k0 + KG;
          const bool two = k1 < g;
          const int lo0 = lo_s[k0], lo1 = lo_s[two ? k1 : k0];
          const float* w0 = Wd + k0 * WB;
          const float* w1 = Wd + (two ? k1 : k0) * WB;
          float a0 = 0.f, a1 = 0.f;
          int j = 0;
          for (; j + 8 <= WB; j += 8) {
            float x0[8], x1[8], c0[8], c1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              x0[u] = x[min(lo0 + j + u, S - 1)]; c0[u] = w0[j + u];
              x1[u] = x[min(lo1 + j + u, S - 1)]; c1[u] = w1[j + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 = fmaf(x0[u], c0[u], a0); a1 = fmaf(x1[u], c1[u], a1); }
          }
          for (; j < WB; ++j) {
            a0 = fmaf(x[min(lo0 + j, S - 1)], w0[j], a0);
            a1 = fmaf(x[min(lo1 + j, S - 1)], w1[j], a1);
          }
          out[gr * g + k0] = a0 * scale;
          if (two) out[gr * g + k1] = a1 * scale;
        }
      } else
      for (int k = kg; k < g; k += KG) {
        const int lo = A.lo[k];
        float acc = 0.f;
        {
          for (int s2 = lo; s2 < A.hi[k]; ++s2) {
            const int b = k - A.start[s2];
            if (b >= 0 && b < B) acc = fmaf(x[s2], A.w[s2 * B + b], acc);
          }
        }
        out[gr * g + k] = acc * scale;
      }
    }
  }
}

// The same pass with the densified bands handed in by the caller (bands.py builds them once per table on the host): the
// round-4 profile showed the kernel above spending its time not on bytes but on the per-workgroup table preparation -- four
// dependent L2 round trips (lo / hi -> start -> w) in front of the first barrier, 2048 times per launch (27-31 us for 67 MB
// at cfg-2).  Here a workgroup requests its first rows, then the dense table (one coalesced load), and only then waits;
// the rows of the next iteration are requested before the dot products of the current one.
template <int KG>
__global__ void __launch_bounds__(kBrThreads)
k_band_reduce_rows_dense(const float* __restrict__ in, const float* __restrict__ in2, float* __restrict__ out, int64_t rows,
                         const float* __restrict__ wd, const int* __restrict__ lo_g, int S, int g, int WB, float scale, int iters) {
  extern __shared__ __attribute__((aligned(16))) float slabs[];      // [waves][RW][S + 1], then Wd[g * WB]
  __shared__ int lo_s[64];
  constexpr int RW = 64 / KG, NWV = kBrThreads / 64;
  constexpr int QMAX = 8;                        // float4 per lane and iteration at most: RW * (S / 4) <= 512
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = S >> 2;
  const int pitch = S + 1;
  float* my = slabs + wave * RW * pitch;
  float* Wd = slabs + NWV * RW * pitch;
  const int r = lane % RW, kg = lane / RW;
  const int nq = RW * q;
  float4 v[QMAX];
  auto request = [&](int it) {
    const int64_t base = (((int64_t)blockIdx.x * iters + it) * NWV + wave) * RW;
#pragma unroll
    for (int u = 0; u < QMAX; ++u) {
      const int idx = lane + 64 * u;
      const int rr = idx / q, x4 = idx - rr * q;
      const int64_t gr = base + rr;
      const bool ok = idx < nq && gr < rows;
      const int64_t off = ok ? gr * S + 4 * x4 : 0;      // (unconditional loads from a clamped address)
      float4 a = *reinterpret_cast<const float4*>(in + off);
      if (in2) {
        const float4 b = *reinterpret_cast<const float4*>(in2 + off);
        a = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
      }
      v[u] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  request(0);
  for (int i = threadIdx.x; i < g * WB; i += kBrThreads) Wd[i] = wd[i];
  if (threadIdx.x < g) lo_s[threadIdx.x] = lo_g[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    const int64_t base = (((int64_t)blockIdx.x * iters + it) * NWV + wave) * RW;
    if (it > 0) __syncthreads();                 // the previous pass has consumed the slab
#pragma unroll
    for (int u = 0; u < QMAX; ++u) {
      const int idx = lane + 64 * u;
      if (idx >= nq) continue;
      const int rr = idx / q, x4 = idx - rr * q;
      float* dst = my + rr * pitch + 4 * x4;     // (rows are S + 1 apart: dword stores)
      dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
    }
    __syncthreads();
    if (it + 1 < iters) request(it + 1);         // in flight under the dot products
    const int64_t gr = base + r;
    if (gr < rows) {
      const float* x = my + r * pitch;
      for (int k0 = kg; k0 < g; k0 += 2 * KG) {
        const int k1 = k0 + KG;
        const bool two = k1 < g;
        const int lo0 = lo_s[k0], lo1 = lo_s[two ? k1 : k0];
        const float* w0 = Wd + k0 * WB;
        const float* w1 = Wd + (two ? k1 : k0) * WB;
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        // (taps the banded kernel skips -- beyond a coefficient's band, or clamped to S - 1 -- enter here with weight 0: the
        // same sums for FINITE input; an inf / NaN inside [lo, lo + WB) of a row turns its coefficient into NaN where the
        // banded kernel would not.  The solver's NaN guard voids such a step either way.)
        for (; j + 8 <= WB; j += 8) {
          float x0[8], x1[8], c0[8], c1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            x0[u] = x[min(lo0 + j + u, S - 1)]; c0[u] = w0[j + u];
            x1[u] = x[min(lo1 + j + u, S - 1)]; c1[u] = w1[j + u];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { a0 = fmaf(x0[u], c0[u], a0); a1 = fmaf(x1[u], c1[u], a1); }
        }
        for (; j < WB; ++j) {
          a0 = fmaf(x[min(lo0 + j, S - 1)], w0[j], a0);
          a1 = fmaf(x[min(lo1 + j, S - 1)], w1[j], a1);
        }
        out[gr * g + k0] = a0 * scale;
        if (two) out[gr * g + k1] = a1 * scale;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bias field
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bias_value(float L, int use_log, float eps, float& e, bool& pass) {
  e = use_log ? expf(L) : 1.f + L;
  const float b = e - 1.f;
  pass = (b >= -eps) && (b <= eps);          // torch.clamp: gradient on the closed interval
  return 1.f + fminf(fmaxf(b, -eps), eps);   // adv_bias.py:352-353
}

// grid = (i1 chunks, S0, N)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_bias_fwd(const float* __restrict__ cp, const float* __restrict__ data, float* __restrict__ out,
           float* __restrict__ field, BandTables T, Dims full, int C, float eps, int use_log, float cp_scale) {
  __shared__ float lds[kTpMaxLds];
  const int n = blockIdx.z, i0 = blockIdx.y;
  const int i1b = blockIdx.x * kTpChunk, i1e = min(i1b + kTpChunk, full.s1);
  const int64_t G = (int64_t)T.a[0].g * T.a[1].g * T.a[2].g;
  const int V = (int)full.voxels();
  tp_rows<VEC>(cp + (int64_t)n * G, T, full, i0, i1b, i1e, lds, [&](int i1, int x, const float (&val)[VEC]) {
    const int v = (i0 * full.s1 + i1) * full.s2 + x;
    float b[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float e;
      bool pass;
      b[q] = bias_value(cp_scale * val[q], use_log, eps, e, pass);
    }
    store_vec<VEC>(field + (int64_t)n * V + v, b);
    if (data) {
      for (int c = 0; c < C; ++c) {
        const int64_t o = ((int64_t)n * C + c) * V + v;
        float dv[VEC];
        load_vec<VEC>(data + o, dv);
#pragma unroll
        for (int q = 0; q < VEC; ++q) dv[q] *= b[q];
        store_vec<VEC>(out + o, dv);
      }
    }
  });
}

// gL = dLoss/dL (full res, one channel); gdata optional.  grid = (i1 chunks, S0, N)
template <int VEC>
__global__ void __launch_bounds__(kBlock)
k_bias_bwd(const float* __restrict__ cp, const float* __restrict__ data, const float* __restrict__ gout,
           float* __restrict__ gL, float* __restrict__ gdata, BandTables T, Dims full, int C, float eps, int use_log,
           float cp_scale) {
  __shared__ float lds[kTpMaxLds];
  const int n = blockIdx.z, i0 = blockIdx.y;
  const int i1b = blockIdx.x * kTpChunk, i1e = min(i1b + kTpChunk, full.s1);
  const int64_t G = (int64_t)T.a[0].g * T.a[1].g * T.a[2].g;
  const int V = (int)full.voxels();
  tp_rows<VEC>(cp + (int64_t)n * G, T, full, i0, i1b, i1e, lds, [&](int i1, int x, const float (&val)[VEC]) {
    const int v = (i0 * full.s1 + i1) * full.s2 + x;
    float b[VEC], e[VEC], sgo[VEC];
    bool pass[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      b[q] = bias_value(cp_scale * val[q], use_log, eps, e[q], pass[q]);
      sgo[q] = 0.f;
    }
    for (int c = 0; c < C; ++c) {
      const int64_t o = ((int64_t)n * C + c) * V + v;
      float go[VEC], dv[VEC];
      load_vec<VEC>(gout + o, go);
      load_vec<VEC>(data + o, dv);
#pragma unroll
      for (int q = 0; q < VEC; ++q) { sgo[q] += go[q] * dv[q]; go[q] *= b[q]; }
      if (gdata) store_vec<VEC>(gdata + o, go);
    }
    if (gL) {
      float gl[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) gl[q] = pass[q] ? sgo[q] * (use_log ? e[q] : 1.f) * cp_scale : 0.f;
      store_vec<VEC>(gL + (int64_t)n * V + v, gl);
    }
  });
}

// ---------------------------------------------------------------------------------------------
// separable Gaussian, one axis per launch (zero padding), with fused prologue / epilogue
//   PRE : 0 none | 1 x*scale | 2 border-identity(x) - identity   (x = sampling position, adv_morph.py:473-487)
//   POST: 0 none | 1 + identity (adv_morph.py:489) | 2 * slope(aux) (adjoint of PRE 2; aux = positions)
// channel c of plane p = p % C addresses axis (x <-> s2, y <-> s1, z <-> s0).
// ---------------------------------------------------------------------------------------------
struct GaussW { float w[9]; };

// value and slope of F.grid_sample(identity_grid, pos, border, align_corners=True) along one axis
__device__ __forceinline__ float border_identity(float pos, int S, float& slope) {
  float mult;
  const float x = source_index<PAD_BORDER>(pos, S, mult);
  const float f = floorf(x);
  const int i0 = (int)f;
  const int i1 = min(i0 + 1, S - 1);
  const float a = lin_coord(i0, S), b = lin_coord(i1, S);
  slope = mult * (b - a);
  return ((f + 1.f) - x) * a + (x - f) * b;
}

template <int PRE, int POST>
__global__ void __launch_bounds__(kBlock)
k_gauss_axis(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ aux, int64_t total,
             Dims d, int C, int axis, GaussW gw, float scale) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= total) return;
  const int V = (int)d.voxels();
  const int plane = (int)(t / V);
  const int v = (int)(t - (int64_t)plane * V);
  int idx[3];
  decode3(v, d, idx[0], idx[1], idx[2]);
  const int S[3] = {d.s0, d.s1, d.s2};
  const int stride = axis == 2 ? 1 : (axis == 1 ? d.s2 : d.s1 * d.s2);
  const int c = plane % C;
  const int caxis = 2 - c;  // the axis this channel's coordinate runs along
  const int ia = idx[axis];
  // all nine taps are loaded unconditionally (index clamped into the row, the value discarded by a select): a load
  // inside `if (in range)` gets its own branch and `s_waitcnt vmcnt(0)` -- nine serial memory round trips
  float tap[9];
#pragma unroll
  for (int k = -4; k <= 4; ++k) {
    const int jc = min(max(ia + k, 0), S[axis] - 1);
    tap[k + 4] = in[t + (int64_t)(jc - ia) * stride];
  }
  float acc = 0.f;
#pragma unroll
  for (int k = -4; k <= 4; ++k) {
    const int j = ia + k;
    float x = tap[k + 4];
    if (PRE == 1) x *= scale;
    if (PRE == 2) {
      const int ci = (caxis == axis) ? j : idx[caxis];
      float slope;
      x = border_identity(x, S[caxis], slope) - lin_coord(ci, S[caxis]);
    }
    acc += (j >= 0 && j < S[axis]) ? gw.w[k + 4] * x : 0.f;
  }
  if (POST == 1) acc += lin_coord(idx[caxis], S[caxis]);
  if (POST == 2) {
    float slope;
    border_identity(aux[t], S[caxis], slope);
    acc *= slope;
  }
  out[t] = acc;
}

// Same pass with 16 bytes per lane: a thread owns 4 consecutive x.  Measured rule on MI355X: a vector-memory
// instruction costs a CU ~26 clk (dword) to ~47 clk (dwordx4) whatever it carries, so the pass time is the number
// of load instructions -- 9 scalar loads per output (scalar kernel: 0.9 TB/s) vs 9 float4 loads per 4 outputs along
// y/z and 3 per 4 outputs along x.  Needs S2 % 4 == 0 and 16-byte aligned planes.
template <int PRE, int POST, int AXIS>
__global__ void __launch_bounds__(kBlock)
k_gauss_axis_v4(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ aux, int64_t total4,
                Dims d, int C, GaussW gw, float scale, int row_in_wave) {
  const int64_t t4 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t t;
  if (AXIS == 2 && row_in_wave == 2) {
    // rows of S2/4 lanes that do not divide a wave (80 voxels = 20 lanes): floor(64 / lanes) whole rows per wave, the
    // tail lanes idle -- the DPP form below needs every row inside one wave
    const int q = d.s2 >> 2, rpw = 64 / q;
    const int lane = (int)(t4 & 63);
    const int rl = lane / q;
    const int64_t row = (t4 >> 6) * rpw + rl;
    if (rl >= rpw || row >= total4 / q) return;
    t = row * d.s2 + (lane - rl * q) * 4;
  } else {
    if (t4 >= total4) return;
    t = t4 * 4;
  }
  const int V = (int)d.voxels();
  const int plane = (int)(t / V);
  const int v = (int)(t - (int64_t)plane * V);
  int idx[3];
  decode3(v, d, idx[0], idx[1], idx[2]);
  const int S[3] = {d.s0, d.s1, d.s2};
  const int c = plane % C;
  const int caxis = 2 - c;
  auto pre = [&](float x, int i0, int i1, int i2) -> float {
    if (PRE == 1) return x * scale;
    if (PRE == 2) {
      const int ci = caxis == 2 ? i2 : (caxis == 1 ? i1 : i0);
      float slope;
      return border_identity(x, S[caxis], slope) - lin_coord(ci, S[caxis]);
    }
    return x;
  };
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (AXIS == 2 && row_in_wave) {
    // a row (S2/4 lanes) never straddles a wave and no lane is idle: the neighbouring quads come from the neighbouring
    // lanes (whole-wave DPP shifts) instead of two more 16-byte loads and eight more prologue evaluations per lane
    const float4 q = *reinterpret_cast<const float4*>(in + t);
    float win[12];
    win[4] = pre(q.x, idx[0], idx[1], idx[2]);
    win[5] = pre(q.y, idx[0], idx[1], idx[2] + 1);
    win[6] = pre(q.z, idx[0], idx[1], idx[2] + 2);
    win[7] = pre(q.w, idx[0], idx[1], idx[2] + 3);
    const bool first = idx[2] == 0, last = idx[2] + 4 >= d.s2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float pv = lane_prev_f(win[4 + k]), nx = lane_next_f(win[4 + k]);
      win[k] = first ? 0.f : pv;
      win[8 + k] = last ? 0.f : nx;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o] += gw.w[k] * win[o + k];
  } else if (AXIS == 2) {
    float win[12];
    float4 qb[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {   // unconditional loads (a quad beyond the row re-reads the own quad and is discarded)
      const int x0 = idx[2] + (b - 1) * 4;
      qb[b] = *reinterpret_cast<const float4*>(in + t + ((x0 >= 0 && x0 < d.s2) ? (b - 1) * 4 : 0));
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int x0 = idx[2] + (b - 1) * 4;
      const bool inr = x0 >= 0 && x0 < d.s2;
      win[b * 4 + 0] = inr ? pre(qb[b].x, idx[0], idx[1], x0) : 0.f;
      win[b * 4 + 1] = inr ? pre(qb[b].y, idx[0], idx[1], x0 + 1) : 0.f;
      win[b * 4 + 2] = inr ? pre(qb[b].z, idx[0], idx[1], x0 + 2) : 0.f;
      win[b * 4 + 3] = inr ? pre(qb[b].w, idx[0], idx[1], x0 + 3) : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o] += gw.w[k] * win[o + k];
  } else {
    const int stride = AXIS == 1 ? d.s2 : d.s1 * d.s2;
    const int ia = idx[AXIS];
    // nine unconditional 16-byte loads in flight together (row index clamped, out-of-range taps discarded by a select):
    // inside `if (in range)` every load got its own branch and `s_waitcnt vmcnt(0)`
    float4 q[9];
#pragma unroll
    for (int k = -4; k <= 4; ++k) {
      const int jc = min(max(ia + k, 0), S[AXIS] - 1);
      q[k + 4] = *reinterpret_cast<const float4*>(in + t + (int64_t)(jc - ia) * stride);
    }
#pragma unroll
    for (int k = -4; k <= 4; ++k) {
      const int j = ia + k;
      const bool inr = j >= 0 && j < S[AXIS];
      const int j0 = AXIS == 0 ? j : idx[0], j1 = AXIS == 1 ? j : idx[1];
      const float w = gw.w[k + 4];
      acc[0] += inr ? w * pre(q[k + 4].x, j0, j1, idx[2]) : 0.f;
      acc[1] += inr ? w * pre(q[k + 4].y, j0, j1, idx[2] + 1) : 0.f;
      acc[2] += inr ? w * pre(q[k + 4].z, j0, j1, idx[2] + 2) : 0.f;
      acc[3] += inr ? w * pre(q[k + 4].w, j0, j1, idx[2] + 3) : 0.f;
    }
  }
  if (POST == 1) {
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] += lin_coord(caxis == 2 ? idx[2] + o : idx[caxis], S[caxis]);
  }
  if (POST == 2) {
    const float4 a = *reinterpret_cast<const float4*>(aux + t);
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float slope;
      border_identity(av[o], S[caxis], slope);
      acc[o] *= slope;
    }
  }
  *reinterpret_cast<float4*>(out + t) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// x and y passes in one launch: a workgroup owns TY whole rows of one (plane, z) slice, stages rows y0-4 .. y0+TY+3 with
// the prologue applied, runs the x pass LDS -> LDS on all staged rows and the y pass LDS -> registers on its own.  One read
// and one write of the tensor instead of two each; the x pass is done (TY+8)/TY times.  Same tap order per output as the
// per-axis kernels (k = -4..4, x first; multiply-adds are fused here, so results agree to a few ulp, not bit for bit).  POST belongs to the LAST pass: this kernel in 2D, the z pass in 3D.
template <int PRE, int POST>
__global__ void __launch_bounds__(kBlock)
k_gauss_xy(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ aux, Dims d, int C, GaussW gw,
           float scale, int TY, const float* __restrict__ in_hi, int planes_lo) {
  extern __shared__ float gx_lds[];
  const int W = d.s2, Q = W >> 2, PA = W + 8;
  const int R = TY + 8;
  float* A = gx_lds;                       // [R][PA]: 4 zeros | row | 4 zeros
  float* B = gx_lds + R * PA;              // [R][W]
  const int plane = blockIdx.z, iz = blockIdx.y;
  const int y0 = blockIdx.x * TY;
  const int caxis = 2 - plane % C;
  const int Sc = caxis == 2 ? d.s2 : (caxis == 1 ? d.s1 : d.s0);
  const int64_t base = ((int64_t)plane * d.s0 + iz) * d.s1 * (int64_t)W;
  // the input may come as two tensors (planes [0, planes_lo) and the rest): the two halves of a paired field's gradient
  const float* src = (in_hi && plane >= planes_lo) ? in_hi + (((int64_t)(plane - planes_lo) * d.s0 + iz) * d.s1 * (int64_t)W) : in + base;
  for (int e = threadIdx.x; e < R * 2; e += kBlock)
    *reinterpret_cast<float4*>(A + (e >> 1) * PA + ((e & 1) ? W + 4 : 0)) = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- stage (unconditional loads from the clamped row, rows outside the volume zeroed by a select)
  for (int e = threadIdx.x; e < R * Q; e += kBlock) {
    const int r = e / Q, q = e - r * Q;
    const int gy = y0 - 4 + r;
    const bool inr = gy >= 0 && gy < d.s1;
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)min(max(gy, 0), d.s1 - 1) * W + 4 * q);
    float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (PRE == 1) t[o] *= scale;
      if (PRE == 2) {
        float sl;
        t[o] = border_identity(t[o], Sc, sl) - lin_coord(caxis == 2 ? 4 * q + o : (caxis == 1 ? gy : iz), Sc);
      }
      if (!inr) t[o] = 0.f;
    }
    *reinterpret_cast<float4*>(A + r * PA + 4 + 4 * q) = make_float4(t[0], t[1], t[2], t[3]);
  }
  __syncthreads();
  // ---- x pass on every staged row
  for (int e = threadIdx.x; e < R * Q; e += kBlock) {
    const int r = e / Q, q = e - r * Q;
    const float* a = A + r * PA + 4 * q;                 // inputs x-4 .. x+7 of outputs x .. x+3
    const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4),
                 a2 = *reinterpret_cast<const float4*>(a + 8);
    const float win[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o] += gw.w[k] * win[o + k];
    *reinterpret_cast<float4*>(B + r * W + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();
  // ---- y pass on the owned rows
  for (int e = threadIdx.x; e < TY * Q; e += kBlock) {
    const int ry = e / Q, q = e - ry * Q;
    const int gy = y0 + ry;
    if (gy >= d.s1) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 b = *reinterpret_cast<const float4*>(B + (ry + k) * W + 4 * q);
      acc[0] += gw.w[k] * b.x; acc[1] += gw.w[k] * b.y; acc[2] += gw.w[k] * b.z; acc[3] += gw.w[k] * b.w;
    }
    const int64_t off = base + (int64_t)gy * W + 4 * q;
    if (POST == 1) {
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += lin_coord(caxis == 2 ? 4 * q + o : (caxis == 1 ? gy : iz), Sc);
    }
    if (POST == 2) {
      const float4 a4 = *reinterpret_cast<const float4*>(aux + off);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float slope;
        border_identity(av[o], Sc, slope);
        acc[o] *= slope;
      }
    }
    *reinterpret_cast<float4*>(out + off) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// z pass, marching: a thread owns 4 consecutive x of one (plane, y) column and walks a chunk of ZC planes with the taps in
// registers -- 1 + 8/ZC 16-byte loads per output instead of 9.  The per-output kernel above asks for every plane nine
// times from workgroups that run on different XCDs at the same moment (no L2 reuse): at 8 x 3 x 160 x 160 x 80 the z pass
// took 204-262 us against 91 us for the y pass, whose nine taps lie within 9 rows.  Same tap order per output (k = -4..4).
constexpr int kGaussMarchU = 9;          // outputs per round = loads in flight per thread
template <int PRE, int POST>
__global__ void __launch_bounds__(kBlock)
k_gauss_march_z(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ aux, int64_t planes, Dims d,
                int C, GaussW gw, float scale, int zc, int nchunk) {
  constexpr int U = kGaussMarchU;
  const int q = d.s2 >> 2;
  int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int xq = (int)(r % q);
  r /= q;
  const int iy = (int)(r % d.s1);
  r /= d.s1;
  const int chunk = (int)(r % nchunk);
  const int64_t plane = r / nchunk;
  if (plane >= planes) return;
  const int V = (int)d.voxels();
  const int stride = d.s1 * d.s2;
  const int caxis = 2 - (int)(plane % C);
  const int Sc = caxis == 2 ? d.s2 : (caxis == 1 ? d.s1 : d.s0);
  const int ix = xq * 4;
  const float* src = in + plane * V + iy * d.s2 + ix;
  float* dst = out + plane * V + iy * d.s2 + ix;
  const float* ax = POST == 2 ? aux + plane * V + iy * d.s2 + ix : nullptr;
  auto fetch = [&](int j) -> float4 {     // unconditional load from the clamped plane
    return *reinterpret_cast<const float4*>(src + (int64_t)min(max(j, 0), d.s0 - 1) * stride);
  };
  auto prep = [&](float4 v, int j) -> float4 {
    if (j < 0 || j >= d.s0) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRE == 1) return make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    if (PRE == 2) {
      float sl;
      float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int o = 0; o < 4; ++o) e[o] = border_identity(e[o], Sc, sl) - lin_coord(caxis == 2 ? ix + o : (caxis == 1 ? iy : j), Sc);
      return make_float4(e[0], e[1], e[2], e[3]);
    }
    return v;
  };
  const int ja = chunk * zc, jb = min(ja + zc, d.s0);
  float4 w[8 + U];                       // inputs j0-4 .. j0+U+3 of the current round
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = prep(fetch(ja - 4 + k), ja - 4 + k);
  for (int j0 = ja; j0 < jb; j0 += U) {
    float4 nw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nw[u] = fetch(j0 + 4 + u);
#pragma unroll
    for (int u = 0; u < U; ++u) w[8 + u] = prep(nw[u], j0 + 4 + u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u;
      if (j >= jb) break;                // uniform over the workgroup's waves (same chunk)
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        acc[0] += gw.w[k] * w[u + k].x;
        acc[1] += gw.w[k] * w[u + k].y;
        acc[2] += gw.w[k] * w[u + k].z;
        acc[3] += gw.w[k] * w[u + k].w;
      }
      if (POST == 1) {
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] += lin_coord(caxis == 2 ? ix + o : (caxis == 1 ? iy : j), Sc);
      }
      if (POST == 2) {
        const float4 a = *reinterpret_cast<const float4*>(ax + (int64_t)j * stride);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float slope;
          border_identity(av[o], Sc, slope);
          acc[o] *= slope;
        }
      }
      *reinterpret_cast<float4*>(dst + (int64_t)j * stride) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = w[U + k];
  }
}

// ---------------------------------------------------------------------------------------------
// Small planes (the low-resolution velocity grids: 16 x 16, 8 x 8 x 32 ...): all axes in ONE launch, one workgroup
// per plane, ping-pong in LDS.  The per-axis launches above take 5-16 us each on such planes (a few dozen
// workgroups, nine dependent loads per thread): 2-3 launches per field and direction, ~60 per solver call.
// Same tap order and zero padding as the per-axis kernels.  pre: 0 | 1 (x * scale); no post.
// ---------------------------------------------------------------------------------------------
constexpr int kGaussSmallMax = 4096;   // voxels per plane: 2 x 16 KiB of LDS

// mirror 0: one plane per workgroup.  mirror 1 (the batch [v; -v] of a paired field, never materialised): 2 P workgroups,
// workgroup b reads plane b % P with the scale negated for b >= P -- (-s) * v == -(s * v) exactly, so the second half is
// bit-identical to smoothing a negated copy.  mirror 2 (its adjoint): P workgroups, out[b] = G(s in[b]) - G(s in[P + b]),
// the two smoothings done one after the other with the arithmetic of two separate calls.
// NT threads: 256 for planes of up to 1024 voxels, 1024 beyond (a plane is ONE workgroup: with 4 waves a cfg-5 plane of 4000
// voxels was a chain of 48 dependent voxel steps per thread on an otherwise idle CU, 46.6 us a launch)
template <int NT>
__global__ void __launch_bounds__(NT)
k_gauss_small(const float* __restrict__ in, float* __restrict__ out, Dims d, int ndim, GaussW gw, float scale, int mirror,
              int P) {
  __shared__ float buf[2][kGaussSmallMax];
  const int V = (int)d.voxels();
  const int S[3] = {d.s0, d.s1, d.s2};
  const int stride[3] = {d.s1 * d.s2, d.s2, 1};
  const int b = blockIdx.x;
  float first[(kGaussSmallMax + NT - 1) / NT];
  const int rounds = mirror == 2 ? 2 : 1;
  for (int round = 0; round < rounds; ++round) {
    const float* src = in + (int64_t)(mirror == 1 ? b % P : b + round * P) * V;
    const float sc = (mirror == 1 && b >= P) ? -scale : scale;
    if (round) __syncthreads();
    for (int i = threadIdx.x; i < V; i += NT) buf[0][i] = src[i] * sc;
    __syncthreads();
    int cur = 0;
    for (int pass = 0; pass < ndim; ++pass) {
      const int axis = 2 - pass;              // innermost first, like advchain_gauss_axis is called
      const int Sa = S[axis], st = stride[axis];
      for (int i = threadIdx.x; i < V; i += NT) {
        const int ia = (i / st) % Sa;
        float acc = 0.f;
#pragma unroll
        for (int k = -4; k <= 4; ++k) {
          const int j = ia + k;
          if (j >= 0 && j < Sa) acc += gw.w[k + 4] * buf[cur][i + k * st];
        }
        buf[cur ^ 1][i] = acc;
      }
      __syncthreads();
      cur ^= 1;
    }
    float* dst = out + (int64_t)b * V;
    if (mirror != 2) {
      for (int i = threadIdx.x; i < V; i += NT) dst[i] = buf[cur][i];
    } else if (round == 0) {
#pragma unroll
      for (int q = 0; q < (kGaussSmallMax + NT - 1) / NT; ++q) {
        const int i = threadIdx.x + q * NT;
        first[q] = i < V ? buf[cur][i] : 0.f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < (kGaussSmallMax + NT - 1) / NT; ++q) {
        const int i = threadIdx.x + q * NT;
        if (i < V) dst[i] = first[q] - buf[cur][i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// streaming elementwise
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_axpy(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, float a, int64_t n4,
       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n4) {
    const float4 p = reinterpret_cast<const float4*>(y)[i];
    float4 q = x ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    q.x += a * p.x; q.y += a * p.y; q.z += a * p.z; q.w += a * p.w;
    reinterpret_cast<float4*>(out)[i] = q;
  }
  const int64_t j = n4 * 4 + i;  // scalar tail (everything when the buffers are not 16-byte aligned)
  if (j < n) out[j] = (x ? x[j] : 0.f) + a * y[j];
}

// out = (base ? base : 0) + a * sign(x)   (sign(0) = 0, sign(NaN) = NaN, like torch.sign)
__global__ void __launch_bounds__(kBlock)
k_sign_axpy(const float* __restrict__ base, const float* __restrict__ x, float* __restrict__ out, float a, int64_t n,
            const float* __restrict__ gate, const float* __restrict__ old) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (gate && !(fabsf(gate[0]) <= 3.0e38f)) { out[i] = old[i]; return; }     // see k_norm_axpy
  const float v = x[i];
  const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);     // (v itself: +-0 stays 0, NaN stays NaN)
  out[i] = (base ? base[i] : 0.f) + a * sg;
}

// out = (x != 0) ? 1 : 0   (NaN != 0 is true)
__global__ void __launch_bounds__(kBlock)
k_nonzero_mask(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = x[i] != 0.f ? 1.f : 0.f;
}

// partial[n][b] = sum over chunk b of x[n][:]^2
__global__ void __launch_bounds__(kBlock)
k_sumsq_partial(const float* __restrict__ x, float* __restrict__ partial, int64_t M, int chunk) {
  __shared__ float smem[4];
  const int n = blockIdx.y, b = blockIdx.x;
  const float* xn = x + (int64_t)n * M;
  const int64_t lo = (int64_t)b * chunk, hi = min(lo + chunk, M);
  float s[1] = {0.f};
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) { const float q = xn[i]; s[0] += q * q; }
  block_sum<1>(s, smem);
  if (threadIdx.x == 0) partial[(int64_t)n * gridDim.x + b] = s[0];
}

// out[n][:] = (base ? base[n][:] : 0) + step * x[n][:] / (sqrt(sum partial[n][:]) + 1e-20)
__global__ void __launch_bounds__(kBlock)
k_norm_axpy(const float* __restrict__ base, const float* __restrict__ x, const float* __restrict__ partial,
            int nb, float step, float* __restrict__ out, int64_t M, int chunk, const float* __restrict__ gate,
            const float* __restrict__ old) {
  __shared__ float smem[4];
  __shared__ float inv_s;
  const int n = blockIdx.y, b = blockIdx.x;
  float s[1] = {0.f};
  for (int i = threadIdx.x; i < nb; i += kBlock) s[0] += partial[(int64_t)n * nb + i];
  block_sum<1>(s, smem);
  if (threadIdx.x == 0) inv_s = step / (sqrtf(s[0]) + 1e-20f);
  __syncthreads();
  const float inv = inv_s;
  const int64_t lo = (int64_t)b * chunk, hi = min(lo + chunk, M);
  const int64_t off = (int64_t)n * M;
  // gate (optional device scalar, e.g. the loss): when it is NaN / inf the update is void and `old` is kept -- the NaN guard
  // of the ascent loop (adv_compose_solver.py:343-347) without a host read-back in the middle of the step
  if (gate && !(fabsf(gate[0]) <= 3.0e38f)) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) out[off + i] = old[off + i];
    return;
  }
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) out[off + i] = (base ? base[off + i] : 0.f) + inv * x[off + i];
}

// rows of at most one chunk (the low-resolution parameters of AdvBias / AdvMorph / AdvAffine): both steps in one launch,
// one workgroup per row, same summation order as the two-launch form
__global__ void __launch_bounds__(kBlock)
k_norm_axpy_row(const float* __restrict__ base, const float* __restrict__ x, float step, float* __restrict__ out, int64_t M,
                const float* __restrict__ gate, const float* __restrict__ old) {
  __shared__ float smem[4];
  __shared__ float inv_s;
  const int64_t off = (int64_t)blockIdx.x * M;
  float s[1] = {0.f};
  for (int64_t i = threadIdx.x; i < M; i += kBlock) { const float q = x[off + i]; s[0] += q * q; }
  block_sum<1>(s, smem);
  if (threadIdx.x == 0) inv_s = step / (sqrtf(s[0]) + 1e-20f);
  __syncthreads();
  const float inv = inv_s;
  if (gate && !(fabsf(gate[0]) <= 3.0e38f)) {      // see k_norm_axpy
    for (int64_t i = threadIdx.x; i < M; i += kBlock) out[off + i] = old[off + i];
    return;
  }
  for (int64_t i = threadIdx.x; i < M; i += kBlock) out[off + i] = (base ? base[off + i] : 0.f) + inv * x[off + i];
}

// ---- the parameter updates of ONE ascent step in ONE launch (round 6: launch consolidation) ------------------------------
// An ascent step ends with one update per transform (adv_compose_solver.py:349-364 -> the optimize_parameters of
// adv_noise.py:51-64, adv_bias.py:139-148, adv_morph.py:501-516, adv_affine.py:182-198): a unit-normalised step for noise /
// bias / morph, a sign step for affine -- five launches of 5-10 us at cfg-2 (two for the 65536-element noise rows), each
// behind a launch boundary.  Here a workgroup takes one (transform, sample) row: 1024 threads square-sum the row (16 bytes
// per lane where the row allows), reduce through the waves, and write base + step * x / (||x|| + 1e-20) (or the sign
// step); the gate is the one of k_norm_axpy.  Same formula as the separate kernels; the order of the square sum differs
// (one workgroup per row at every size), i.e. the result agrees with theirs to rounding, not bit for bit.
struct UpdDesc {
  const float* base;
  const float* x;
  float* out;
  const float* old;
  long long M;
  int rows, kind, block0;
  float step;
};
constexpr int kUpdMax = 8;
struct UpdPack {
  UpdDesc d[kUpdMax];
  int n;
};

__global__ void __launch_bounds__(1024) k_update_multi(UpdPack p, const float* __restrict__ gate) {
  __shared__ float wsum[16];
  __shared__ float inv_s;
  int i = 0;
#pragma unroll
  for (int k = 1; k < kUpdMax; ++k)
    if (k < p.n && (int)blockIdx.x >= p.d[k].block0) i = k;
  const UpdDesc u = p.d[i];
  const long long off = (long long)((int)blockIdx.x - u.block0) * u.M;
  const float* __restrict__ x = u.x + off;
  const float* __restrict__ base = u.base ? u.base + off : nullptr;
  float* __restrict__ out = u.out + off;
  const int tid = threadIdx.x;
  if (gate && !(fabsf(gate[0]) <= 3.0e38f)) {      // the NaN guard of the step: the old parameters stay
    for (long long j = tid; j < u.M; j += 1024) out[j] = u.old[off + j];
    return;
  }
  if (u.kind == 1) {                                // sign step
    for (long long j = tid; j < u.M; j += 1024) {
      const float v = x[j];
      const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);
      out[j] = (base ? base[j] : 0.f) + u.step * sg;
    }
    return;
  }
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0 &&
                  (u.M & 3) == 0;
  // (eight 16-byte requests per thread in flight in both passes: a 65536-element noise row is 16 float4 per thread, and one
  // request per loop trip made the row a chain of 16 memory round trips -- as slow as the two launches it replaces)
  constexpr int UB = 8;
  float s = 0.f;
  if (al) {
    const long long M4 = u.M >> 2;
    for (long long j0 = tid; j0 < M4; j0 += (long long)UB * 1024) {
      float4 q[UB];
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const long long j = j0 + (long long)k * 1024;
        q[k] = reinterpret_cast<const float4*>(x)[j < M4 ? j : M4 - 1];
      }
#pragma unroll
      for (int k = 0; k < UB; ++k)
        if (j0 + (long long)k * 1024 < M4) s += (q[k].x * q[k].x + q[k].y * q[k].y) + (q[k].z * q[k].z + q[k].w * q[k].w);
    }
  } else {
    for (long long j = tid; j < u.M; j += 1024) { const float q = x[j]; s += q * q; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wsum[w];
    inv_s = u.step / (sqrtf(t) + 1e-20f);
  }
  __syncthreads();
  const float inv = inv_s;
  if (al) {
    const long long M4 = u.M >> 2;
    for (long long j0 = tid; j0 < M4; j0 += (long long)UB * 1024) {
      float4 q[UB], b[UB];
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const long long j = j0 + (long long)k * 1024, jc = j < M4 ? j : M4 - 1;
        q[k] = reinterpret_cast<const float4*>(x)[jc];
        b[k] = base ? reinterpret_cast<const float4*>(base)[jc] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < UB; ++k) {
        const long long j = j0 + (long long)k * 1024;
        if (j < M4)
          reinterpret_cast<float4*>(out)[j] = make_float4(b[k].x + inv * q[k].x, b[k].y + inv * q[k].y, b[k].z + inv * q[k].z,
                                                          b[k].w + inv * q[k].w);
      }
    }
  } else {
    for (long long j = tid; j < u.M; j += 1024) out[j] = (base ? base[j] : 0.f) + inv * x[j];
  }
}

}  // namespace advchain

using namespace advchain;

// (the descriptor of include/advchain_hip.h; this translation unit does not include the public header)
struct advchain_update_desc {
  const float* base;
  const float* x;
  float* out;
  const float* old;
  int64_t N;
  int64_t M;
  int32_t kind;
  float step;
};

static inline bool fdims_ok(int ndim, const int64_t* s) {
  if (ndim != 2 && ndim != 3) return false;
  for (int i = 0; i < ndim; ++i)
    if (s[i] < 1 || s[i] > (1 << 24)) return false;
  return true;
}
static inline Dims fmake_dims(int ndim, const int64_t* s) {
  Dims d;
  if (ndim == 3) { d.s0 = (int)s[0]; d.s1 = (int)s[1]; d.s2 = (int)s[2]; }
  else { d.s0 = 1; d.s1 = (int)s[0]; d.s2 = (int)s[1]; }
  return d;
}

// Band-table buffers (device), built by the host (advchain_amd/bands.py):
//   itab: for each axis a (3 axes; 2D passes a trivial leading axis):  start[S_a] | lo[g_a] | hi[g_a]
//   ftab: for each axis a: w[S_a * B_a]
static bool unpack_tables(const int32_t* itab, const float* ftab, const int64_t* S, const int64_t* g, const int64_t* B,
                          BandTables& T) {
  const int32_t* ip = itab;
  const float* fp = ftab;
  for (int a = 0; a < 3; ++a) {
    if (S[a] < 1 || g[a] < 1 || B[a] < 1 || B[a] > kBandMax || B[a] > g[a]) return false;
    T.a[a].S = (int)S[a]; T.a[a].g = (int)g[a]; T.a[a].B = (int)B[a];
    T.a[a].start = ip; ip += S[a];
    T.a[a].lo = ip; ip += g[a];
    T.a[a].hi = ip; ip += g[a];
    T.a[a].w = fp; fp += S[a] * B[a];
  }
  return true;
}

extern "C" {

int advchain_tp_interp_fwd(const float* coef, float* out, const int32_t* itab, const float* ftab, const int64_t* S,
                           const int64_t* g, const int64_t* B, int64_t planes, int64_t C, int ndim, int add_identity,
                           float scale, float* sumsq, float* disp_out, void* stream) {
  ADVCHAIN_CHECK_ARG(coef && itab && ftab && (out || sumsq), "tp_interp_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(planes >= 0 && planes < 65536 && C >= 1 && S[0] < 65536, "tp_interp_fwd: bad planes/C");
  BandTables T;
  ADVCHAIN_CHECK_ARG(unpack_tables(itab, ftab, S, g, B, T), "tp_interp_fwd: bad band tables");
  if (planes == 0) return ADVCHAIN_OK;
  Dims full{(int)S[0], (int)S[1], (int)S[2]};
  ADVCHAIN_CHECK_ARG(full.voxels() < (1ll << 31), "tp_interp_fwd: volume too large");
  ADVCHAIN_CHECK_ARG(T.a[2].g <= kTpMaxLds, "tp_interp_fwd: coefficient row too long");
  dim3 grid((unsigned)((full.s1 + kTpChunk - 1) / kTpChunk), (unsigned)full.s0, (unsigned)planes);
  static const bool no_planes = getenv("ADVCHAIN_NO_TP_PLANES") != nullptr;   // A/B knob
  if (!no_planes && full.s0 >= 2 * kTpZB && T.a[0].B == 2 && T.a[1].B == 2 && T.a[2].B == 2 && kTpChunk * T.a[2].g <= kTpMaxLds) {
    // 3D linear upsampling: kTpZB planes per workgroup, pipelined
    dim3 gp(grid.x, (unsigned)((full.s0 + kTpZB - 1) / kTpZB), (unsigned)planes);
    // 16 bytes per lane for long rows and for rows that do not fill whole waves one voxel per lane (cfg-5's rows of 80: 90
    // against 107 us); rows of exactly 64 time the same either way
    if (full.s2 % 4 == 0 && (full.s2 >= 128 || full.s2 % 64 != 0) && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
      hipLaunchKernelGGL(k_tp_interp_fwd_planes<4>, gp, dim3(kBlock), 0, (hipStream_t)stream, coef, out, T, full, (int)C, add_identity,
                         scale, sumsq, disp_out);
    else
      hipLaunchKernelGGL(k_tp_interp_fwd_planes<1>, gp, dim3(kBlock), 0, (hipStream_t)stream, coef, out, T, full, (int)C, add_identity,
                         scale, sumsq, disp_out);
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  // short rows (S2 < 128) leave a thread only two rows to amortise the bands of its 4 columns: measured slower (3D 53 -> 61 us)
  if (full.s2 % 4 == 0 && full.s2 >= 128 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(k_tp_interp_fwd<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, coef, out, T, full, (int)C,
                       add_identity, scale, sumsq, disp_out);
  else
    hipLaunchKernelGGL(k_tp_interp_fwd<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, coef, out, T, full, (int)C,
                       add_identity, scale, sumsq, disp_out);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// advchain_gauss_small_pair (forward) + advchain_tp_interp_fwd in ONE launch: vel (P planes of g1 x g2) -> s1 (2P planes, the
// smoothed batch [v; -v]) and out (2P planes of S1 x S2) = (add_identity ? identity : 0) + scale * up(s1).  2D, planes of at most
// 1024 values, the 9-tap window.  ADVCHAIN_ERR_UNSUPPORTED (-2), nothing enqueued, for anything else: issue the two calls.
int advchain_tp_interp_fwd_smoothed_pair(const float* vel, float* s1, float* out, const int32_t* itab, const float* ftab,
                                         const int64_t* S, const int64_t* g, const int64_t* B, int64_t P, int64_t C,
                                         int add_identity, float scale, float* disp_out, const float* weights9, float gscale,
                                         void* stream) {
  ADVCHAIN_CHECK_ARG(vel && out && itab && ftab && weights9, "tp_interp_fwd_smoothed_pair: null pointer");
  ADVCHAIN_CHECK_ARG(P >= 0 && 2 * P < 65536 && C >= 1, "tp_interp_fwd_smoothed_pair: bad planes/C");
  BandTables T;
  ADVCHAIN_CHECK_ARG(unpack_tables(itab, ftab, S, g, B, T), "tp_interp_fwd_smoothed_pair: bad band tables");
  static const bool off = getenv("ADVCHAIN_NO_TP_GS_FUSE") != nullptr;   // A/B knob
  if (off || S[0] != 1 || g[0] != 1 || g[1] * g[2] > kGsFuseMax || T.a[2].g > kTpMaxLds) return ADVCHAIN_ERR_UNSUPPORTED;
  if (P == 0) return ADVCHAIN_OK;
  Dims full{(int)S[0], (int)S[1], (int)S[2]};
  ADVCHAIN_CHECK_ARG(full.voxels() < (1ll << 31), "tp_interp_fwd_smoothed_pair: volume too large");
  GaussW9 gw;
  for (int k = 0; k < 9; ++k) gw.w[k] = weights9[k];
  dim3 grid((unsigned)((full.s1 + kTpChunk - 1) / kTpChunk), 1u, (unsigned)(2 * P));
  if (full.s2 % 4 == 0 && full.s2 >= 128 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(k_tp_interp_fwd_gs<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, vel, s1, out, T, full, (int)C, add_identity,
                       scale, disp_out, gw, gscale, (int)P);
  else
    hipLaunchKernelGGL(k_tp_interp_fwd_gs<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, vel, s1, out, T, full, (int)C, add_identity,
                       scale, disp_out, gw, gscale, (int)P);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// One adjoint pass along `axis` (0..2 of the padded 3-axis view).  in: (outer, S_axis, inner) -> out: (outer, g_axis, inner)
int advchain_band_reduce_axis(const float* in, const float* in2, float* out, const int32_t* itab, const float* ftab,
                              const int64_t* S, const int64_t* g, const int64_t* B, int axis, int64_t outer,
                              int64_t inner, float scale, void* stream) {
  ADVCHAIN_CHECK_ARG(in && out && itab && ftab, "band_reduce_axis: null pointer");
  ADVCHAIN_CHECK_ARG(axis >= 0 && axis < 3 && outer >= 0 && inner >= 1 && inner < (1ll << 31), "band_reduce_axis: bad axis/outer/inner");
  BandTables T;
  ADVCHAIN_CHECK_ARG(unpack_tables(itab, ftab, S, g, B, T), "band_reduce_axis: bad band tables");
  const int64_t total = outer * T.a[axis].g * inner;
  if (total == 0) return ADVCHAIN_OK;
  const BandAxis& A = T.a[axis];
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
  if (inner == 1 && A.S % 4 == 0 && A.S <= 1024 && A.g <= 64 && aligned) {
    // coefficient groups per wave KG (a wave holds 64 / KG rows of S + 1 floats): 4 is conflict-free for a 16x upsampling,
    // but 16 rows a wave leave a 16 k-row problem with 1024 waves -- take 8 (2-way conflicts) or 16 (4-way) until there
    // are ~4096 waves, and whatever keeps the slab of a wave within 16 KiB
    int KG = A.g >= 4 ? 4 : (A.g >= 2 ? 2 : 1);
    while (KG < 16 && KG * 2 <= A.g && ((size_t)(64 / KG) * (A.S + 1) * sizeof(float) > 16 * 1024 + 512 || outer / (64 / KG) < 4096)) KG *= 2;
    const int RW = 64 / KG;
    int iters = (int)(outer / ((int64_t)RW * (kBrThreads / 64) * 2048));      // ~2048 workgroups at least
    iters = iters < 1 ? 1 : (iters > 4 ? 4 : iters);
    const int64_t rows_per_block = (int64_t)(kBrThreads / 64) * RW * iters;
    const size_t lds = (size_t)(kBrThreads / 64) * RW * (A.S + 1) * sizeof(float);
    dim3 grid(advchain_blocks(outer, (int)rows_per_block)), block(kBrThreads);
    hipStream_t st = (hipStream_t)stream;
    switch (KG) {
      case 1: hipLaunchKernelGGL(k_band_reduce_rows<1>, grid, block, lds, st, in, in2, out, outer, A, scale, iters); break;
      case 2: hipLaunchKernelGGL(k_band_reduce_rows<2>, grid, block, lds, st, in, in2, out, outer, A, scale, iters); break;
      case 4: hipLaunchKernelGGL(k_band_reduce_rows<4>, grid, block, lds, st, in, in2, out, outer, A, scale, iters); break;
      case 8: hipLaunchKernelGGL(k_band_reduce_rows<8>, grid, block, lds, st, in, in2, out, outer, A, scale, iters); break;
      default: hipLaunchKernelGGL(k_band_reduce_rows<16>, grid, block, lds, st, in, in2, out, outer, A, scale, iters); break;
    }
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  static const bool no_slab = getenv("ADVCHAIN_NO_BAND_SLAB") != nullptr;   // A/B knob: one thread per output, taps from global memory
  // (small problems only -- one thread per output leaves cfg-2's y pass with 128 workgroups walking 40 dependent taps:
  // 14.4 -> 8.3 us; with 3 k+ workgroups, the 3D passes, the per-thread kernel is 2-4 % faster)
  if (!no_slab && total <= 131072 && inner > 1 && inner % 4 == 0 && aligned && outer < (1ll << 31) && (int64_t)A.S * 5 <= kBrSlabFloats) {
    int IC = (int)inner;                       // inner columns per workgroup: the slab S x (IC + 1) within 48 KiB
    while ((int64_t)A.S * (IC + 1) > kBrSlabFloats) IC = (IC / 2 + 3) / 4 * 4;
    const int chunks = (int)((inner + IC - 1) / IC);
    if (chunks <= 65535) {
      const size_t lds = (size_t)A.S * (IC + 1) * sizeof(float);
      hipLaunchKernelGGL(k_band_reduce_axis_slab, dim3((unsigned)outer, (unsigned)chunks), dim3(kBlock), lds, (hipStream_t)stream, in, in2,
                         out, (int)inner, IC, A, scale);
      ADVCHAIN_LAUNCH_CHECK();
      return ADVCHAIN_OK;
    }
  }
  hipLaunchKernelGGL(k_band_reduce_axis, dim3(advchain_blocks(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, in,
                     in2, out, outer, (int)inner, T.a[axis], scale);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// The innermost-axis adjoint pass with the caller's densified bands: wd[g][WB] (weight of input lo[k] + j for coefficient
// k, zero beyond the band) and lo[g].  in: (rows, S) [- in2] -> out: (rows, g).  ADVCHAIN_ERR_UNSUPPORTED for shapes the
// kernel does not take (the caller then uses advchain_band_reduce_axis): S % 4, S <= 1024, g <= 64, g * WB <= 4096,
// 16-byte aligned inputs.
int advchain_band_reduce_rows_dense(const float* in, const float* in2, float* out, const float* wd, const int32_t* lo,
                                    int64_t rows, int64_t S, int64_t g, int64_t WB, float scale, void* stream) {
  ADVCHAIN_CHECK_ARG(in && out && wd && lo, "band_reduce_rows_dense: null pointer");
  ADVCHAIN_CHECK_ARG(rows >= 0 && S >= 1 && g >= 1 && WB >= 1, "band_reduce_rows_dense: bad shape");
  if (rows == 0) return ADVCHAIN_OK;
  const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
  if (S % 4 != 0 || S > 1024 || g > 64 || g * WB > kBrMaxW || !aligned) return ADVCHAIN_ERR_UNSUPPORTED;
  // coefficient groups per wave KG (a wave holds RW = 64 / KG rows): 4 is conflict-free for a 16x upsampling; fewer rows a
  // wave (KG = 8, 16) until there are ~4096 waves and RW * S / 4 <= 512 float4 per wave and iteration
  int KG = g >= 4 ? 4 : (g >= 2 ? 2 : 1);
  while (KG < 16 && ((64 / KG) * (S / 4) > 512 || rows / (64 / KG) < 4096)) KG *= 2;      // (KG may exceed g: idle lanes in the dot phase)
  const int RW = 64 / KG;
  if (RW * (S / 4) > 512) return ADVCHAIN_ERR_UNSUPPORTED;
  int iters = (int)(rows / ((int64_t)RW * (kBrThreads / 64) * 2048));      // ~2048 workgroups at least
  iters = iters < 1 ? 1 : (iters > 4 ? 4 : iters);
  const int64_t rows_per_block = (int64_t)(kBrThreads / 64) * RW * iters;
  const size_t lds = ((size_t)(kBrThreads / 64) * RW * (S + 1) + (size_t)(g * WB)) * sizeof(float);
  dim3 grid(advchain_blocks(rows, (int)rows_per_block)), block(kBrThreads);
  hipStream_t st = (hipStream_t)stream;
#define BRD_GO(KG_) hipLaunchKernelGGL(k_band_reduce_rows_dense<KG_>, grid, block, lds, st, in, in2, out, rows, wd, lo, (int)S, (int)g, (int)WB, scale, iters)
  switch (KG) {
    case 1: BRD_GO(1); break;
    case 2: BRD_GO(2); break;
    case 4: BRD_GO(4); break;
    case 8: BRD_GO(8); break;
    default: BRD_GO(16); break;
  }
#undef BRD_GO
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_bias_field_fwd(const float* cp, const float* data, float* out, float* field, const int32_t* itab,
                            const float* ftab, const int64_t* S, const int64_t* g, const int64_t* B, int64_t N,
                            int64_t C, float eps, int use_log, float cp_scale, void* stream) {
  ADVCHAIN_CHECK_ARG(cp && field && itab && ftab && (!data || out), "bias_field_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "bias_field_fwd: bad N/C");
  BandTables T;
  ADVCHAIN_CHECK_ARG(unpack_tables(itab, ftab, S, g, B, T), "bias_field_fwd: bad band tables");
  if (N == 0) return ADVCHAIN_OK;
  Dims full{(int)S[0], (int)S[1], (int)S[2]};
  ADVCHAIN_CHECK_ARG(full.voxels() < (1ll << 31), "bias_field_fwd: volume too large");
  ADVCHAIN_CHECK_ARG(T.a[2].g <= kTpMaxLds, "bias_field_fwd: too many control points per row");
  dim3 grid((unsigned)((full.s1 + kTpChunk - 1) / kTpChunk), (unsigned)full.s0, (unsigned)N);
  if (full.s2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(field)) & 15) == 0)
    hipLaunchKernelGGL(k_bias_fwd<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, cp, data, out, field, T, full, (int)C, eps,
                       use_log, cp_scale);
  else
    hipLaunchKernelGGL(k_bias_fwd<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, cp, data, out, field, T, full, (int)C, eps,
                       use_log, cp_scale);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_bias_field_bwd(const float* cp, const float* data, const float* grad_out, float* grad_L, float* grad_data,
                            const int32_t* itab, const float* ftab, const int64_t* S, const int64_t* g,
                            const int64_t* B, int64_t N, int64_t C, float eps, int use_log, float cp_scale,
                            void* stream) {
  ADVCHAIN_CHECK_ARG(cp && data && grad_out && itab && ftab && (grad_L || grad_data), "bias_field_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && C >= 1, "bias_field_bwd: bad N/C");
  BandTables T;
  ADVCHAIN_CHECK_ARG(unpack_tables(itab, ftab, S, g, B, T), "bias_field_bwd: bad band tables");
  if (N == 0) return ADVCHAIN_OK;
  Dims full{(int)S[0], (int)S[1], (int)S[2]};
  ADVCHAIN_CHECK_ARG(full.voxels() < (1ll << 31), "bias_field_bwd: volume too large");
  ADVCHAIN_CHECK_ARG(T.a[2].g <= kTpMaxLds, "bias_field_bwd: too many control points per row");
  dim3 grid((unsigned)((full.s1 + kTpChunk - 1) / kTpChunk), (unsigned)full.s0, (unsigned)N);
  if (full.s2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(grad_out) |
                            reinterpret_cast<uintptr_t>(grad_L) | reinterpret_cast<uintptr_t>(grad_data)) & 15) == 0)
    hipLaunchKernelGGL(k_bias_bwd<4>, grid, dim3(kBlock), 0, (hipStream_t)stream, cp, data, grad_out, grad_L, grad_data, T,
                       full, (int)C, eps, use_log, cp_scale);
  else
    hipLaunchKernelGGL(k_bias_bwd<1>, grid, dim3(kBlock), 0, (hipStream_t)stream, cp, data, grad_out, grad_L, grad_data, T,
                       full, (int)C, eps, use_log, cp_scale);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// ---------------------------------------------------------------------------------------------
// Any odd window (sigma outside the 9-tap range of the kernels above: adv_morph.py:393-398 sizes the window as
// 2 * int(4 sigma + 0.5) + 1).  The reference never leaves sigma = 1; this is the plain per-voxel form -- one output per
// thread, taps in ascending order, zero padding -- for the callers that do.
// ---------------------------------------------------------------------------------------------
constexpr int kGenericMaxTaps = 129;
struct GaussWN { float w[kGenericMaxTaps]; };

namespace advchain {
__global__ void __launch_bounds__(kBlock) k_gauss_axis_generic(const float* __restrict__ in, float* __restrict__ out, int64_t total,
                                                               Dims d, int axis, GaussWN gw, int ntaps, float scale) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int V = (int)d.voxels();
  const int v = (int)(i % V);
  const int x = v % d.s2, y = (v / d.s2) % d.s1, z = v / (d.s2 * d.s1);
  const int S = axis == 2 ? d.s2 : (axis == 1 ? d.s1 : d.s0);
  const int c = axis == 2 ? x : (axis == 1 ? y : z);
  const int stride = axis == 2 ? 1 : (axis == 1 ? d.s2 : d.s1 * d.s2);
  const int R = ntaps / 2;
  const float* p = in + (i - (int64_t)c * stride);
  float acc = 0.f;
  for (int t = 0; t < ntaps; ++t) {
    const int q = c + t - R;
    if (q >= 0 && q < S) acc = fmaf(gw.w[t], p[(int64_t)q * stride] * scale, acc);
  }
  out[i] = acc;
}
}  // namespace advchain

int advchain_gauss_axis_generic(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims, int axis,
                                const float* weights, int ntaps, float scale, void* stream) {
  ADVCHAIN_CHECK_ARG(in && out && in != out && weights, "gauss_axis_generic: null/aliased pointer");
  ADVCHAIN_CHECK_ARG(fdims_ok(ndim, dims), "gauss_axis_generic: bad dims");
  ADVCHAIN_CHECK_ARG(axis >= 0 && axis < 3, "gauss_axis_generic: bad axis");
  ADVCHAIN_CHECK_ARG(ntaps >= 1 && ntaps <= kGenericMaxTaps && (ntaps & 1) == 1, "gauss_axis_generic: the window must have 1..129 taps, odd");
  const Dims d = fmake_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "gauss_axis_generic: volume too large");
  const int64_t total = planes * d.voxels();
  if (total == 0) return ADVCHAIN_OK;
  GaussWN gw;
  for (int k = 0; k < ntaps; ++k) gw.w[k] = weights[k];
  hipLaunchKernelGGL(advchain::k_gauss_axis_generic, dim3(advchain_blocks(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, in, out,
                     total, d, axis, gw, ntaps, scale);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// weights: 9 taps (sigma = 1 -> radius 4).  axis is 0..2 of the (s0,s1,s2) view (2D: s0 == 1).
int advchain_gauss_axis(const float* in, float* out, const float* aux, int64_t planes, int64_t C, int ndim,
                        const int64_t* dims, int axis, const float* weights9, int pre, int post, float scale,
                        void* stream) {
  ADVCHAIN_CHECK_ARG(in && out && in != out && weights9, "gauss_axis: null/aliased pointer");
  ADVCHAIN_CHECK_ARG(fdims_ok(ndim, dims), "gauss_axis: bad dims");
  ADVCHAIN_CHECK_ARG(axis >= 0 && axis < 3 && C >= 1 && C <= 3, "gauss_axis: bad axis/C");
  ADVCHAIN_CHECK_ARG(pre >= 0 && pre <= 2 && post >= 0 && post <= 2 && (post != 2 || aux), "gauss_axis: bad pre/post");
  const Dims d = fmake_dims(ndim, dims);
  const int64_t total = planes * d.voxels();
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "gauss_axis: volume too large");
  if (total == 0) return ADVCHAIN_OK;
  GaussW gw;
  for (int k = 0; k < 9; ++k) gw.w[k] = weights9[k];
  hipStream_t st = (hipStream_t)stream;
  dim3 blk(kBlock);
  const bool v4 = (d.s2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
                                       reinterpret_cast<uintptr_t>(aux)) & 15) == 0;
  dim3 grid(advchain_blocks(v4 ? total / 4 : total, kBlock));
  int riw = (v4 && d.s2 >= 4 && 64 % (d.s2 / 4) == 0 && (total / 4) % 64 == 0) ? 1 : 0;   // x rows aligned to waves
  static const bool no_riw2 = false, no_zmarch = getenv("ADVCHAIN_NO_GAUSS_ZMARCH") != nullptr;   // A/B knob
  if (!riw && v4 && axis == 2 && d.s2 >= 8 && d.s2 / 4 <= 64 && !no_riw2) {   // rows that do not divide a wave: whole rows per wave
    riw = 2;
    const int64_t rows = total / d.s2, rpw = 64 / (d.s2 / 4);
    grid = dim3(advchain_blocks(((rows + rpw - 1) / rpw) * 64, kBlock));
  }
  if (v4 && axis == 0 && d.s0 >= 32 && !no_zmarch) {   // z pass of a volume: marching form
    const int zc = 27;                                   // 3 rounds of 9 outputs: 1.3 loads per output
    const int nchunk = (d.s0 + zc - 1) / zc;
    const int64_t threads = planes * nchunk * d.s1 * (d.s2 / 4);
    dim3 gm(advchain_blocks(threads, kBlock));
#define GM(PRE, POST) hipLaunchKernelGGL((k_gauss_march_z<PRE, POST>), gm, blk, 0, st, in, out, aux, planes, d, (int)C, gw, scale, zc, nchunk)
    switch (pre * 3 + post) {
      case 0: GM(0, 0); break;
      case 1: GM(0, 1); break;
      case 2: GM(0, 2); break;
      case 3: GM(1, 0); break;
      case 4: GM(1, 1); break;
      case 5: GM(1, 2); break;
      case 6: GM(2, 0); break;
      case 7: GM(2, 1); break;
      default: GM(2, 2); break;
    }
#undef GM
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
#define GA(PRE, POST)                                                                                                     \
  do {                                                                                                                    \
    if (!v4) hipLaunchKernelGGL((k_gauss_axis<PRE, POST>), grid, blk, 0, st, in, out, aux, total, d, (int)C, axis, gw, scale); \
    else if (axis == 2) hipLaunchKernelGGL((k_gauss_axis_v4<PRE, POST, 2>), grid, blk, 0, st, in, out, aux, total / 4, d, (int)C, gw, scale, riw); \
    else if (axis == 1) hipLaunchKernelGGL((k_gauss_axis_v4<PRE, POST, 1>), grid, blk, 0, st, in, out, aux, total / 4, d, (int)C, gw, scale, 0); \
    else hipLaunchKernelGGL((k_gauss_axis_v4<PRE, POST, 0>), grid, blk, 0, st, in, out, aux, total / 4, d, (int)C, gw, scale, 0); \
  } while (0)
  switch (pre * 3 + post) {
    case 0: GA(0, 0); break;
    case 1: GA(0, 1); break;
    case 2: GA(0, 2); break;
    case 3: GA(1, 0); break;
    case 4: GA(1, 1); break;
    case 5: GA(1, 2); break;
    case 6: GA(2, 0); break;
    case 7: GA(2, 1); break;
    default: GA(2, 2); break;
  }
#undef GA
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

}  // extern "C"

// The shapes advchain_gauss_xy takes (pointer alignment aside): for callers that must know before they enqueue anything
// (the composite entries of demons_compose.cpp).
bool advchain_gauss_xy_takes(int ndim, const int64_t* dims, int64_t planes) {
  static const bool off = getenv("ADVCHAIN_NO_GAUSS_XY") != nullptr;   // A/B knob (the one advchain_gauss_xy reads)
  if (off || !fdims_ok(ndim, dims)) return false;
  const Dims d = fmake_dims(ndim, dims);
  if ((d.s2 & 3) != 0 || d.s2 < 8 || d.s2 > 512 || d.s1 < 8 || planes > 65535 || d.s0 > 65535) return false;
  int TY = 32;
  while (TY > 8 && (size_t)(TY + 8) * (2 * d.s2 + 8) * 4 > 53248) TY >>= 1;
  if (TY > d.s1) TY = (d.s1 + 7) / 8 * 8;
  return (size_t)(TY + 8) * (2 * d.s2 + 8) * sizeof(float) <= 65536;
}

extern "C" {

// x and y passes of the separable Gaussian in one launch (rows of 4k <= 512 voxels, 16-byte aligned tensors).  `post` is
// only legal when y is the last axis (ndim == 2).  Returns ADVCHAIN_ERR_UNSUPPORTED when the shape does not qualify: the
// caller then runs the per-axis passes.
int advchain_gauss_xy(const float* in, float* out, const float* aux, int64_t planes, int64_t C, int ndim, const int64_t* dims,
                      const float* weights9, int pre, int post, float scale, void* stream, const float* in_hi, int64_t planes_lo) {
  ADVCHAIN_CHECK_ARG(in && out && in != out && weights9, "gauss_xy: null/aliased pointer");
  ADVCHAIN_CHECK_ARG(fdims_ok(ndim, dims), "gauss_xy: bad dims");
  ADVCHAIN_CHECK_ARG(C >= 1 && C <= 3 && pre >= 0 && pre <= 2 && post >= 0 && post <= 2 && (post != 2 || aux), "gauss_xy: bad C/pre/post");
  ADVCHAIN_CHECK_ARG(post == 0 || ndim == 2, "gauss_xy: post belongs to the last axis");
  const Dims d = fmake_dims(ndim, dims);
  static const bool off = getenv("ADVCHAIN_NO_GAUSS_XY") != nullptr;   // A/B knob
  if (off || (d.s2 & 3) != 0 || d.s2 < 8 || d.s2 > 512 || d.s1 < 8 || planes > 65535 || d.s0 > 65535 ||
      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(aux) |
        reinterpret_cast<uintptr_t>(in_hi)) & 15) != 0)
    return ADVCHAIN_ERR_UNSUPPORTED;
  ADVCHAIN_CHECK_ARG(!in_hi || (planes_lo > 0 && planes_lo < planes), "gauss_xy: bad split of the input planes");
  if (planes == 0 || d.voxels() == 0) return ADVCHAIN_OK;
  int TY = 32;
  while (TY > 8 && (size_t)(TY + 8) * (2 * d.s2 + 8) * 4 > 53248) TY >>= 1;     // 52 KiB: three workgroups a CU
  if (TY > d.s1) TY = (d.s1 + 7) / 8 * 8;
  const size_t lds = (size_t)(TY + 8) * (2 * d.s2 + 8) * sizeof(float);
  if (lds > 65536) return ADVCHAIN_ERR_UNSUPPORTED;
  GaussW gw;
  for (int k = 0; k < 9; ++k) gw.w[k] = weights9[k];
  dim3 grid((unsigned)((d.s1 + TY - 1) / TY), (unsigned)d.s0, (unsigned)planes);
  hipStream_t st = (hipStream_t)stream;
#define GXY(PRE, POST) hipLaunchKernelGGL((k_gauss_xy<PRE, POST>), grid, dim3(kBlock), lds, st, in, out, aux, d, (int)C, gw, scale, TY, in_hi, (int)planes_lo)
  switch (pre * 3 + post) {
    case 0: GXY(0, 0); break;
    case 1: GXY(0, 1); break;
    case 2: GXY(0, 2); break;
    case 3: GXY(1, 0); break;
    case 4: GXY(1, 1); break;
    case 5: GXY(1, 2); break;
    case 6: GXY(2, 0); break;
    case 7: GXY(2, 1); break;
    default: GXY(2, 2); break;
  }
#undef GXY
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// All axes of a small plane in one launch (planes of at most 4096 voxels); pre = 0 | 1, no epilogue.
static int gauss_small_launch(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims,
                              const float* weights9, int pre, float scale, int mirror, void* stream) {
  ADVCHAIN_CHECK_ARG(in && out && in != out && weights9, "gauss_small: null/aliased pointer");
  ADVCHAIN_CHECK_ARG(fdims_ok(ndim, dims), "gauss_small: bad dims");
  ADVCHAIN_CHECK_ARG(pre == 0 || pre == 1, "gauss_small: pre must be 0 or 1");
  const Dims d = fmake_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() <= kGaussSmallMax, "gauss_small: plane larger than 4096 voxels");
  ADVCHAIN_CHECK_ARG(planes >= 0 && planes < (1ll << 30), "gauss_small: bad plane count");
  if (planes == 0) return ADVCHAIN_OK;
  GaussW gw;
  for (int k = 0; k < 9; ++k) gw.w[k] = weights9[k];
  const unsigned blocks = (unsigned)(mirror == 1 ? 2 * planes : planes);
  if (d.voxels() > 1024)
    hipLaunchKernelGGL(k_gauss_small<1024>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, in, out, d, ndim, gw,
                       pre == 1 ? scale : 1.f, mirror, (int)planes);
  else
    hipLaunchKernelGGL(k_gauss_small<kBlock>, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, in, out, d, ndim, gw,
                       pre == 1 ? scale : 1.f, mirror, (int)planes);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_gauss_small(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims,
                         const float* weights9, int pre, float scale, void* stream) {
  return gauss_small_launch(in, out, planes, ndim, dims, weights9, pre, scale, 0, stream);
}

int advchain_gauss_small_pair(const float* in, float* out, int64_t planes, int ndim, const int64_t* dims,
                              const float* weights9, float scale, int adjoint, void* stream) {
  return gauss_small_launch(in, out, planes, ndim, dims, weights9, 1, scale, adjoint ? 2 : 1, stream);
}

int advchain_sign_axpy(const float* base, const float* x, float* out, float a, int64_t n, const float* gate, const float* old,
                       void* stream) {
  ADVCHAIN_CHECK_ARG(x && out && n >= 0 && (!gate || old), "sign_axpy: null pointer");
  if (n == 0) return ADVCHAIN_OK;
  hipLaunchKernelGGL(k_sign_axpy, dim3(advchain_blocks(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, base, x, out, a, n, gate, old);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_nonzero_mask(const float* x, float* out, int64_t n, void* stream) {
  ADVCHAIN_CHECK_ARG(x && out && n >= 0, "nonzero_mask: null pointer");
  if (n == 0) return ADVCHAIN_OK;
  hipLaunchKernelGGL(k_nonzero_mask, dim3(advchain_blocks(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, out, n);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// out = (x ? x : 0) + a * y
int advchain_axpy(const float* x, const float* y, float* out, float a, int64_t n, void* stream) {
  ADVCHAIN_CHECK_ARG(y && out && n >= 0, "axpy: null pointer");
  if (n == 0) return ADVCHAIN_OK;
  const bool al = ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  const int64_t threads = n4 > (n - 4 * n4) ? n4 : (n - 4 * n4);
  hipLaunchKernelGGL(k_axpy, dim3(advchain_blocks(threads, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, y, out, a, n4, n);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int64_t advchain_norm_workspace(int64_t N, int64_t M) {
  const int64_t chunk = 4096;   // (16384 left a 32 x 65536 parameter with 128 workgroups on 256 CUs)
  return N * ((M + chunk - 1) / chunk);  // floats
}

// out[n] = (base ? base[n] : 0) + step * x[n] / (||x[n]||_2 + 1e-20)       x: (N, M); gate / old: see k_norm_axpy
int advchain_norm_axpy_gated(const float* base, const float* x, float* out, float* workspace, float step, int64_t N,
                             int64_t M, const float* gate, const float* old, void* stream) {
  ADVCHAIN_CHECK_ARG(x && out && workspace && (!gate || old), "norm_axpy: null pointer");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && M >= 0, "norm_axpy: bad N/M");
  if (N == 0 || M == 0) return ADVCHAIN_OK;
  const int chunk = 4096;
  const int nb = (int)((M + chunk - 1) / chunk);
  dim3 grid(nb, (unsigned)N), blk(kBlock);
  if (M <= 16384) {      // low-resolution parameters: one workgroup per sample does both steps in one launch
    hipLaunchKernelGGL(k_norm_axpy_row, dim3((unsigned)N), blk, 0, (hipStream_t)stream, base, x, step, out, M, gate, old);
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  hipLaunchKernelGGL(k_sumsq_partial, grid, blk, 0, (hipStream_t)stream, x, workspace, M, chunk);
  hipLaunchKernelGGL(k_norm_axpy, grid, blk, 0, (hipStream_t)stream, base, x, workspace, nb, step, out, M, chunk, gate, old);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_norm_axpy(const float* base, const float* x, float* out, float* workspace, float step, int64_t N,
                       int64_t M, void* stream) {
  return advchain_norm_axpy_gated(base, x, out, workspace, step, N, M, nullptr, nullptr, stream);
}

int advchain_update_multi(const advchain_update_desc* descs, int n, const float* gate, void* stream) {
  ADVCHAIN_CHECK_ARG(descs && n >= 1 && n <= kUpdMax, "update_multi: 1..8 descriptors");
  UpdPack p;
  p.n = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const advchain_update_desc& d = descs[i];
    ADVCHAIN_CHECK_ARG(d.x && d.out && (!gate || d.old), "update_multi: null pointer");
    ADVCHAIN_CHECK_ARG(d.N >= 0 && d.N < 65536 && d.M >= 0 && (d.kind == 0 || d.kind == 1), "update_multi: bad N / M / kind");
    if (d.N == 0 || d.M == 0) continue;
    UpdDesc& u = p.d[p.n++];
    u.base = d.base; u.x = d.x; u.out = d.out; u.old = d.old;
    u.M = d.M; u.rows = (int)d.N; u.kind = d.kind; u.block0 = blocks; u.step = d.step;
    blocks += (int)d.N;
  }
  if (blocks == 0) return ADVCHAIN_OK;
  hipLaunchKernelGGL(k_update_multi, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, p, gate);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

}  // extern "C"
