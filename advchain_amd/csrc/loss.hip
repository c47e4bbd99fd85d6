// Fused segmentation-consistency loss ('mse' + 'contour' terms) for gfx950.
//
//   advchain_consistency_fwd/bwd <- calc_segmentation_consistency + contour_loss,
//                                   advchain/common/loss.py:8-87,102-220 (Q13, Q14)
//
// forward : P = softmax(pred), T = softmax(ref) (or ref itself when it is already a probability
//           map), D = P - T;   sums[0] = sum_k ((P_k m_k) - (T_k m_k))^2
//           edge stencils A, B on D (classes 1..K-1), masked by m_0:
//             2D: A = Sobel-x, B = Sobel-y;  3D (Q14): A = h (x) hp (x) h  (used for BOTH conv_x and
//             conv_y in the reference), B = h (x) h (x) hp;  sums[1] = sum (A*D m)^2, sums[2] = sum (B*D m)^2
//           R = 2 m^2 (A*D), 2 m^2 (B*D)   (kept for the backward)
// backward: grad_pred = softmax'(P) [ gs * ( c_mse 2 m^2 D + c_a A^T R_A + c_b B^T R_B ) ]
// 'kl' term (loss.py:223-249), folded into the same kernels: with T' = softmax(ref) (is_gt: where(ref == 0, 1e-8, 1 - 1e-8)),
//           sums[3] = sum_k m_k T'_k (log T'_k - log P_k)   (log-softmax from the max / log-sum-exp already in registers);
//           its gradient w.r.t. the logits is added behind the softmax Jacobian:  gs c_kl (P_j sum_k m_k T'_k - m_j T'_j),
//           T' recovered from the saved P and D (T = P - D).
// The caller owns the normalisers (global N under batch sharding).
// Streaming + 3^d stencil: HBM-bound, no MFMA.
#include <stdlib.h>
#include "common.h"

// exp of a softmax argument x - max <= 0: the library expf (13 VALU instructions: extended-precision argument reduction,
// ldexp, range selects) or v_exp_f32(x log2 e) (2; its argument product is rounded once: |x| 6e-8 relative, at most 2.2e-8
// absolute in a probability, against the 6e-8 of a correctly rounded exp).  A/B of tools/sessions/r06_s18.sh
#ifndef ADVCHAIN_SOFTMAX_FAST_EXP
#define ADVCHAIN_SOFTMAX_FAST_EXP 1
#endif
#if ADVCHAIN_SOFTMAX_FAST_EXP
#define ADVCHAIN_SM_EXP(x) __expf(x)
#else
#define ADVCHAIN_SM_EXP(x) expf(x)
#endif

namespace advchain {

constexpr int kMaxK = 16;

__device__ __forceinline__ float hsm(int i) { return i == 1 ? 2.f : 1.f; }        // [1, 2, 1]
__device__ __forceinline__ float hdf(int i) { return i == 0 ? 1.f : (i == 1 ? 0.f : -1.f); }  // [1, 0, -1]

// stencil weights at tap (a0,a1,a2) in {0,1,2}^3 (a0 unused in 2D)
template <int DIM>
__device__ __forceinline__ void stencil_w(int a0, int a1, int a2, float& wa, float& wb) {
  if (DIM == 2) {
    // conv2d cross-correlation, kernel[a1][a2]: Sobel-x = h[a1]*hp[a2], Sobel-y = hp[a1]*h[a2]
    wa = hsm(a1) * hdf(a2);
    wb = hdf(a1) * hsm(a2);
  } else {
    wa = hsm(a0) * hdf(a1) * hsm(a2);
    wb = hsm(a0) * hsm(a1) * hdf(a2);
  }
}

// 'kl' (loss.py:239-248): m * p * (log p - log q).  is_gt: p = where(ref == 0, 1e-8, 1 - 1e-8) (= 1.0f in fp32), log p = log(p)
__device__ __forceinline__ float kl_prob(float t, int is_gt) { return is_gt ? (t == 0.f ? 1e-8f : 1.f) : t; }
__device__ __forceinline__ float kl_term(float t, float log_t, float log_q, float m, int is_gt) {
  const float p = kl_prob(t, is_gt);
  const float lp = is_gt ? logf(p) : log_t;
  return m * (p * lp) - m * (p * log_q);
}

__global__ void __launch_bounds__(kBlock)
k_softmax_diff(const float* __restrict__ pred, const float* __restrict__ ref, const float* __restrict__ mask,
               float* __restrict__ P, float* __restrict__ D, float* __restrict__ sums, int K, int V, int mask_ch,
               int ref_is_prob, int want_kl) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int v = blockIdx.x * kBlock + threadIdx.x;
  float acc[2] = {0.f, 0.f};
  if (v < V) {
    const float* pn = pred + (int64_t)n * K * V + v;
    const float* rn = ref + (int64_t)n * K * V + v;
    float mp = -INFINITY, mr = -INFINITY;
    for (int k = 0; k < K; ++k) {
      mp = fmaxf(mp, pn[(int64_t)k * V]);
      mr = fmaxf(mr, rn[(int64_t)k * V]);
    }
    float sp = 0.f, sr = 0.f;
    for (int k = 0; k < K; ++k) {
      sp += ADVCHAIN_SM_EXP(pn[(int64_t)k * V] - mp);
      sr += ADVCHAIN_SM_EXP(rn[(int64_t)k * V] - mr);
    }
    const float lsp = logf(sp), lsr = logf(sr);
    const float isp = 1.f / sp, isr = 1.f / sr;     // ONE division per voxel and side, K products (see softmax_quads)
    for (int k = 0; k < K; ++k) {
      const float zp = pn[(int64_t)k * V] - mp, zr = rn[(int64_t)k * V] - mr;
      const float p = mul_nc(ADVCHAIN_SM_EXP(zp), isp);
      const float t = ref_is_prob ? rn[(int64_t)k * V] : mul_nc(ADVCHAIN_SM_EXP(zr), isr);
      const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
      const int64_t o = ((int64_t)n * K + k) * V + v;
      P[o] = p;
      D[o] = p - t;
      const float e = p * m - t * m;
      acc[0] += e * e;
      if (want_kl) acc[1] += kl_term(t, zr - lsr, zp - lsp, m, ref_is_prob);
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + sum_slot(), acc[0]);
    if (want_kl) atomic_add_f32(sums + 3 * kSumSlots + sum_slot(), acc[1]);
  }
}

// K known at compile time, V % 4 == 0: 4 voxels per thread with 16-byte loads and stores, every input read once
// (the generic kernel above re-reads its 2K inputs three times with 4-byte loads: 36 vector-memory instructions per
// voxel-wave at K = 4, against 4.25 here -- these streaming kernels are bound by that count).
template <int K>
__global__ void __launch_bounds__(kBlock)
k_softmax_diff_v4(const float* __restrict__ pred, const float* __restrict__ ref, const float* __restrict__ mask,
                  float* __restrict__ P, float* __restrict__ D, float* __restrict__ sums, int V, int mask_ch,
                  int ref_is_prob, int want_kl) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int v = (blockIdx.x * kBlock + threadIdx.x) * 4;
  float acc[2] = {0.f, 0.f};
  if (v < V) {
    float p[K][4], r[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(pred + ((int64_t)n * K + k) * V + v);
      const float4 b = *reinterpret_cast<const float4*>(ref + ((int64_t)n * K + k) * V + v);
      p[k][0] = a.x; p[k][1] = a.y; p[k][2] = a.z; p[k][3] = a.w;
      r[k][0] = b.x; r[k][1] = b.y; r[k][2] = b.z; r[k][3] = b.w;
    }
    float m1[4] = {1.f, 1.f, 1.f, 1.f};
    if (mask && mask_ch == 1) {
      const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
      m1[0] = mm.x; m1[1] = mm.y; m1[2] = mm.z; m1[3] = mm.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mp = -INFINITY, mr = -INFINITY;
#pragma unroll
      for (int k = 0; k < K; ++k) { mp = fmaxf(mp, p[k][q]); mr = fmaxf(mr, r[k][q]); }
      float sp = 0.f, sr = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { sp += ADVCHAIN_SM_EXP(p[k][q] - mp); sr += ADVCHAIN_SM_EXP(r[k][q] - mr); }
      const float lsp = want_kl ? logf(sp) : 0.f, lsr = want_kl ? logf(sr) : 0.f;
      const float isp = 1.f / sp, isr = 1.f / sr;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float zp = p[k][q] - mp, zr = r[k][q] - mr;
        const float pp = mul_nc(ADVCHAIN_SM_EXP(zp), isp);
        const float tt = ref_is_prob ? r[k][q] : mul_nc(ADVCHAIN_SM_EXP(zr), isr);
        p[k][q] = pp;
        r[k][q] = pp - tt;          // D
        const float tq = tt;
        float m = m1[q];
        if (mask && mask_ch > 1) m = mask[((int64_t)n * mask_ch + k) * V + v + q];
        const float e = pp * m - tq * m;
        acc[0] += e * e;
        if (want_kl) acc[1] += kl_term(tt, zr - lsr, zp - lsp, m, ref_is_prob);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t o = ((int64_t)n * K + k) * V + v;
      *reinterpret_cast<float4*>(P + o) = make_float4(p[k][0], p[k][1], p[k][2], p[k][3]);
      *reinterpret_cast<float4*>(D + o) = make_float4(r[k][0], r[k][1], r[k][2], r[k][3]);
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + sum_slot(), acc[0]);
    if (want_kl) atomic_add_f32(sums + 3 * kSumSlots + sum_slot(), acc[1]);
  }
}

template <int DIM>
__global__ void __launch_bounds__(kBlock)
k_edge_fwd(const float* __restrict__ D, const float* __restrict__ mask, float* __restrict__ R,
           float* __restrict__ sums, int K, Dims d, int mask_ch) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const int v = blockIdx.x * kBlock + threadIdx.x;
  float acc[2] = {0.f, 0.f};
  if (v < V) {
    const int i2 = v % d.s2;
    const int r = v / d.s2;
    const int i1 = r % d.s1;
    const int i0 = r / d.s1;
    const float m = mask ? mask[(int64_t)n * mask_ch * V + v] : 1.f;
    for (int k = 1; k < K; ++k) {
      const float* Dk = D + ((int64_t)n * K + k) * V;
      float ga = 0.f, gb = 0.f;
#pragma unroll
      for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0)
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
          for (int a2 = 0; a2 < 3; ++a2) {
            // unconditional load from the clamped index, value discarded by a select outside the volume (a load
            // inside the `if` is its own branch + s_waitcnt: 27 serial memory round trips per class)
            const int j0 = i0 + a0 - 1, j1 = i1 + a1 - 1, j2 = i2 + a2 - 1;
            const bool in = j0 >= 0 && j0 < d.s0 && j1 >= 0 && j1 < d.s1 && j2 >= 0 && j2 < d.s2;
            const int c0 = min(max(j0, 0), d.s0 - 1), c1 = min(max(j1, 0), d.s1 - 1), c2 = min(max(j2, 0), d.s2 - 1);
            float wa, wb;
            stencil_w<DIM>(a0, a1, a2, wa, wb);
            const float xv = Dk[(c0 * d.s1 + c1) * d.s2 + c2];
            const float x = in ? xv : 0.f;
            ga += wa * x;
            gb += wb * x;
          }
      const float ea = ga * m, eb = gb * m;
      acc[0] += ea * ea;
      acc[1] += eb * eb;
      if (R) {
        R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V + v] = 2.f * m * m * ga;
        R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1) + 1) * V + v] = 2.f * m * m * gb;
      }
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + kSumSlots + sum_slot(), acc[0]);
    atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[1]);
  }
}

template <int DIM>
__global__ void __launch_bounds__(kBlock)
k_consistency_bwd(const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ R,
                  const float* __restrict__ mask, const float* __restrict__ gscale, float* __restrict__ gpred,
                  float c_mse, float c_a, float c_b, int K, Dims d, int mask_ch, float c_kl, int kl_gt) {
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const int v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;
  const int i2 = v % d.s2;
  const int r = v / d.s2;
  const int i1 = r % d.s1;
  const int i0 = r / d.s1;
  const float gs = gscale ? gscale[0] : 1.f;
  float gp[kMaxK];
  float dot = 0.f, klS = 0.f;
  for (int k = 0; k < K; ++k) {
    const int64_t o = ((int64_t)n * K + k) * V + v;
    const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
    float g = c_mse * 2.f * m * m * D[o];
    if (k >= 1 && R) {
      const float* Ra = R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V;
      const float* Rb = Ra + V;
      float s = 0.f;
#pragma unroll
      for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0)
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1)
#pragma unroll
          for (int a2 = 0; a2 < 3; ++a2) {
            // adjoint: contribution of output voxel u - delta, delta = a - 1
            const int j0 = i0 - (a0 - 1), j1 = i1 - (a1 - 1), j2 = i2 - (a2 - 1);
            const bool in = j0 >= 0 && j0 < d.s0 && j1 >= 0 && j1 < d.s1 && j2 >= 0 && j2 < d.s2;
            const int c0 = min(max(j0, 0), d.s0 - 1), c1 = min(max(j1, 0), d.s1 - 1), c2 = min(max(j2, 0), d.s2 - 1);
            float wa, wb;
            stencil_w<DIM>(a0, a1, a2, wa, wb);
            const int q = (c0 * d.s1 + c1) * d.s2 + c2;          // unconditional loads, see k_edge_fwd
            const float ra = Ra[q], rb = Rb[q];
            s += in ? c_a * wa * ra + c_b * wb * rb : 0.f;
          }
      g += s;
    }
    g *= gs;
    gp[k] = g;
    dot += g * P[o];
    if (c_kl != 0.f) klS += m * kl_prob(P[o] - D[o], kl_gt);
  }
  for (int k = 0; k < K; ++k) {
    const int64_t o = ((int64_t)n * K + k) * V + v;
    float g = P[o] * (gp[k] - dot);
    if (c_kl != 0.f) {   // 'kl': gs c_kl (P_j sum_k m_k T'_k - m_j T'_j)
      const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
      g += gs * c_kl * (P[o] * klS - m * kl_prob(P[o] - D[o], kl_gt));
    }
    gpred[o] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// Row formulation of the 3^d stencils for S2 % 64 == 0 (lane <-> x, a wave never straddles a row): each of
// the 3^(d-1) neighbouring rows is loaded ONCE (coalesced) and its x-1 / x+1 values come from whole-wave DPP
// shifts, so a voxel costs 3^(d-1) loads per channel instead of 3^d gathers through L1 (which bounded the
// per-voxel version: 27-tap edge_fwd 200 us, 2x27-tap consistency_bwd 400 us at 4x4x128x128x64).
// ---------------------------------------------------------------------------------------------
struct RowTaps { float c, l, r; };

// value at (j0, j1, x) and its x-neighbours; zero outside the volume (zero padding of conv)
__device__ __forceinline__ RowTaps load_row_taps(const float* __restrict__ p, int j0, int j1, int x, const Dims& d) {
  RowTaps t;
  const bool in = (j0 >= 0) && (j0 < d.s0) && (j1 >= 0) && (j1 < d.s1);
  // unconditional, always-in-range loads (clamped row): they can all be issued before the first use
  const int c0 = min(max(j0, 0), d.s0 - 1), c1 = min(max(j1, 0), d.s1 - 1);
  const float* row = p + ((int64_t)c0 * d.s1 + c1) * d.s2;
  const float c = row[x];
  t.c = in ? c : 0.f;
  const int lane = threadIdx.x & 63;
  t.l = lane_prev_f(t.c);
  t.r = lane_next_f(t.c);
  if (d.s2 > 64) {  // wave seams inside a row
    const float lf = row[max(x - 1, 0)], rf = row[min(x + 1, d.s2 - 1)];
    if (lane == 0) t.l = (in && x > 0) ? lf : 0.f;
    if (lane == 63) t.r = (in && x + 1 < d.s2) ? rf : 0.f;
  }
  return t;
}

template <int DIM>
__global__ void __launch_bounds__(kBlock)
k_edge_fwd_rows(const float* __restrict__ D, const float* __restrict__ mask, float* __restrict__ R,
                float* __restrict__ sums, int K, Dims d, int mask_ch) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const int v = blockIdx.x * kBlock + threadIdx.x;   // V % 64 == 0 and S2 % 64 == 0: all lanes of a wave share a row
  float acc[2] = {0.f, 0.f};
  if (v < V) {
    const int i2 = v % d.s2;
    const int r = v / d.s2;
    const int i1 = r % d.s1;
    const int i0 = r / d.s1;
    const float m = mask ? mask[(int64_t)n * mask_ch * V + v] : 1.f;
    for (int k = 1; k < K; ++k) {
      const float* Dk = D + ((int64_t)n * K + k) * V;
      float ga = 0.f, gb = 0.f;
#pragma unroll
      for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0)
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1) {
          const RowTaps t = load_row_taps(Dk, i0 + a0 - 1, i1 + a1 - 1, i2, d);
          const float sm = t.l + 2.f * t.c + t.r;   // h  along x
          const float df = t.l - t.r;               // hp along x
          if (DIM == 2) {   // A = h[a1] hp[a2],  B = hp[a1] h[a2]
            ga += hsm(a1) * df;
            gb += hdf(a1) * sm;
          } else {          // A = h[a0] hp[a1] h[a2],  B = h[a0] h[a1] hp[a2]
            ga += hsm(a0) * hdf(a1) * sm;
            gb += hsm(a0) * hsm(a1) * df;
          }
        }
      const float ea = ga * m, eb = gb * m;
      acc[0] += ea * ea;
      acc[1] += eb * eb;
      if (R) {
        R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V + v] = 2.f * m * m * ga;
        R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1) + 1) * V + v] = 2.f * m * m * gb;
      }
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + kSumSlots + sum_slot(), acc[0]);
    atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[1]);
  }
}

template <int DIM>
__global__ void __launch_bounds__(kBlock)
k_consistency_bwd_rows(const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ R,
                       const float* __restrict__ mask, const float* __restrict__ gscale, float* __restrict__ gpred,
                       float c_mse, float c_a, float c_b, int K, Dims d, int mask_ch, float c_kl, int kl_gt) {
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const int v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= V) return;   // whole waves only (V % 64 == 0)
  const int i2 = v % d.s2;
  const int r = v / d.s2;
  const int i1 = r % d.s1;
  const int i0 = r / d.s1;
  const float gs = gscale ? gscale[0] : 1.f;
  float gp[kMaxK];
  float dot = 0.f, klS = 0.f;
  for (int k = 0; k < K; ++k) {
    const int64_t o = ((int64_t)n * K + k) * V + v;
    const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
    float g = c_mse * 2.f * m * m * D[o];
    if (k >= 1 && R) {
      const float* Ra = R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V;
      const float* Rb = Ra + V;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0)
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1) {
          // adjoint: tap a reads the output voxel u - (a - 1); along x that flips hp: (r - l)
          const int j0 = i0 - (a0 - 1), j1 = i1 - (a1 - 1);
          const RowTaps ta = load_row_taps(Ra, j0, j1, i2, d);
          const RowTaps tb = load_row_taps(Rb, j0, j1, i2, d);
          if (DIM == 2) {
            sa += hsm(a1) * (ta.r - ta.l);
            sb += hdf(a1) * (tb.l + 2.f * tb.c + tb.r);
          } else {
            sa += hsm(a0) * hdf(a1) * (ta.l + 2.f * ta.c + ta.r);
            sb += hsm(a0) * hsm(a1) * (tb.r - tb.l);
          }
        }
      g += c_a * sa + c_b * sb;
    }
    g *= gs;
    gp[k] = g;
    dot += g * P[o];
    if (c_kl != 0.f) klS += m * kl_prob(P[o] - D[o], kl_gt);
  }
  for (int k = 0; k < K; ++k) {
    const int64_t o = ((int64_t)n * K + k) * V + v;
    float g = P[o] * (gp[k] - dot);
    if (c_kl != 0.f) {   // 'kl': gs c_kl (P_j sum_k m_k T'_k - m_j T'_j)
      const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
      g += gs * c_kl * (P[o] * klS - m * kl_prob(P[o] - D[o], kl_gt));
    }
    gpred[o] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// Marching form of the row kernels (K known at compile time, S2 % 64 == 0): a wave owns a strip of kMarch
// consecutive rows (fixed slice i0 and 64-voxel x range) and walks along y.  The stencils are separable,
// A = h(z) hp(y) h(x), B = h(z) h(y) hp(x)  (2D: A = h(y) hp(x), B = hp(y) h(x)), so each new row needs only its own
// 3^(d-2) loads per array: they are folded over z and x into one value per array, and the y taps come from a
// 3-row window kept in registers.  3 (3D) / 1 (2D) loads per row, array and class instead of 9 / 3: the row
// kernels above were bound by their vector-memory instruction count (65 per voxel-wave at K = 4).
// ---------------------------------------------------------------------------------------------
constexpr int kMarch = 8;

// row (i0 +- 1, j1) folded over z (weights h) and over x: zs = h along x, zd = hp along x (l - r)
template <int DIM>
__device__ __forceinline__ void fold_row(const float* __restrict__ p, int i0, int j1, int x, const Dims& d, float& zs,
                                         float& zd) {
  zs = 0.f; zd = 0.f;
#pragma unroll
  for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0) {
    const RowTaps t = load_row_taps(p, i0 + a0 - 1, j1, x, d);
    const float w = DIM == 3 ? hsm(a0) : 1.f;
    zs += w * (t.l + 2.f * t.c + t.r);
    zd += w * (t.l - t.r);
  }
}

__device__ __forceinline__ bool strip_decode(int strip, const Dims& d, int& i0, int& y0, int& x) {
  const int nx = d.s2 >> 6, ny = (d.s1 + kMarch - 1) / kMarch;
  if (strip >= nx * ny * d.s0) return false;
  const int xc = strip % nx;
  const int r = strip / nx;
  y0 = (r % ny) * kMarch;
  i0 = r / ny;
  x = xc * 64 + (threadIdx.x & 63);
  return true;
}

template <int DIM, int K>
__global__ void __launch_bounds__(kBlock)
k_edge_fwd_march(const float* __restrict__ D, const float* __restrict__ mask, float* __restrict__ R,
                 float* __restrict__ sums, Dims d, int mask_ch) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  float acc[2] = {0.f, 0.f};
  int i0, y0, x;
  if (strip_decode(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), d, i0, y0, x)) {
    float zs[K - 1][3], zd[K - 1][3];   // window: rows i1-1, i1, i1+1
#pragma unroll
    for (int k = 1; k < K; ++k) {
      const float* Dk = D + ((int64_t)n * K + k) * V;
      fold_row<DIM>(Dk, i0, y0 - 1, x, d, zs[k - 1][0], zd[k - 1][0]);
      fold_row<DIM>(Dk, i0, y0, x, d, zs[k - 1][1], zd[k - 1][1]);
    }
    const int y1 = min(y0 + kMarch, d.s1);
    for (int i1 = y0; i1 < y1; ++i1) {
      const int v = (i0 * d.s1 + i1) * d.s2 + x;
      const float m = mask ? mask[(int64_t)n * mask_ch * V + v] : 1.f;
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const float* Dk = D + ((int64_t)n * K + k) * V;
        fold_row<DIM>(Dk, i0, i1 + 1, x, d, zs[k - 1][2], zd[k - 1][2]);
        float ga, gb;
        if (DIM == 2) {   // A = h[a1] hp[a2], B = hp[a1] h[a2]
          ga = zd[k - 1][0] + 2.f * zd[k - 1][1] + zd[k - 1][2];
          gb = zs[k - 1][0] - zs[k - 1][2];
        } else {          // A = h[a0] hp[a1] h[a2], B = h[a0] h[a1] hp[a2]
          ga = zs[k - 1][0] - zs[k - 1][2];
          gb = zd[k - 1][0] + 2.f * zd[k - 1][1] + zd[k - 1][2];
        }
        const float ea = ga * m, eb = gb * m;
        acc[0] += ea * ea;
        acc[1] += eb * eb;
        if (R) {
          R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V + v] = 2.f * m * m * ga;
          R[((int64_t)n * 2 * (K - 1) + 2 * (k - 1) + 1) * V + v] = 2.f * m * m * gb;
        }
        zs[k - 1][0] = zs[k - 1][1]; zs[k - 1][1] = zs[k - 1][2];
        zd[k - 1][0] = zd[k - 1][1]; zd[k - 1][1] = zd[k - 1][2];
      }
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + kSumSlots + sum_slot(), acc[0]);
    atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[1]);
  }
}

template <int DIM, int K, bool KL>
__global__ void __launch_bounds__(kBlock)
k_consistency_bwd_march(const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ R,
                        const float* __restrict__ mask, const float* __restrict__ gscale, float* __restrict__ gpred,
                        float c_mse, float c_a, float c_b, Dims d, int mask_ch, float c_kl, int kl_gt) {
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  int i0, y0, x;
  if (!strip_decode(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), d, i0, y0, x)) return;
  const float gs = gscale ? gscale[0] : 1.f;
  // per class and array one folded value per row: A wants h along x in 3D, hp^T = (r - l) in 2D; B the other way
  float za[K - 1][3], zb[K - 1][3];
  auto fold = [&](int k, int j1, float& a, float& b) {
    const float* Ra = R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V;
    float as, ad, bs, bd;
    fold_row<DIM>(Ra, i0, j1, x, d, as, ad);
    fold_row<DIM>(Ra + V, i0, j1, x, d, bs, bd);
    a = DIM == 3 ? as : -ad;
    b = DIM == 3 ? -bd : bs;
  };
  if (R) {
#pragma unroll
    for (int k = 1; k < K; ++k) {
      fold(k, y0 - 1, za[k - 1][0], zb[k - 1][0]);
      fold(k, y0, za[k - 1][1], zb[k - 1][1]);
    }
  }
  const int y1 = min(y0 + kMarch, d.s1);
  for (int i1 = y0; i1 < y1; ++i1) {
    const int v = (i0 * d.s1 + i1) * d.s2 + x;
    float gp[K], pk[K], mt[KL ? K : 1];
    float dot = 0.f, klS = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t o = ((int64_t)n * K + k) * V + v;
      const float m = mask ? mask[((int64_t)n * mask_ch + (mask_ch > 1 ? k : 0)) * V + v] : 1.f;
      const float dk = D[o];
      float g = c_mse * 2.f * m * m * dk;
      if (k >= 1 && R) {
        fold(k, i1 + 1, za[k - 1][2], zb[k - 1][2]);
        // adjoint: tap a1 reads row i1 - (a1 - 1)
        float sa, sb;
        if (DIM == 2) {
          sa = za[k - 1][0] + 2.f * za[k - 1][1] + za[k - 1][2];
          sb = zb[k - 1][2] - zb[k - 1][0];
        } else {
          sa = za[k - 1][2] - za[k - 1][0];
          sb = zb[k - 1][0] + 2.f * zb[k - 1][1] + zb[k - 1][2];
        }
        g += c_a * sa + c_b * sb;
        za[k - 1][0] = za[k - 1][1]; za[k - 1][1] = za[k - 1][2];
        zb[k - 1][0] = zb[k - 1][1]; zb[k - 1][1] = zb[k - 1][2];
      }
      g *= gs;
      gp[k] = g;
      pk[k] = P[o];
      dot += g * pk[k];
      if (KL) { mt[k] = m * kl_prob(pk[k] - dk, kl_gt); klS += mt[k]; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float g = pk[k] * (gp[k] - dot);
      if (KL) g += gs * c_kl * (pk[k] * klS - mt[k]);
      gpred[((int64_t)n * K + k) * V + v] = g;
    }
  }
}

// sums[r] = sum of the 64 partial slots of row r; value = sum_r coef[r] * sums[r]; the slots are zeroed for the next call
__global__ void __launch_bounds__(kBlock)
k_consistency_finish(float* __restrict__ slots, float c0, float c1, float c2, float c3, float* __restrict__ sums,
                     float* __restrict__ value, int reset) {
  __shared__ float row[4];
  const int r = threadIdx.x >> 6, j = threadIdx.x & 63;
  float v = slots[r * kSumSlots + j];
  if (reset) slots[r * kSumSlots + j] = 0.f;
  v = wave_sum(v);
  if (j == 0) { row[r] = v; sums[r] = v; }
  __syncthreads();
  if (threadIdx.x == 0) value[0] = ((c0 * row[0] + c1 * row[1]) + c2 * row[2]) + c3 * row[3];
}

}  // namespace advchain

using namespace advchain;


// ---------------------------------------------------------------------------------------------
// 16-byte form of the marching kernels (S2 % 4 == 0, S2 <= 256): a lane owns 4 consecutive x, a group of S2/4 lanes
// one row, a wave floor(64/(S2/4)) independent strips.  The x neighbours of a quad's ends come from the
// neighbouring lanes (whole-wave DPP shifts, zero across a row end = the zero padding of the convolution).  Same
// arithmetic per voxel as the scalar march; a quarter of its vector-memory instructions.
// ---------------------------------------------------------------------------------------------
struct Quad { float v[4]; };

// Storage type of the K-channel tensors of the fused loss (round 6 experiment, advchain_consistency_fused_*_bf16): fp32 is the
// product; Bf16 halves the bytes of pred / ref / R / grad_pred and keeps every operation in fp32 registers.
struct Bf16 { unsigned short v; };
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const Bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned bf16_rne(float x) {          // round to nearest even; NaN stays NaN
  const unsigned b = __float_as_uint(x);
  return (x != x) ? 0x7fc0u : (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float e) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, e);
}
__device__ __forceinline__ void st4(Bf16* p, float a, float b, float c, float e) {
  *reinterpret_cast<uint2*>(p) = make_uint2(bf16_rne(a) | (bf16_rne(b) << 16), bf16_rne(c) | (bf16_rne(e) << 16));
}

// rows (i0 +- 1, j1) folded over z and x for the 4 voxels of this lane: zs = h along x, zd = hp along x
template <int DIM, typename ST>
__device__ __forceinline__ void fold_row4(const ST* __restrict__ p, int i0, int j1, int x, bool first, bool last,
                                          const Dims& d, Quad& zs, Quad& zd) {
#pragma unroll
  for (int q = 0; q < 4; ++q) { zs.v[q] = 0.f; zd.v[q] = 0.f; }
#pragma unroll
  for (int a0 = (DIM == 3 ? 0 : 1); a0 < (DIM == 3 ? 3 : 2); ++a0) {
    const int j0 = i0 + a0 - 1;
    const bool in = (j0 >= 0) && (j0 < d.s0) && (j1 >= 0) && (j1 < d.s1);
    const int c0 = min(max(j0, 0), d.s0 - 1), c1 = min(max(j1, 0), d.s1 - 1);
    float4 c = ld4(p + ((int64_t)c0 * d.s1 + c1) * d.s2 + x);
    if (!in) c = make_float4(0.f, 0.f, 0.f, 0.f);
    const float pw = lane_prev_f(c.w), nx = lane_next_f(c.x);
    const float l[4] = {first ? 0.f : pw, c.x, c.y, c.z};
    const float r[4] = {c.y, c.z, c.w, last ? 0.f : nx};
    const float cc[4] = {c.x, c.y, c.z, c.w};
    const float w = DIM == 3 ? hsm(a0) : 1.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      zs.v[q] += w * (l[q] + 2.f * cc[q] + r[q]);
      zd.v[q] += w * (l[q] - r[q]);
    }
  }
}

// strip of this lane group; every lane stays active (DPP), `ok` masks the stores
__device__ __forceinline__ void strip_decode4(const Dims& d, int mlen, int& i0, int& y0, int& x, bool& first, bool& last,
                                              bool& ok) {
  const int lpr = d.s2 >> 2;                     // lanes per row
  const int gpw = 64 / lpr;                      // row groups per wave; lanes beyond gpw * lpr idle (but stay active: DPP)
  const int lane = threadIdx.x & 63;
  const int grp = lane / lpr, xq = lane - grp * lpr;
  const int ny = (d.s1 + mlen - 1) / mlen;
  const int strip = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * gpw + grp;
  ok = grp < gpw && strip < ny * d.s0;
  const int sc = ok ? strip : 0;
  y0 = (sc % ny) * mlen;
  i0 = sc / ny;
  x = 4 * xq;
  first = xq == 0;
  last = xq == lpr - 1;
}

template <int DIM, int K>
__global__ void __launch_bounds__(kBlock)
k_edge_fwd_march4(const float* __restrict__ D, const float* __restrict__ mask, float* __restrict__ R,
                  float* __restrict__ sums, Dims d, int mask_ch, int mlen) {
  __shared__ float smem[8];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  float acc[2] = {0.f, 0.f};
  int i0, y0, x;
  bool first, last, ok;
  strip_decode4(d, mlen, i0, y0, x, first, last, ok);
  Quad zs[K - 1][3], zd[K - 1][3];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const float* Dk = D + ((int64_t)n * K + k) * V;
    fold_row4<DIM>(Dk, i0, y0 - 1, x, first, last, d, zs[k - 1][0], zd[k - 1][0]);
    fold_row4<DIM>(Dk, i0, y0, x, first, last, d, zs[k - 1][1], zd[k - 1][1]);
  }
  const int y1 = min(y0 + mlen, d.s1);
  for (int i1 = y0; i1 < y1; ++i1) {
    const int v = (i0 * d.s1 + i1) * d.s2 + x;
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (mask) {
      const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * mask_ch * V + v);
      m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
    }
#pragma unroll
    for (int k = 1; k < K; ++k) {
      const float* Dk = D + ((int64_t)n * K + k) * V;
      fold_row4<DIM>(Dk, i0, i1 + 1, x, first, last, d, zs[k - 1][2], zd[k - 1][2]);
      float ra[4], rb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float ga, gb;
        if (DIM == 2) {
          ga = zd[k - 1][0].v[q] + 2.f * zd[k - 1][1].v[q] + zd[k - 1][2].v[q];
          gb = zs[k - 1][0].v[q] - zs[k - 1][2].v[q];
        } else {
          ga = zs[k - 1][0].v[q] - zs[k - 1][2].v[q];
          gb = zd[k - 1][0].v[q] + 2.f * zd[k - 1][1].v[q] + zd[k - 1][2].v[q];
        }
        const float ea = ga * m[q], eb = gb * m[q];
        if (ok) { acc[0] += ea * ea; acc[1] += eb * eb; }
        ra[q] = 2.f * m[q] * m[q] * ga;
        rb[q] = 2.f * m[q] * m[q] * gb;
      }
      if (R && ok) {
        *reinterpret_cast<float4*>(R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V + v) = make_float4(ra[0], ra[1], ra[2], ra[3]);
        *reinterpret_cast<float4*>(R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1) + 1) * V + v) = make_float4(rb[0], rb[1], rb[2], rb[3]);
      }
      zs[k - 1][0] = zs[k - 1][1]; zs[k - 1][1] = zs[k - 1][2];
      zd[k - 1][0] = zd[k - 1][1]; zd[k - 1][1] = zd[k - 1][2];
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + kSumSlots + sum_slot(), acc[0]);
    atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[1]);
  }
}

template <int DIM, int K, bool KL>
__global__ void __launch_bounds__(kBlock)
k_consistency_bwd_march4(const float* __restrict__ P, const float* __restrict__ D, const float* __restrict__ R,
                         const float* __restrict__ mask, const float* __restrict__ gscale, float* __restrict__ gpred,
                         float c_mse, float c_a, float c_b, Dims d, int mask_ch, int mlen, float c_kl, int kl_gt) {
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  int i0, y0, x;
  bool first, last, ok;
  strip_decode4(d, mlen, i0, y0, x, first, last, ok);
  const float gs = gscale ? gscale[0] : 1.f;
  Quad za[K - 1][3], zb[K - 1][3];
  auto fold = [&](int k, int j1, Quad& a, Quad& b) {
    const float* Ra = R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V;
    Quad as, ad, bs, bd;
    fold_row4<DIM>(Ra, i0, j1, x, first, last, d, as, ad);
    fold_row4<DIM>(Ra + V, i0, j1, x, first, last, d, bs, bd);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a.v[q] = DIM == 3 ? as.v[q] : -ad.v[q];
      b.v[q] = DIM == 3 ? -bd.v[q] : bs.v[q];
    }
  };
  if (R) {
#pragma unroll
    for (int k = 1; k < K; ++k) {
      fold(k, y0 - 1, za[k - 1][0], zb[k - 1][0]);
      fold(k, y0, za[k - 1][1], zb[k - 1][1]);
    }
  }
  const int y1 = min(y0 + mlen, d.s1);
  for (int i1 = y0; i1 < y1; ++i1) {
    const int v = (i0 * d.s1 + i1) * d.s2 + x;
    float gp[K][4], pk[K][4], mt[KL ? K : 1][4];
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, klS[4] = {0.f, 0.f, 0.f, 0.f};
    float m1[4] = {1.f, 1.f, 1.f, 1.f};
    if (mask && mask_ch == 1) {
      const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
      m1[0] = mm.x; m1[1] = mm.y; m1[2] = mm.z; m1[3] = mm.w;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t o = ((int64_t)n * K + k) * V + v;
      float m[4] = {m1[0], m1[1], m1[2], m1[3]};
      if (mask && mask_ch > 1) {
        const float4 mm = *reinterpret_cast<const float4*>(mask + ((int64_t)n * mask_ch + k) * V + v);
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
      }
      const float4 dd = *reinterpret_cast<const float4*>(D + o);
      const float4 pp = *reinterpret_cast<const float4*>(P + o);
      const float dv[4] = {dd.x, dd.y, dd.z, dd.w};
      pk[k][0] = pp.x; pk[k][1] = pp.y; pk[k][2] = pp.z; pk[k][3] = pp.w;
      if (k >= 1 && R) fold(k, i1 + 1, za[k - 1][2], zb[k - 1][2]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float g = c_mse * 2.f * m[q] * m[q] * dv[q];
        if (k >= 1 && R) {
          float sa, sb;
          if (DIM == 2) {
            sa = za[k - 1][0].v[q] + 2.f * za[k - 1][1].v[q] + za[k - 1][2].v[q];
            sb = zb[k - 1][2].v[q] - zb[k - 1][0].v[q];
          } else {
            sa = za[k - 1][2].v[q] - za[k - 1][0].v[q];
            sb = zb[k - 1][0].v[q] + 2.f * zb[k - 1][1].v[q] + zb[k - 1][2].v[q];
          }
          g += c_a * sa + c_b * sb;
        }
        g *= gs;
        gp[k][q] = g;
        dot[q] += g * pk[k][q];
        if (KL) { mt[k][q] = m[q] * kl_prob(pk[k][q] - dv[q], kl_gt); klS[q] += mt[k][q]; }
      }
      if (k >= 1 && R) {
        za[k - 1][0] = za[k - 1][1]; za[k - 1][1] = za[k - 1][2];
        zb[k - 1][0] = zb[k - 1][1]; zb[k - 1][1] = zb[k - 1][2];
      }
    }
    if (ok) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o4[q] = pk[k][q] * (gp[k][q] - dot[q]);
          if (KL) o4[q] += gs * c_kl * (pk[k][q] * klS[q] - mt[k][q]);
        }
        *reinterpret_cast<float4*>(gpred + ((int64_t)n * K + k) * V + v) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// f2, really fused (round 4): the forward keeps NO per-voxel intermediate but the edge responses R the backward's stencil
// adjoint needs -- softmax(pred), softmax(ref), their difference, the masked squared error, the KL sum and the 3^d edge
// stencils all happen in the marching kernel's registers, straight from the logits; the backward recomputes the softmax at
// its own voxel instead of reading P and D back.  Round 3 ran three kernels with P and D (2K channels, written once and
// read twice) in between: 27 channel passes forward + 19 backward at K = 4; now 15 + 19 (9 compulsory reads + 6 R forward),
// and two launches instead of three.  The price is arithmetic nobody waits for: a voxel's softmax is evaluated by every
// strip that reads it as a stencil neighbour (3x in z, (mlen + 2) / mlen in y).
// Same per-voxel arithmetic as k_softmax_diff_v4 / k_edge_fwd_march4 / k_consistency_bwd_march4.
// ---------------------------------------------------------------------------------------------
// softmax of the 4 voxels at `o` (+ k V) of both logit maps: P, T (the reference itself when it already holds probabilities);
// with LOGS also log-softmax terms for the KL sum
// softmax of 4 voxels of both logit maps, from registers: P, T (the reference itself when it already holds probabilities); with
// LOGS also the log-softmax terms for the KL sum
template <int K, bool LOGS>
__device__ __forceinline__ void softmax_quads(const float (&p)[K][4], const float (&r)[K][4], int ref_is_prob, float (&P)[K][4],
                                              float (&T)[K][4], float (&lq)[LOGS ? K : 1][4], float (&lt)[LOGS ? K : 1][4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float mp = -INFINITY, mr = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { mp = fmaxf(mp, p[k][q]); mr = fmaxf(mr, r[k][q]); }
    float sp = 0.f, sr = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { sp += ADVCHAIN_SM_EXP(p[k][q] - mp); sr += ADVCHAIN_SM_EXP(r[k][q] - mr); }
    const float lsp = LOGS ? logf(sp) : 0.f, lsr = LOGS ? logf(sr) : 0.f;
    // ONE correctly rounded division per voxel and side and K products instead of K divisions (round 6: a division is ten
    // VALU instructions, and 41 % of the marching loop's were divisions): P differs from exp / sum by at most one ulp.
    // mul_nc (common.h; __fmul_rn is a plain `*` on this toolchain): the product must be ROUNDED before anything uses it -- contracted into a consumer's fma, `P - (P - T)` of
    // the backward's 'kl' term (kl_prob) is the product's rounding error instead of T = 0 (caught by
    // test_fused_loss_3d_marching_along_z[2-dims0]).  The hardware reciprocal (v_rcp_f32, one ulp) instead of the division
    // measured another 1-4 us per launch and nothing per call (profiles/r06/softmax_rcp/): not taken.
    const float isp = 1.f / sp, isr = 1.f / sr;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float zp = p[k][q] - mp, zr = r[k][q] - mr;
      P[k][q] = mul_nc(ADVCHAIN_SM_EXP(zp), isp);
      T[k][q] = ref_is_prob ? r[k][q] : mul_nc(ADVCHAIN_SM_EXP(zr), isr);
      if (LOGS) { lq[k][q] = zp - lsp; lt[k][q] = zr - lsr; }
    }
  }
}

// the same for the 4 voxels at `o` (+ k V) of both logit maps in memory
template <int K, bool LOGS, typename ST>
__device__ __forceinline__ void softmax_pair4(const ST* __restrict__ pred, const ST* __restrict__ ref, int64_t o, int V,
                                              int ref_is_prob, float (&P)[K][4], float (&T)[K][4], float (&lq)[LOGS ? K : 1][4],
                                              float (&lt)[LOGS ? K : 1][4]) {
  float p[K][4], r[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float4 a = ld4(pred + o + (int64_t)k * V);
    const float4 b = ld4(ref + o + (int64_t)k * V);
    p[k][0] = a.x; p[k][1] = a.y; p[k][2] = a.z; p[k][3] = a.w;
    r[k][0] = b.x; r[k][1] = b.y; r[k][2] = b.z; r[k][3] = b.w;
  }
  softmax_quads<K, LOGS>(p, r, ref_is_prob, P, T, lq, lt);
}

template <int DIM, int K, bool KL, bool EDGES, typename ST = float>
__global__ void __launch_bounds__(kBlock)
k_loss_fused_fwd4(const ST* __restrict__ pred, const ST* __restrict__ ref, const float* __restrict__ mask,
                  ST* __restrict__ R, float* __restrict__ sums, Dims d, int mlen, int ref_is_prob) {
  __shared__ float smem[16];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};          // mse, edge A, edge B, kl
  int i0, y0, x;
  bool first, last, ok;
  strip_decode4(d, mlen, i0, y0, x, first, last, ok);
  const int y1 = min(y0 + mlen, d.s1);
  Quad zs[EDGES ? K - 1 : 1][3], zd[EDGES ? K - 1 : 1][3];
  // row j1 of the planes i0 - 1 .. i0 + 1: softmax difference of its 4 voxels, folded over z and x into window slot `slot`;
  // the voxels of the strip's OWN rows (plane i0) also enter the mse / kl sums
  auto row = [&](int j1, int slot) {
    if (EDGES) {
#pragma unroll
      for (int k = 1; k < K; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) { zs[k - 1][slot].v[q] = 0.f; zd[k - 1][slot].v[q] = 0.f; }
    }
#pragma unroll
    for (int a0 = ((DIM == 3 && EDGES) ? 0 : 1); a0 < ((DIM == 3 && EDGES) ? 3 : 2); ++a0) {
      const int j0 = i0 + a0 - 1;
      const bool in = (j0 >= 0) && (j0 < d.s0) && (j1 >= 0) && (j1 < d.s1);
      const int c0 = min(max(j0, 0), d.s0 - 1), c1 = min(max(j1, 0), d.s1 - 1);
      const int v = (c0 * d.s1 + c1) * d.s2 + x;
      float P[K][4], T[K][4], lq[KL ? K : 1][4], lt[KL ? K : 1][4];
      softmax_pair4<K, KL, ST>(pred, ref, (int64_t)n * K * V + v, V, ref_is_prob, P, T, lq, lt);
      if (a0 == 1 && in && ok && j1 >= y0 && j1 < y1) {
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (mask) {
          const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
          m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float e = P[k][q] * m[q] - T[k][q] * m[q];
            acc[0] += e * e;
            if (KL) acc[3] += kl_term(T[k][q], lt[k][q], lq[k][q], m[q], ref_is_prob);
          }
      }
      if (EDGES) {
        const float w = DIM == 3 ? hsm(a0) : 1.f;
#pragma unroll
        for (int k = 1; k < K; ++k) {
          float c[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) c[q] = in ? P[k][q] - T[k][q] : 0.f;      // D; zero padding outside the volume
          const float pw = lane_prev_f(c[3]), nx = lane_next_f(c[0]);
          const float l[4] = {first ? 0.f : pw, c[0], c[1], c[2]};
          const float r[4] = {c[1], c[2], c[3], last ? 0.f : nx};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            zs[k - 1][slot].v[q] += w * (l[q] + 2.f * c[q] + r[q]);
            zd[k - 1][slot].v[q] += w * (l[q] - r[q]);
          }
        }
      }
    }
  };
  if (EDGES) { row(y0 - 1, 0); }
  row(y0, 1);
  for (int i1 = y0; i1 < y1; ++i1) {
    if (EDGES) {
      row(i1 + 1, 2);
      const int v = (i0 * d.s1 + i1) * d.s2 + x;
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (mask) {
        const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
      }
#pragma unroll
      for (int k = 1; k < K; ++k) {
        float ra[4], rb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float ga, gb;
          if (DIM == 2) {
            ga = zd[k - 1][0].v[q] + 2.f * zd[k - 1][1].v[q] + zd[k - 1][2].v[q];
            gb = zs[k - 1][0].v[q] - zs[k - 1][2].v[q];
          } else {
            ga = zs[k - 1][0].v[q] - zs[k - 1][2].v[q];
            gb = zd[k - 1][0].v[q] + 2.f * zd[k - 1][1].v[q] + zd[k - 1][2].v[q];
          }
          const float ea = ga * m[q], eb = gb * m[q];
          if (ok) { acc[1] += ea * ea; acc[2] += eb * eb; }
          ra[q] = 2.f * m[q] * m[q] * ga;
          rb[q] = 2.f * m[q] * m[q] * gb;
        }
        if (R && ok) {
          st4(R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V + v, ra[0], ra[1], ra[2], ra[3]);
          st4(R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1) + 1) * V + v, rb[0], rb[1], rb[2], rb[3]);
        }
        zs[k - 1][0] = zs[k - 1][1]; zs[k - 1][1] = zs[k - 1][2];
        zd[k - 1][0] = zd[k - 1][1]; zd[k - 1][1] = zd[k - 1][2];
      }
    } else if (i1 + 1 < y1) {
      row(i1 + 1, 1);
    }
  }
  block_sum<4>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + sum_slot(), acc[0]);
    if (EDGES) {
      atomic_add_f32(sums + kSumSlots + sum_slot(), acc[1]);
      atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[2]);
    }
    if (KL) atomic_add_f32(sums + 3 * kSumSlots + sum_slot(), acc[3]);
  }
}

// backward of the fused form: k_consistency_bwd_march4 with P and D = P - T recomputed from the logits at the voxel
template <int DIM, int K, bool KL, typename ST = float>
__global__ void __launch_bounds__(kBlock)
k_loss_fused_bwd4(const ST* __restrict__ pred, const ST* __restrict__ ref, const ST* __restrict__ R,
                  const float* __restrict__ mask, const float* __restrict__ gscale, ST* __restrict__ gpred,
                  float c_mse, float c_a, float c_b, Dims d, int mlen, float c_kl, int ref_is_prob) {
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  int i0, y0, x;
  bool first, last, ok;
  strip_decode4(d, mlen, i0, y0, x, first, last, ok);
  const float gs = gscale ? gscale[0] : 1.f;
  Quad za[K - 1][3], zb[K - 1][3];
  auto fold = [&](int k, int j1, Quad& a, Quad& b) {
    const ST* Ra = R + ((int64_t)n * 2 * (K - 1) + 2 * (k - 1)) * V;
    Quad as, ad, bs, bd;
    fold_row4<DIM>(Ra, i0, j1, x, first, last, d, as, ad);
    fold_row4<DIM>(Ra + V, i0, j1, x, first, last, d, bs, bd);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a.v[q] = DIM == 3 ? as.v[q] : -ad.v[q];
      b.v[q] = DIM == 3 ? -bd.v[q] : bs.v[q];
    }
  };
  if (R) {
#pragma unroll
    for (int k = 1; k < K; ++k) {
      fold(k, y0 - 1, za[k - 1][0], zb[k - 1][0]);
      fold(k, y0, za[k - 1][1], zb[k - 1][1]);
    }
  }
  const int y1 = min(y0 + mlen, d.s1);
  for (int i1 = y0; i1 < y1; ++i1) {
    const int v = (i0 * d.s1 + i1) * d.s2 + x;
    float gp[K][4], pk[K][4], tk[K][4], mt[KL ? K : 1][4], lq[1][4], lt[1][4];
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, klS[4] = {0.f, 0.f, 0.f, 0.f};
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (mask) {
      const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
      m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
    }
    softmax_pair4<K, false, ST>(pred, ref, (int64_t)n * K * V + v, V, ref_is_prob, pk, tk, lq, lt);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k >= 1 && R) fold(k, i1 + 1, za[k - 1][2], zb[k - 1][2]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dv = pk[k][q] - tk[k][q];
        float g = c_mse * 2.f * m[q] * m[q] * dv;
        if (k >= 1 && R) {
          float sa, sb;
          if (DIM == 2) {
            sa = za[k - 1][0].v[q] + 2.f * za[k - 1][1].v[q] + za[k - 1][2].v[q];
            sb = zb[k - 1][2].v[q] - zb[k - 1][0].v[q];
          } else {
            sa = za[k - 1][2].v[q] - za[k - 1][0].v[q];
            sb = zb[k - 1][0].v[q] + 2.f * zb[k - 1][1].v[q] + zb[k - 1][2].v[q];
          }
          g += c_a * sa + c_b * sb;
        }
        g *= gs;
        gp[k][q] = g;
        dot[q] += g * pk[k][q];
        if (KL) { mt[k][q] = m[q] * kl_prob(pk[k][q] - dv, ref_is_prob); klS[q] += mt[k][q]; }
      }
      if (k >= 1 && R) {
        za[k - 1][0] = za[k - 1][1]; za[k - 1][1] = za[k - 1][2];
        zb[k - 1][0] = zb[k - 1][1]; zb[k - 1][1] = zb[k - 1][2];
      }
    }
    if (ok) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o4[q] = pk[k][q] * (gp[k][q] - dot[q]);
          if (KL) o4[q] += gs * c_kl * (pk[k][q] * klS[q] - mt[k][q]);
        }
        st4(gpred + ((int64_t)n * K + k) * V + v, o4[0], o4[1], o4[2], o4[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// f2 in 3D (round 5): the fused loss marching along Z, with the y neighbours of a row exchanged through LDS.
// k_loss_fused_fwd4 / _bwd4 march along y inside ONE plane and fold z by reading three planes per row: every voxel's softmax
// is evaluated 3 x (mlen + 2) / mlen times and a row step issues 24 loads instead of 9 -- slower in 3D than the three-kernel
// form it was meant to replace (lesson 48).  Here a workgroup of 4 waves owns RW = 4 x (64 / lanes-per-row) consecutive rows
// of a chunk of planes (its first and last row are halo: they only feed their neighbours) and walks the planes za-1 .. zb:
//   * a thread reads the logits of ITS 4 voxels of the plane once, forms the softmax difference and its x fold (DPP lane
//     shifts) in registers,
//   * the z fold (1, 2, 1) is two running partial sums per value -- (x[z-2] + 2 x[z-1]) + x[z], the order of the a0 loop of
//     fold_row4, so the folded values are the bits of the y-marching kernels --,
//   * the folded values of plane z-1 go to LDS (2 (K-1) quads per thread, double-buffered: ONE barrier per plane) and every
//     output row takes rows y-1 / y+1 from there for the y part of the two stencils.
// Reads per voxel: (RW / (RW - 2)) x ((zc + 2) / zc) = 1.14 x 1.25 of the logits instead of 3.75 x; nothing but R is written.
// Backward: the same walk over R (adjoint stencils), the softmax recomputed at the own voxel of the output plane only.
// ---------------------------------------------------------------------------------------------
constexpr int kZ3Waves = 4;

struct Z3Pos {
  int lpr, RW, RO, rr, xq, x, y, yc, za, zb;
  bool first, last, lane_ok, yin, own;
};
__device__ __forceinline__ Z3Pos z3_decode(const Dims& d, int zc) {
  Z3Pos p;
  p.lpr = d.s2 >> 2;
  const int gpw = 64 / p.lpr;
  p.RW = kZ3Waves * gpw;
  p.RO = p.RW - 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / p.lpr;
  p.xq = lane - grp * p.lpr;
  p.lane_ok = grp < gpw;
  p.rr = wave * gpw + min(grp, gpw - 1);
  const int nyt = (d.s1 + p.RO - 1) / p.RO;
  const int ty = blockIdx.x % nyt, tz = blockIdx.x / nyt;
  p.y = ty * p.RO - 1 + p.rr;
  p.yin = p.y >= 0 && p.y < d.s1;
  p.yc = min(max(p.y, 0), d.s1 - 1);
  p.za = tz * zc;
  p.zb = min(p.za + zc, d.s0);
  p.x = 4 * p.xq;
  p.first = p.xq == 0;
  p.last = p.xq == p.lpr - 1;
  p.own = p.lane_ok && p.rr >= 1 && p.rr <= p.RO && p.yin;
  return p;
}

// x fold of a quad with the neighbouring lanes' ends (zero across a row end): s = l + 2 c + r, dd = l - r
__device__ __forceinline__ void xfold4(const float (&c)[4], bool first, bool last, Quad& sq, Quad& dq) {
  const float pw = lane_prev_f(c[3]), nx = lane_next_f(c[0]);
  const float l[4] = {first ? 0.f : pw, c[0], c[1], c[2]};
  const float r[4] = {c[1], c[2], c[3], last ? 0.f : nx};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sq.v[q] = l[q] + 2.f * c[q] + r[q];
    dq.v[q] = l[q] - r[q];
  }
}

template <int K, bool KL>
__global__ void __launch_bounds__(kZ3Waves * 64) __attribute__((amdgpu_waves_per_eu(3)))
k_loss_fused_fwd3d_z(const float* __restrict__ pred, const float* __restrict__ ref, const float* __restrict__ mask,
                     float* __restrict__ R, float* __restrict__ sums, Dims d, int zc, int ref_is_prob) {
  constexpr int CH = 2 * (K - 1);
  extern __shared__ float4 xch[];                 // [2][RW][CH][lpr]
  __shared__ float smem[16];
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const Z3Pos t = z3_decode(d, zc);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};            // mse, edge A, edge B, kl
  Quad xprev[CH], part[CH];                       // [2 (k-1)] = x-smoothed, [2 (k-1) + 1] = x-differenced
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { xprev[i].v[q] = 0.f; part[i].v[q] = 0.f; }
  float4 na[K], nb[K];
  auto request = [&](int z) {
    const int zq = min(max(z, 0), d.s0 - 1);
    const int64_t o = (int64_t)n * K * V + ((int64_t)zq * d.s1 + t.yc) * d.s2 + t.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      na[k] = *reinterpret_cast<const float4*>(pred + o + (int64_t)k * V);
      nb[k] = *reinterpret_cast<const float4*>(ref + o + (int64_t)k * V);
    }
  };
  // (no software prefetch of the next plane here: with it the K = 4 kernel needs 184 VGPRs -- two waves a SIMD; without, 153 and
  // three: 81 against 89 us at 4 x 4 x 128 x 128 x 64.  The backward is the other way round: 74 against 102 us.)
  int buf = 0;
  for (int z = t.za - 1; z <= t.zb; ++z) {
    request(z);
    float p[K][4], r[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      p[k][0] = na[k].x; p[k][1] = na[k].y; p[k][2] = na[k].z; p[k][3] = na[k].w;
      r[k][0] = nb[k].x; r[k][1] = nb[k].y; r[k][2] = nb[k].z; r[k][3] = nb[k].w;
    }
    const bool in = z >= 0 && z < d.s0 && t.yin;
    float P[K][4], T[K][4], lq[KL ? K : 1][4], lt[KL ? K : 1][4];
    softmax_quads<K, KL>(p, r, ref_is_prob, P, T, lq, lt);
    if (t.own && z >= t.za && z < t.zb) {          // the own voxels of the chunk's planes: mse / kl sums
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (mask) {
        const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + ((int64_t)z * d.s1 + t.y) * d.s2 + t.x);
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float e = P[k][q] * m[q] - T[k][q] * m[q];
          acc[0] += e * e;
          if (KL) acc[3] += kl_term(T[k][q], lt[k][q], lq[k][q], m[q], ref_is_prob);
        }
    }
    // x fold of D = P - T (zero outside the volume), then the z fold of plane z - 1: (x[z-2] + 2 x[z-1]) + x[z]
    Quad fz[CH];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      float c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = in ? P[k][q] - T[k][q] : 0.f;
      Quad xs, xd;
      xfold4(c, t.first, t.last, xs, xd);
      const int is = 2 * (k - 1), id = is + 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fz[is].v[q] = part[is].v[q] + 1.f * xs.v[q];
        fz[id].v[q] = part[id].v[q] + 1.f * xd.v[q];
        part[is].v[q] = (0.f + 1.f * xprev[is].v[q]) + 2.f * xs.v[q];
        part[id].v[q] = (0.f + 1.f * xprev[id].v[q]) + 2.f * xd.v[q];
        xprev[is].v[q] = xs.v[q];
        xprev[id].v[q] = xd.v[q];
      }
    }
    const int pz = z - 1;
    if (pz < t.za) continue;                      // (uniform: the chunk's first output plane needs one more step)
    float4* mine = xch + ((buf * t.RW + t.rr) * CH) * t.lpr + t.xq;
    if (t.lane_ok) {
#pragma unroll
      for (int i = 0; i < CH; ++i) mine[i * t.lpr] = make_float4(fz[i].v[0], fz[i].v[1], fz[i].v[2], fz[i].v[3]);
    }
    __syncthreads();
    if (t.own) {
      const float4* up = xch + ((buf * t.RW + t.rr - 1) * CH) * t.lpr + t.xq;
      const float4* dn = xch + ((buf * t.RW + t.rr + 1) * CH) * t.lpr + t.xq;
      const int v = (pz * d.s1 + t.y) * d.s2 + t.x;
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (mask) {
        const float4 mm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + v);
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
      }
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const int is = 2 * (k - 1), id = is + 1;
        const float4 su = up[is * t.lpr], sd = dn[is * t.lpr], du = up[id * t.lpr], dd = dn[id * t.lpr];
        const float s0[4] = {su.x, su.y, su.z, su.w}, s2[4] = {sd.x, sd.y, sd.z, sd.w};
        const float d0[4] = {du.x, du.y, du.z, du.w}, d2[4] = {dd.x, dd.y, dd.z, dd.w};
        float ra[4], rb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float ga = s0[q] - s2[q];
          const float gb = d0[q] + 2.f * fz[id].v[q] + d2[q];
          const float ea = ga * m[q], eb = gb * m[q];
          acc[1] += ea * ea;
          acc[2] += eb * eb;
          ra[q] = 2.f * m[q] * m[q] * ga;
          rb[q] = 2.f * m[q] * m[q] * gb;
        }
        if (R) {
          *reinterpret_cast<float4*>(R + ((int64_t)n * CH + is) * V + v) = make_float4(ra[0], ra[1], ra[2], ra[3]);
          *reinterpret_cast<float4*>(R + ((int64_t)n * CH + id) * V + v) = make_float4(rb[0], rb[1], rb[2], rb[3]);
        }
      }
    }
    buf ^= 1;
  }
  block_sum<4>(acc, smem);
  if (threadIdx.x == 0) {
    atomic_add_f32(sums + sum_slot(), acc[0]);
    atomic_add_f32(sums + kSumSlots + sum_slot(), acc[1]);
    atomic_add_f32(sums + 2 * kSumSlots + sum_slot(), acc[2]);
    if (KL) atomic_add_f32(sums + 3 * kSumSlots + sum_slot(), acc[3]);
  }
}

template <int K, bool KL>
__global__ void __launch_bounds__(kZ3Waves * 64)
k_loss_fused_bwd3d_z(const float* __restrict__ pred, const float* __restrict__ ref, const float* __restrict__ R,
                     const float* __restrict__ mask, const float* __restrict__ gscale, float* __restrict__ gpred, float c_mse,
                     float c_a, float c_b, Dims d, int zc, float c_kl, int ref_is_prob) {
  constexpr int CH = 2 * (K - 1);
  extern __shared__ float4 xch[];                 // [2][RW][CH][lpr]: a_k (x- and z-smoothed R_A), b_k (-(x-differenced, z-smoothed R_B))
  const int n = blockIdx.y;
  const int V = (int)d.voxels();
  const Z3Pos t = z3_decode(d, zc);
  const float gs = gscale ? gscale[0] : 1.f;
  Quad xprev[CH], part[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { xprev[i].v[q] = 0.f; part[i].v[q] = 0.f; }
  float4 nr[CH], na[K], nb[K], nm = make_float4(1.f, 1.f, 1.f, 1.f);
  auto request_r = [&](int z) {
    const int zq = min(max(z, 0), d.s0 - 1);
    const int64_t o = (int64_t)n * CH * V + ((int64_t)zq * d.s1 + t.yc) * d.s2 + t.x;
#pragma unroll
    for (int i = 0; i < CH; ++i) nr[i] = *reinterpret_cast<const float4*>(R + o + (int64_t)i * V);
  };
  auto request_own = [&](int pz) {                // logits + mask of the output plane (own rows only use them)
    const int zq = min(max(pz, 0), d.s0 - 1);
    const int64_t sp = ((int64_t)zq * d.s1 + t.yc) * d.s2 + t.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      na[k] = *reinterpret_cast<const float4*>(pred + ((int64_t)n * K + k) * V + sp);
      nb[k] = *reinterpret_cast<const float4*>(ref + ((int64_t)n * K + k) * V + sp);
    }
    if (mask) nm = *reinterpret_cast<const float4*>(mask + (int64_t)n * V + sp);
  };
  request_r(t.za - 1);
  int buf = 0;
  for (int z = t.za - 1; z <= t.zb; ++z) {
    float rc[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) { rc[i][0] = nr[i].x; rc[i][1] = nr[i].y; rc[i][2] = nr[i].z; rc[i][3] = nr[i].w; }
    const bool in = z >= 0 && z < d.s0 && t.yin;
    const int pz = z - 1;
    if (z < t.zb) request_r(z + 1);               // the next plane of R travels while this one is worked on
    if (pz >= t.za && t.own) request_own(pz);
    // x fold of R_A (smoothed) and R_B (differenced), z fold of plane z - 1
    Quad fz[CH];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      const int ia = 2 * (k - 1), ib = ia + 1;
      float ca[4], cb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { ca[q] = in ? rc[ia][q] : 0.f; cb[q] = in ? rc[ib][q] : 0.f; }
      Quad as, ad, bs, bd;
      xfold4(ca, t.first, t.last, as, ad);
      xfold4(cb, t.first, t.last, bs, bd);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fz[ia].v[q] = part[ia].v[q] + 1.f * as.v[q];
        fz[ib].v[q] = -(part[ib].v[q] + 1.f * bd.v[q]);
        part[ia].v[q] = (0.f + 1.f * xprev[ia].v[q]) + 2.f * as.v[q];
        part[ib].v[q] = (0.f + 1.f * xprev[ib].v[q]) + 2.f * bd.v[q];
        xprev[ia].v[q] = as.v[q];
        xprev[ib].v[q] = bd.v[q];
      }
      (void)ad; (void)bs;
    }
    if (pz < t.za) continue;                      // (uniform)
    float4* mine = xch + ((buf * t.RW + t.rr) * CH) * t.lpr + t.xq;
    if (t.lane_ok) {
#pragma unroll
      for (int i = 0; i < CH; ++i) mine[i * t.lpr] = make_float4(fz[i].v[0], fz[i].v[1], fz[i].v[2], fz[i].v[3]);
    }
    __syncthreads();
    if (t.own) {
      const float4* up = xch + ((buf * t.RW + t.rr - 1) * CH) * t.lpr + t.xq;
      const float4* dn = xch + ((buf * t.RW + t.rr + 1) * CH) * t.lpr + t.xq;
      const int v = (pz * d.s1 + t.y) * d.s2 + t.x;
      float p[K][4], r[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        p[k][0] = na[k].x; p[k][1] = na[k].y; p[k][2] = na[k].z; p[k][3] = na[k].w;
        r[k][0] = nb[k].x; r[k][1] = nb[k].y; r[k][2] = nb[k].z; r[k][3] = nb[k].w;
      }
      const float m[4] = {nm.x, nm.y, nm.z, nm.w};
      float pk[K][4], tk[K][4], lq[1][4], lt[1][4];
      softmax_quads<K, false>(p, r, ref_is_prob, pk, tk, lq, lt);
      float gp[K][4], mt[KL ? K : 1][4];
      float dot[4] = {0.f, 0.f, 0.f, 0.f}, klS[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float a0[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f}, b0[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
        if (k >= 1) {
          const int ia = 2 * (k - 1), ib = ia + 1;
          const float4 au = up[ia * t.lpr], ad = dn[ia * t.lpr], bu = up[ib * t.lpr], bd = dn[ib * t.lpr];
          a0[0] = au.x; a0[1] = au.y; a0[2] = au.z; a0[3] = au.w;
          a2[0] = ad.x; a2[1] = ad.y; a2[2] = ad.z; a2[3] = ad.w;
          b0[0] = bu.x; b0[1] = bu.y; b0[2] = bu.z; b0[3] = bu.w;
          b2[0] = bd.x; b2[1] = bd.y; b2[2] = bd.z; b2[3] = bd.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dv = pk[k][q] - tk[k][q];
          float g = c_mse * 2.f * m[q] * m[q] * dv;
          if (k >= 1) {
            const float sa = a2[q] - a0[q];
            const float sb = b0[q] + 2.f * fz[2 * (k >= 1 ? k - 1 : 0) + 1].v[q] + b2[q];
            g += c_a * sa + c_b * sb;
          }
          g *= gs;
          gp[k][q] = g;
          dot[q] += g * pk[k][q];
          if (KL) { mt[k][q] = m[q] * kl_prob(pk[k][q] - dv, ref_is_prob); klS[q] += mt[k][q]; }
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o4[q] = pk[k][q] * (gp[k][q] - dot[q]);
          if (KL) o4[q] += gs * c_kl * (pk[k][q] * klS[q] - mt[k][q]);
        }
        *reinterpret_cast<float4*>(gpred + ((int64_t)n * K + k) * V + v) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
    }
    buf ^= 1;
  }
}

// ---- the z-marching fused 3D loss: shapes, chunk length, grid, LDS
static inline bool z3_takes(const Dims& d, bool edges) {
  static const bool off = getenv("ADVCHAIN_NO_FUSED_LOSS_3DZ") != nullptr;   // A/B knob
  if (off || !edges || d.s0 < 2 || d.s2 % 4 != 0) return false;
  const int lpr = d.s2 / 4;
  return lpr >= 1 && lpr <= 32;        // at least two rows a wave: RW = 4 x (64 / lpr) >= 8 rows, 6 of them own
}
static inline int z3_rows_out(const Dims& d) { return kZ3Waves * (64 / (d.s2 / 4)) - 2; }
// planes per chunk: every chunk re-reads 2 halo planes, so as long as possible while the launch keeps ~3 workgroups a CU
static inline int z3_chunk(const Dims& d, int64_t N) {
  const int64_t tiles = N * ((d.s1 + z3_rows_out(d) - 1) / z3_rows_out(d));
  int zc = d.s0;
  while (zc > 4 && tiles * ((d.s0 + zc - 1) / zc) < 768) zc = (zc + 1) / 2;
  return zc;
}
static inline dim3 z3_grid(const Dims& d, int64_t N, int zc) {
  const int nyt = (d.s1 + z3_rows_out(d) - 1) / z3_rows_out(d);
  return dim3((unsigned)(nyt * ((d.s0 + zc - 1) / zc)), (unsigned)N);
}
static inline size_t z3_lds(const Dims& d, int64_t K) {
  const int lpr = d.s2 / 4, RW = kZ3Waves * (64 / lpr);
  return (size_t)2 * RW * 2 * (K - 1) * lpr * sizeof(float4);
}

static inline dim3 march_grid(const Dims& d, int64_t N) {
  const int strips = (d.s2 >> 6) * ((d.s1 + kMarch - 1) / kMarch) * d.s0;
  return dim3((unsigned)((strips + kBlock / 64 - 1) / (kBlock / 64)), (unsigned)N);
}
static const bool g_no_march = getenv("ADVCHAIN_NO_MARCH") != nullptr;   // A/B knob
static const bool g_no_march4 = getenv("ADVCHAIN_NO_MARCH4") != nullptr;   // A/B knob

static inline bool march4_ok(const Dims& d, const void* a, const void* b, const void* c, const void* e) {
  if (g_no_march4 || d.s2 % 4 != 0) return false;
  const int lpr = d.s2 / 4;
  if (lpr > 64 || lpr < 1) return false;       // a row must fit one wave; rows need not divide it (idle tail lanes)
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                       reinterpret_cast<uintptr_t>(e);
  return (al & 15) == 0;
}
// rows marched per strip: as long as possible (each strip re-reads 2 halo rows) while the launch still has ~1024 waves
// (measured: 8 rows beat 4 at 2048 waves, 3D 84 vs 92 us; 2D 1024 waves: 8 and 4 rows equal)
static inline int march4_len(const Dims& d, int64_t N) {
  static const int forced = 0;   // measured optimum (was a tuning knob until round 4)
  if (forced > 0) return forced;
  const int groups_per_wave = 64 / (d.s2 / 4);
  for (int m = kMarch; m > 2; m /= 2)
    if (N * d.s0 * ((d.s1 + m - 1) / m) / groups_per_wave >= 1024) return m;
  return 2;
}
static inline dim3 march4_grid(const Dims& d, int64_t N, int mlen) {
  const int strips = ((d.s1 + mlen - 1) / mlen) * d.s0;
  const int per_block = (kBlock / 64) * (64 / (d.s2 / 4));
  return dim3((unsigned)((strips + per_block - 1) / per_block), (unsigned)N);
}

template <int DIM>
static bool launch_edge_march(int64_t K, int64_t N, const Dims& d, hipStream_t st, const float* D, const float* mask,
                              float* R, float* sums, int mask_ch, bool rows) {
  if (g_no_march) return false;
  if (march4_ok(d, D, mask, R, nullptr)) {
    const int mlen = march4_len(d, N);
    const dim3 g4 = march4_grid(d, N, mlen), b4(kBlock);
    switch (K) {
      case 2: hipLaunchKernelGGL((k_edge_fwd_march4<DIM, 2>), g4, b4, 0, st, D, mask, R, sums, d, mask_ch, mlen); return true;
      case 3: hipLaunchKernelGGL((k_edge_fwd_march4<DIM, 3>), g4, b4, 0, st, D, mask, R, sums, d, mask_ch, mlen); return true;
      case 4: hipLaunchKernelGGL((k_edge_fwd_march4<DIM, 4>), g4, b4, 0, st, D, mask, R, sums, d, mask_ch, mlen); return true;
      case 5: hipLaunchKernelGGL((k_edge_fwd_march4<DIM, 5>), g4, b4, 0, st, D, mask, R, sums, d, mask_ch, mlen); return true;
      default: break;
    }
  }
  if (!rows) return false;   // the scalar march needs whole waves per row (S2 % 64 == 0)
  const dim3 g = march_grid(d, N), b(kBlock);
  switch (K) {
    case 2: hipLaunchKernelGGL((k_edge_fwd_march<DIM, 2>), g, b, 0, st, D, mask, R, sums, d, mask_ch); return true;
    case 3: hipLaunchKernelGGL((k_edge_fwd_march<DIM, 3>), g, b, 0, st, D, mask, R, sums, d, mask_ch); return true;
    case 4: hipLaunchKernelGGL((k_edge_fwd_march<DIM, 4>), g, b, 0, st, D, mask, R, sums, d, mask_ch); return true;
    case 5: hipLaunchKernelGGL((k_edge_fwd_march<DIM, 5>), g, b, 0, st, D, mask, R, sums, d, mask_ch); return true;
    default: return false;
  }
}

template <int DIM>
static bool launch_bwd_march(int64_t K, int64_t N, const Dims& d, hipStream_t st, const float* P, const float* D,
                             const float* R, const float* mask, const float* gscale, float* gpred, float c_mse,
                             float c_a, float c_b, int mask_ch, bool rows, float c_kl, int kl_gt) {
  const bool kl = c_kl != 0.f;
  if (g_no_march) return false;
  if (march4_ok(d, P, D, R, mask) && (reinterpret_cast<uintptr_t>(gpred) & 15) == 0) {
    const int mlen = march4_len(d, N);
    const dim3 g4 = march4_grid(d, N, mlen), b4(kBlock);
    switch (K) {
      case 2:
        if (kl) hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 2, true>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        else hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 2, false>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        return true;
      case 3:
        if (kl) hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 3, true>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        else hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 3, false>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        return true;
      case 4:
        if (kl) hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 4, true>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        else hipLaunchKernelGGL((k_consistency_bwd_march4<DIM, 4, false>), g4, b4, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, mlen, c_kl, kl_gt);
        return true;
      default: break;   // K = 5: the scalar march (register budget)
    }
  }
  if (!rows) return false;   // the scalar march needs whole waves per row (S2 % 64 == 0)
  const dim3 g = march_grid(d, N), b(kBlock);
  switch (K) {
    case 2:
      if (kl) hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 2, true>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      else hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 2, false>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      return true;
    case 3:
      if (kl) hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 3, true>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      else hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 3, false>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      return true;
    case 4:
      if (kl) hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 4, true>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      else hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 4, false>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      return true;
    case 5:
      if (kl) hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 5, true>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      else hipLaunchKernelGGL((k_consistency_bwd_march<DIM, 5, false>), g, b, 0, st, P, D, R, mask, gscale, gpred, c_mse, c_a, c_b, d, mask_ch, c_kl, kl_gt);
      return true;
    default: return false;
  }
}

static inline bool ldims_ok(int ndim, const int64_t* s) {
  if (ndim != 2 && ndim != 3) return false;
  for (int i = 0; i < ndim; ++i)
    if (s[i] < 1 || s[i] > (1 << 24)) return false;
  return true;
}
static inline Dims lmake_dims(int ndim, const int64_t* s) {
  Dims d;
  if (ndim == 3) { d.s0 = (int)s[0]; d.s1 = (int)s[1]; d.s2 = (int)s[2]; }
  else { d.s0 = 1; d.s1 = (int)s[0]; d.s2 = (int)s[1]; }
  return d;
}

extern "C" {

int advchain_consistency_fwd(const float* pred, const float* ref, const float* mask, float* P, float* D, float* R,
                             float* sums, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                             int ref_is_prob, int want_edges, int want_kl, void* stream) {
  ADVCHAIN_CHECK_ARG(pred && ref && P && D && sums, "consistency_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && K >= 1 && K <= kMaxK, "consistency_fwd: bad N/K (K <= 16)");
  ADVCHAIN_CHECK_ARG(!mask || mask_channels == 1 || mask_channels == K, "consistency_fwd: mask must have 1 or K channels");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = lmake_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "consistency_fwd: volume too large");
  const int V = (int)d.voxels();
  dim3 g(advchain_blocks(V, kBlock), (unsigned)N), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  const bool al16 = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(mask) |
                      reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(D)) & 15) == 0;
  if (V % 4 == 0 && al16 && K >= 2 && K <= 5) {
    dim3 g4(advchain_blocks(V / 4, kBlock), (unsigned)N);
    switch (K) {
      case 2: hipLaunchKernelGGL(k_softmax_diff_v4<2>, g4, b, 0, st, pred, ref, mask, P, D, sums, V, mask_channels, ref_is_prob, want_kl); break;
      case 3: hipLaunchKernelGGL(k_softmax_diff_v4<3>, g4, b, 0, st, pred, ref, mask, P, D, sums, V, mask_channels, ref_is_prob, want_kl); break;
      case 4: hipLaunchKernelGGL(k_softmax_diff_v4<4>, g4, b, 0, st, pred, ref, mask, P, D, sums, V, mask_channels, ref_is_prob, want_kl); break;
      default: hipLaunchKernelGGL(k_softmax_diff_v4<5>, g4, b, 0, st, pred, ref, mask, P, D, sums, V, mask_channels, ref_is_prob, want_kl); break;
    }
  } else {
    hipLaunchKernelGGL(k_softmax_diff, g, b, 0, st, pred, ref, mask, P, D, sums, (int)K, V, mask_channels, ref_is_prob, want_kl);
  }
  if (want_edges && K > 1) {
    const bool rows = (d.s2 % 64) == 0;   // lane <-> x with whole waves per row: DPP neighbour exchange
    if (ndim == 3) {
      if (mask_channels <= 1 && launch_edge_march<3>(K, N, d, st, D, mask, R, sums, mask_channels, rows)) {}
      else if (rows) hipLaunchKernelGGL(k_edge_fwd_rows<3>, g, b, 0, st, D, mask, R, sums, (int)K, d, mask_channels);
      else hipLaunchKernelGGL(k_edge_fwd<3>, g, b, 0, st, D, mask, R, sums, (int)K, d, mask_channels);
    } else {
      if (mask_channels <= 1 && launch_edge_march<2>(K, N, d, st, D, mask, R, sums, mask_channels, rows)) {}
      else if (rows) hipLaunchKernelGGL(k_edge_fwd_rows<2>, g, b, 0, st, D, mask, R, sums, (int)K, d, mask_channels);
      else hipLaunchKernelGGL(k_edge_fwd<2>, g, b, 0, st, D, mask, R, sums, (int)K, d, mask_channels);
    }
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_consistency_finish(float* slots, const float* coef4_host, float* sums, float* value, int reset, void* stream) {
  ADVCHAIN_CHECK_ARG(slots && coef4_host && sums && value, "consistency_finish: null pointer");
  static_assert(kSumSlots == 64 && kBlock == 256, "one wave per row of slots");
  hipLaunchKernelGGL(k_consistency_finish, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, slots, coef4_host[0], coef4_host[1],
                     coef4_host[2], coef4_host[3], sums, value, reset);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_consistency_bwd(const float* P, const float* D, const float* R, const float* mask,
                             const float* grad_scale, float* grad_pred, float c_mse, float c_a, float c_b, float c_kl,
                             int kl_is_gt, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                             void* stream) {
  ADVCHAIN_CHECK_ARG(P && D && grad_pred, "consistency_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && K >= 1 && K <= kMaxK, "consistency_bwd: bad N/K (K <= 16)");
  ADVCHAIN_CHECK_ARG(!mask || mask_channels == 1 || mask_channels == K, "consistency_bwd: mask must have 1 or K channels");
  if (N == 0) return ADVCHAIN_OK;
  const Dims d = lmake_dims(ndim, dims);
  ADVCHAIN_CHECK_ARG(d.voxels() < (1ll << 31), "consistency_bwd: volume too large");
  dim3 g(advchain_blocks(d.voxels(), kBlock), (unsigned)N), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  const bool rows = (d.s2 % 64) == 0;
  if (ndim == 3) {
    if (launch_bwd_march<3>(K, N, d, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, mask_channels, rows, c_kl, kl_is_gt)) {}
    else if (rows) hipLaunchKernelGGL(k_consistency_bwd_rows<3>, g, b, 0, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, (int)K, d, mask_channels, c_kl, kl_is_gt);
    else hipLaunchKernelGGL(k_consistency_bwd<3>, g, b, 0, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, (int)K, d, mask_channels, c_kl, kl_is_gt);
  } else {
    if (launch_bwd_march<2>(K, N, d, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, mask_channels, rows, c_kl, kl_is_gt)) {}
    else if (rows) hipLaunchKernelGGL(k_consistency_bwd_rows<2>, g, b, 0, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, (int)K, d, mask_channels, c_kl, kl_is_gt);
    else hipLaunchKernelGGL(k_consistency_bwd<2>, g, b, 0, st, P, D, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, (int)K, d, mask_channels, c_kl, kl_is_gt);
  }
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// ---- f2 fused (round 4): see k_loss_fused_fwd4.  ADVCHAIN_ERR_UNSUPPORTED (-2) for what the 16-byte marching form does not
// take (rows of 4j <= 256 voxels, K = 2..4, at most a one-channel mask, 16-byte aligned tensors): use the entries above.
int advchain_consistency_fused_fwd(const float* pred, const float* ref, const float* mask, float* R, float* sums, int64_t N,
                                   int64_t K, int ndim, const int64_t* dims, int mask_channels, int ref_is_prob, int want_edges,
                                   int want_kl, void* stream) {
  ADVCHAIN_CHECK_ARG(pred && ref && sums, "consistency_fused_fwd: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_fused_fwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && K >= 1 && K <= kMaxK, "consistency_fused_fwd: bad N/K (K <= 16)");
  static const bool off = getenv("ADVCHAIN_NO_FUSED_LOSS") != nullptr;   // A/B knob
  // 3D: the y-marching kernel below was measured SLOWER than the three-kernel form (K = 4: 4 x 4 x 128 x 128 x 64 230 against
  // 199 us -- its z fold reads three planes per row) and is instantiated for 2D only since round 5; 3D takes
  // k_loss_fused_fwd3d_z (z-marching with the y neighbours through LDS; edges wanted, rows of at most 128 voxels;
  // ADVCHAIN_NO_FUSED_LOSS_3DZ switches it off for A/B) or the unfused entries.
  const Dims d = lmake_dims(ndim, dims);
  const bool edges = want_edges && K > 1;
  const bool zmarch = ndim == 3 && z3_takes(d, edges);
  if (off || (ndim == 3 && !zmarch) || K < 2 || K > 4 || (mask && mask_channels != 1)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (d.voxels() >= (1ll << 31) || !march4_ok(d, pred, ref, mask, R)) return ADVCHAIN_ERR_UNSUPPORTED;
  if (N == 0) return ADVCHAIN_OK;
  if (zmarch) {
    const int zc = z3_chunk(d, N);
    const dim3 gz = z3_grid(d, N, zc), bz(kZ3Waves * 64);
    const size_t lds = z3_lds(d, K);
    hipStream_t stz = (hipStream_t)stream;
#define FWD3DZ(K_) do { \
      if (want_kl) hipLaunchKernelGGL((k_loss_fused_fwd3d_z<K_, true>), gz, bz, lds, stz, pred, ref, mask, R, sums, d, zc, ref_is_prob); \
      else hipLaunchKernelGGL((k_loss_fused_fwd3d_z<K_, false>), gz, bz, lds, stz, pred, ref, mask, R, sums, d, zc, ref_is_prob); } while (0)
    switch (K) { case 2: FWD3DZ(2); break; case 3: FWD3DZ(3); break; default: FWD3DZ(4); break; }
#undef FWD3DZ
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  if (edges && !R) R = nullptr;                      // (no gradient wanted: the sums only)
  const int mlen = march4_len(d, N);
  const dim3 g4 = march4_grid(d, N, mlen), b4(kBlock);
  hipStream_t st = (hipStream_t)stream;
#define FUSED_FWD(DIM_, K_) do { \
    if (want_kl) { if (edges) hipLaunchKernelGGL((k_loss_fused_fwd4<DIM_, K_, true, true>), g4, b4, 0, st, pred, ref, mask, R, sums, d, mlen, ref_is_prob); \
                   else hipLaunchKernelGGL((k_loss_fused_fwd4<DIM_, K_, true, false>), g4, b4, 0, st, pred, ref, mask, R, sums, d, mlen, ref_is_prob); } \
    else { if (edges) hipLaunchKernelGGL((k_loss_fused_fwd4<DIM_, K_, false, true>), g4, b4, 0, st, pred, ref, mask, R, sums, d, mlen, ref_is_prob); \
           else hipLaunchKernelGGL((k_loss_fused_fwd4<DIM_, K_, false, false>), g4, b4, 0, st, pred, ref, mask, R, sums, d, mlen, ref_is_prob); } } while (0)
  switch (K) { case 2: FUSED_FWD(2, 2); break; case 3: FUSED_FWD(2, 3); break; default: FUSED_FWD(2, 4); break; }
#undef FUSED_FWD
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_consistency_fused_bwd(const float* pred, const float* ref, const float* R, const float* mask,
                                   const float* grad_scale, float* grad_pred, float c_mse, float c_a, float c_b, float c_kl,
                                   int ref_is_prob, int64_t N, int64_t K, int ndim, const int64_t* dims, int mask_channels,
                                   void* stream) {
  ADVCHAIN_CHECK_ARG(pred && ref && grad_pred, "consistency_fused_bwd: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_fused_bwd: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536 && K >= 1 && K <= kMaxK, "consistency_fused_bwd: bad N/K (K <= 16)");
  if (K < 2 || K > 4 || (mask && mask_channels != 1)) return ADVCHAIN_ERR_UNSUPPORTED;
  const Dims d = lmake_dims(ndim, dims);
  if (d.voxels() >= (1ll << 31) || !march4_ok(d, pred, ref, R, mask) || (reinterpret_cast<uintptr_t>(grad_pred) & 15) != 0)
    return ADVCHAIN_ERR_UNSUPPORTED;
  if (N == 0) return ADVCHAIN_OK;
  const bool kl = c_kl != 0.f;
  if (ndim == 3 && R && z3_takes(d, true)) {          // the forward of this shape marched along z: so does its backward
    const int zc = z3_chunk(d, N);
    const dim3 gz = z3_grid(d, N, zc), bz(kZ3Waves * 64);
    const size_t lds = z3_lds(d, K);
    hipStream_t stz = (hipStream_t)stream;
#define BWD3DZ(K_) do { \
      if (kl) hipLaunchKernelGGL((k_loss_fused_bwd3d_z<K_, true>), gz, bz, lds, stz, pred, ref, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, d, zc, c_kl, ref_is_prob); \
      else hipLaunchKernelGGL((k_loss_fused_bwd3d_z<K_, false>), gz, bz, lds, stz, pred, ref, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, d, zc, c_kl, ref_is_prob); } while (0)
    switch (K) { case 2: BWD3DZ(2); break; case 3: BWD3DZ(3); break; default: BWD3DZ(4); break; }
#undef BWD3DZ
    ADVCHAIN_LAUNCH_CHECK();
    return ADVCHAIN_OK;
  }
  if (ndim == 3) return ADVCHAIN_ERR_UNSUPPORTED;      // (no forward of such a shape ran fused)
  const int mlen = march4_len(d, N);
  const dim3 g4 = march4_grid(d, N, mlen), b4(kBlock);
  hipStream_t st = (hipStream_t)stream;
#define FUSED_BWD(DIM_, K_) do { \
    if (kl) hipLaunchKernelGGL((k_loss_fused_bwd4<DIM_, K_, true>), g4, b4, 0, st, pred, ref, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, d, mlen, c_kl, ref_is_prob); \
    else hipLaunchKernelGGL((k_loss_fused_bwd4<DIM_, K_, false>), g4, b4, 0, st, pred, ref, R, mask, grad_scale, grad_pred, c_mse, c_a, c_b, d, mlen, c_kl, ref_is_prob); } while (0)
  switch (K) { case 2: FUSED_BWD(2, 2); break; case 3: FUSED_BWD(2, 3); break; default: FUSED_BWD(2, 4); break; }
#undef FUSED_BWD
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

// ---- bf16 STORAGE experiment (round 6; BASELINE config 2 says "bf16").  The 2D K = 4 fused loss with pred / ref / R /
// grad_pred stored as bfloat16 (8 bytes per lane instead of 16) and every operation in fp32 registers: what the byte-heaviest
// entries of a cfg-2 step gain from half the bytes.  NOT on the product path (the parity contract is fp32 at 1e-4;
// tools/kernel_bench.py "bf16 storage" rows, tests/test_ops_gpu.py::test_bf16_storage_experiment...).  2D, K == 4, logits on both
// sides, mse + edge terms, a one-channel fp32 mask or none; -2 for anything else.
int advchain_consistency_fused_fwd_bf16(const void* pred, const void* ref, const float* mask, void* R, float* sums, int64_t N,
                                        int64_t K, int ndim, const int64_t* dims, void* stream) {
  ADVCHAIN_CHECK_ARG(pred && ref && sums, "consistency_fused_fwd_bf16: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_fused_fwd_bf16: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "consistency_fused_fwd_bf16: bad N");
  const Dims d = lmake_dims(ndim, dims);
  const uintptr_t al = reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(R);
  if (ndim != 2 || K != 4 || d.s2 % 4 != 0 || d.s2 > 256 || (al & 7) != 0 || (reinterpret_cast<uintptr_t>(mask) & 15) != 0 ||
      d.voxels() >= (1ll << 31))
    return ADVCHAIN_ERR_UNSUPPORTED;
  if (N == 0) return ADVCHAIN_OK;
  const int mlen = march4_len(d, N);
  hipLaunchKernelGGL((k_loss_fused_fwd4<2, 4, false, true, Bf16>), march4_grid(d, N, mlen), dim3(kBlock), 0, (hipStream_t)stream,
                     static_cast<const Bf16*>(pred), static_cast<const Bf16*>(ref), mask, static_cast<Bf16*>(R), sums, d, mlen, 0);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

int advchain_consistency_fused_bwd_bf16(const void* pred, const void* ref, const void* R, const float* mask,
                                        const float* grad_scale, void* grad_pred, float c_mse, float c_a, float c_b, int64_t N,
                                        int64_t K, int ndim, const int64_t* dims, void* stream) {
  ADVCHAIN_CHECK_ARG(pred && ref && R && grad_pred, "consistency_fused_bwd_bf16: null pointer");
  ADVCHAIN_CHECK_ARG(ldims_ok(ndim, dims), "consistency_fused_bwd_bf16: bad dims");
  ADVCHAIN_CHECK_ARG(N >= 0 && N < 65536, "consistency_fused_bwd_bf16: bad N");
  const Dims d = lmake_dims(ndim, dims);
  const uintptr_t al = reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(R) |
                       reinterpret_cast<uintptr_t>(grad_pred);
  if (ndim != 2 || K != 4 || d.s2 % 4 != 0 || d.s2 > 256 || (al & 7) != 0 || (reinterpret_cast<uintptr_t>(mask) & 15) != 0 ||
      d.voxels() >= (1ll << 31))
    return ADVCHAIN_ERR_UNSUPPORTED;
  if (N == 0) return ADVCHAIN_OK;
  const int mlen = march4_len(d, N);
  hipLaunchKernelGGL((k_loss_fused_bwd4<2, 4, false, Bf16>), march4_grid(d, N, mlen), dim3(kBlock), 0, (hipStream_t)stream,
                     static_cast<const Bf16*>(pred), static_cast<const Bf16*>(ref), static_cast<const Bf16*>(R), mask, grad_scale,
                     static_cast<Bf16*>(grad_pred), c_mse, c_a, c_b, d, mlen, 0.f, 0);
  ADVCHAIN_LAUNCH_CHECK();
  return ADVCHAIN_OK;
}

}  // extern "C"
