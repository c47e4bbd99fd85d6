// Device helpers shared by the sampler kernels (sampler.hip, scatter_tiled.hip).
#pragma once
#include "common.h"

namespace advchain {

template <int DIM, int PAD>
struct Taps {
  AxisTap x, y, z;
  __device__ __forceinline__ void build(float gx, float gy, float gz, const Dims& d) {
    x = make_tap<PAD>(gx, d.s2);
    y = make_tap<PAD>(gy, d.s1);
    if (DIM == 3) z = make_tap<PAD>(gz, d.s0);
    else { z.i0 = 0; z.w0 = 1.f; z.w1 = 0.f; z.mult = 0.f; z.v0 = true; z.v1 = false; }
  }
  __device__ __forceinline__ bool ok(int cz, int cy, int cx) const {
    return (cx ? x.v1 : x.v0) && (cy ? y.v1 : y.v0) && (DIM == 3 ? (cz ? z.v1 : z.v0) : true);
  }
  __device__ __forceinline__ int off(int cz, int cy, int cx, const Dims& d) const {
    return ((z.i0 + cz) * d.s1 + (y.i0 + cy)) * d.s2 + (x.i0 + cx);
  }
  __device__ __forceinline__ float wx(int c) const { return c ? x.w1 : x.w0; }
  __device__ __forceinline__ float wy(int c) const { return c ? y.w1 : y.w0; }
  __device__ __forceinline__ float wz(int c) const { return c ? z.w1 : z.w0; }
  __device__ __forceinline__ float w(int cz, int cy, int cx) const {
    float r = wx(cx) * wy(cy);
    if (DIM == 3) r *= wz(cz);
    return r;
  }
};

// Corner offsets with every index clamped into the volume: ALL corner loads of a sample are then issued
// unconditionally and back to back, and an invalid corner's value is discarded by a select.  (With `if (ok) acc +=
// in[off] * w` the compiler wraps every load in its own branch with an `s_waitcnt vmcnt(0)` inside: the 2D squaring
// kernel made 32 SERIAL memory round trips per thread -- that, not instruction issue, was its "ceiling".)
// The two x corners of a (z, y) row are neighbours in memory: ONE 8-byte gather fetches both (a CU retires a
// vector-memory wave-instruction per ~26 clk whatever it carries, lesson 4; tools/microbench/pairbench.hip: 8 dword
// gathers 19.6 us, 4 dwordx2 gathers 14.4 us for the same corners).  The pair of a row starts at its element
// clamp(i0, 0, S2-2) and a corner takes the first or the second element -- exactly the value in[clamp(i0 + cx, 0, S2-1)]
// the dword form loads.  The address is only 4-byte aligned: gfx9+ under ROCm runs in unaligned access mode and the
// compiler emits global_load_dwordx2 for an aligned(4) packed pair.  No branch (control flow between the channels of a
// sample would serialise their round trips): rows of one voxel (S2 == 1) pair an element with its predecessor in
// memory, which needs a volume of at least two voxels (the C entries check it).
struct __attribute__((packed, aligned(4))) FloatPair { float lo, hi; };

// PAIRED = false keeps the dword form: the fallback lanes of the marching forward kernels (sample_march.hip), where the
// gathers are rare and the extra bookkeeping cost the main path 6-12 % (16.4 -> 17.8 us at 4 x 1 x 128 x 128 x 64).
template <int DIM, int PAD, bool PAIRED = true>
struct CornerOffsets {
  int x[2], y[2], z[2];   // x index, y index * S2, z index * S1 * S2
  int base[2][2];         // first element of the x pair of row (cz, cy)
  bool second[2][2][2];   // corner (cz, cy, cx) is the pair's second element
  __device__ __forceinline__ CornerOffsets(const Taps<DIM, PAD>& t, const Dims& d) {
    x[0] = min(max(t.x.i0, 0), d.s2 - 1);
    x[1] = min(max(t.x.i0 + 1, 0), d.s2 - 1);
    const int xa = min(x[0], d.s2 - 2);                    // (-1 for rows of one voxel)
    y[0] = min(max(t.y.i0, 0), d.s1 - 1) * d.s2;
    y[1] = min(max(t.y.i0 + 1, 0), d.s1 - 1) * d.s2;
    if (DIM == 3) {
      z[0] = min(max(t.z.i0, 0), d.s0 - 1) * (d.s1 * d.s2);
      z[1] = min(max(t.z.i0 + 1, 0), d.s0 - 1) * (d.s1 * d.s2);
    } else {
      z[0] = z[1] = 0;
    }
    if constexpr (PAIRED) {
#pragma unroll
      for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
          base[cz][cy] = max(z[cz] + y[cy] + xa, 0);
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) second[cz][cy][cx] = z[cz] + y[cy] + x[cx] != base[cz][cy];
        }
    }
  }
  __device__ __forceinline__ int at(int cz, int cy, int cx) const { return z[cz] + y[cy] + x[cx]; }
  // all corner values, v[(cz * 2 + cy) * 2 + cx], loaded unconditionally and back to back
  __device__ __forceinline__ void load(const float* __restrict__ in, float (&v)[8]) const {
#pragma unroll
    for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy) {
        if constexpr (PAIRED) {
          const FloatPair p = *reinterpret_cast<const FloatPair*>(in + base[cz][cy]);
          v[(cz * 2 + cy) * 2 + 0] = second[cz][cy][0] ? p.hi : p.lo;
          v[(cz * 2 + cy) * 2 + 1] = second[cz][cy][1] ? p.hi : p.lo;
        } else {
          v[(cz * 2 + cy) * 2 + 0] = in[at(cz, cy, 0)];
          v[(cz * 2 + cy) * 2 + 1] = in[at(cz, cy, 1)];
        }
      }
  }
};

// One term of a (bi/tri)linear tap sum, with the rounding PINNED so that every forward kernel (direct gathers, LDS tiles,
// z-marching ring) produces the same bits -- `acc += v * w` is contracted to an fma in one kernel and not in another, and
// the displacement hint that picks the 3D kernel comes from asynchronous read-backs: results depended on timing at the
// ulp level, amplified 2^8-fold by the squaring chain.  3D: product, then sum (what ATen's scalar CPU loop does: the
// 9-squaring golden field is reproduced to 1.6e-5 with it, to 3e-5 with the fma chain); 2D: fma (the form the 2D goldens
// were pinned with).
template <int DIM>
__device__ __forceinline__ float tap_acc(float acc, float v, float w) {
  return DIM == 3 ? acc + mul_nc(v, w) : fmaf(v, w, acc);
}

template <int DIM, int PAD, bool PAIRED = true>
__device__ __forceinline__ float sample_linear(const float* __restrict__ in, const Taps<DIM, PAD>& t, const Dims& d) {
  const CornerOffsets<DIM, PAD, PAIRED> o(t, d);
  float v[8];
  o.load(in, v);
  float acc = 0.f;
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx)   // a select, not a branch: control flow here would make the next sample's loads wait
        acc = tap_acc<DIM>(acc, t.ok(cz, cy, cx) ? v[(cz * 2 + cy) * 2 + cx] : 0.f, t.w(cz, cy, cx));
  return acc;
}

// scatter go*w into gin and accumulate d(out)/d(unnormalised coordinate) * go into (ax, ay, az)
template <int DIM, int PAD, bool NEED_GIN, bool NEED_GGRID>
__device__ __forceinline__ void sample_linear_bwd(const float* __restrict__ in, float* __restrict__ gin, float go,
                                                  const Taps<DIM, PAD>& t, const Dims& d, float& ax, float& ay,
                                                  float& az) {
  float v[8];
  if (NEED_GGRID) {   // corner values first, unconditionally (see CornerOffsets)
    const CornerOffsets<DIM, PAD> o(t, d);
    o.load(in, v);
  }
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        const bool ok = t.ok(cz, cy, cx);
        if (NEED_GIN) {
          if (ok) atomic_add_f32(gin + t.off(cz, cy, cx, d), t.w(cz, cy, cx) * go);
        }
        if (NEED_GGRID) {
          // a select on the value, no branch: the loads of the next channel / sample must not wait behind control flow
          const float val = ok ? v[(cz * 2 + cy) * 2 + cx] : 0.f;
          if (DIM == 3) {
            ax += (cx ? 1.f : -1.f) * (val * t.wy(cy) * t.wz(cz) * go);
            ay += (cy ? 1.f : -1.f) * (val * t.wx(cx) * t.wz(cz) * go);
            az += (cz ? 1.f : -1.f) * (val * t.wx(cx) * t.wy(cy) * go);
          } else {
            ax += (cx ? 1.f : -1.f) * (val * t.wy(cy) * go);
            ay += (cy ? 1.f : -1.f) * (val * t.wx(cx) * go);
          }
        }
      }
}

// The coordinate-path part of sample_linear_bwd on corner values that are already in registers (CornerOffsets::load): a
// caller with several channels requests the corners of all of them first, so that their round trips overlap (per channel
// -- load, wait, arithmetic, next channel -- they were serial: 96 loads and 33 full waits in the 3D window scatter).
template <int DIM, int PAD>
__device__ __forceinline__ void sample_linear_bwd_values(const float (&v)[8], float go, const Taps<DIM, PAD>& t, float& ax,
                                                         float& ay, float& az) {
#pragma unroll
  for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        const float val = t.ok(cz, cy, cx) ? v[(cz * 2 + cy) * 2 + cx] : 0.f;
        if (DIM == 3) {
          ax += (cx ? 1.f : -1.f) * (val * t.wy(cy) * t.wz(cz) * go);
          ay += (cy ? 1.f : -1.f) * (val * t.wx(cx) * t.wz(cz) * go);
          az += (cz ? 1.f : -1.f) * (val * t.wx(cx) * t.wy(cy) * go);
        } else {
          ax += (cx ? 1.f : -1.f) * (val * t.wy(cy) * go);
          ay += (cy ? 1.f : -1.f) * (val * t.wx(cx) * go);
        }
      }
}

// affine warp: grid = theta_n * (x, y[, z], 1) evaluated in registers (F.affine_grid, align_corners=True)
template <int DIM>
struct Theta { float m[DIM][DIM + 1]; };

template <int DIM>
__device__ __forceinline__ void affine_position_xyz(const Theta<DIM>& th, int ix, int iy, int iz, const Dims& d,
                                                    float& bx, float& by, float& bz, float& gx, float& gy, float& gz) {
  bx = affine_base_coord(ix, d.s2);
  by = affine_base_coord(iy, d.s1);
  if constexpr (DIM == 3) {
    bz = affine_base_coord(iz, d.s0);
    gx = th.m[0][0] * bx + th.m[0][1] * by + th.m[0][2] * bz + th.m[0][3];
    gy = th.m[1][0] * bx + th.m[1][1] * by + th.m[1][2] * bz + th.m[1][3];
    gz = th.m[DIM - 1][0] * bx + th.m[DIM - 1][1] * by + th.m[DIM - 1][2] * bz + th.m[DIM - 1][DIM];
  } else {
    bz = 0.f; gz = 0.f;
    gx = th.m[0][0] * bx + th.m[0][1] * by + th.m[0][2];
    gy = th.m[1][0] * bx + th.m[1][1] * by + th.m[1][2];
  }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) r[k] = p[k];
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) p[k] = r[k];
  }
}

// Compiler-only ordering point for a wave's private LDS scratch: the hardware executes one wave's LDS instructions in
// order, but the compiler may move a 16-byte read of the scratch across the 4-byte writes that fill it (it treats the
// two access types as non-aliasing: seen as channel 2's values stored for channel 0 in sample_march.hip).
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ float clamp_unit(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

// displacement of a sampling position from its own voxel, in voxels (NaN -> ignored by the max, inf -> capped)
__device__ __forceinline__ float voxel_displacement(float g, int S, int s) {
  const float dv = fabsf(((g + 1.f) * 0.5f) * (float)(S - 1) - (float)s);
  return dv < 1.0e9f ? dv : 1.0e9f;
}

// wave maximum of a non-negative value -> one of kDispSlots float slots (atomic max on the bit pattern).  Measured: tens of thousands of waves max-ing into 64 slots that start at zero cost 100+ us per launch (every
// wave sees a stale smaller value and issues its atomic; contended same-address atomics across XCDs are far slower than
// the uncontended 10 ns) -- with 4096 slots a slot sees a handful of waves.  Callable by partially active waves.
constexpr int kDispSlots = 4096;
__device__ __forceinline__ void wave_max_to_slots(float m, float* __restrict__ slots) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const unsigned blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* p = slots + (blk * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kDispSlots;
  // fire and forget (no read-before: a dependent load would keep every wave alive for another memory round trip)
  if ((threadIdx.x & 63) == __builtin_ffsll(__builtin_amdgcn_ballot_w64(true)) - 1 && m > 0.f)
    atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(m));
}


}  // namespace advchain
