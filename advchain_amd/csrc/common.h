// Shared device helpers for the advchain HIP kernels (gfx950 / CDNA4, wave64).
//
// Conventions used by every kernel in this directory
//   * tensors are contiguous fp32, channels-first: (N, C, S0, S1, S2); 2D tensors use S0 = 1.
//   * sampling grids are PLANAR channels-first (N, d, S0, S1, S2) -- the reference keeps them that
//     way (adv_morph.py:14-55) and only permutes a strided view at the F.grid_sample call site; we
//     never materialise the channels-last copy.  Grid channel 0 = x <-> S2 (fastest), 1 = y <-> S1,
//     2 = z <-> S0, exactly F.grid_sample's convention.
//   * align_corners=True everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace advchain {

enum { INTERP_LINEAR = 0, INTERP_NEAREST = 1 };
enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_REFLECTION = 2 };

constexpr int kBlock = 256;  // 4 waves of 64

struct Dims {
  int s0, s1, s2;  // s0 == 1 for 2D
  __host__ __device__ int64_t voxels() const { return (int64_t)s0 * s1 * s2; }
};

// ---------------------------------------------------------------------------------------------
// torch.linspace(-1, 1, S)[i]  (ATen RangeFactories: symmetric two-sided formula).  The identity
// sampling grid of adv_morph.py:14-55 and F.affine_grid's base grid are both built from it; we
// evaluate it in registers instead of reading a materialised (N,d,...) base grid from HBM.
// ---------------------------------------------------------------------------------------------
// A product that is never contracted into an fma with a neighbouring sum.  HIP compiles with -ffp-contract=fast-honor-
// pragmas and `acc += v * w` / `t * top - floor(..)` are fused in one kernel and not in another (it depends on the
// surrounding code); mul_nc() is a plain `*` on this toolchain and fuses as well.  The forward samplers (direct gathers,
// LDS tiles, z-marching ring) must agree BIT FOR BIT -- the displacement hint that picks the 3D kernel comes from
// asynchronous read-backs, so anything else makes results depend on timing, 2^8-fold amplified by the squaring chain.
__device__ __forceinline__ float mul_nc(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

__device__ __forceinline__ float lin_coord(int i, int S) {
  if (S <= 1) return S == 1 ? -1.f : 0.f;
  const float step = 2.f / (float)(S - 1);
  return (i < S / 2) ? (-1.f + step * (float)i) : (1.f - step * (float)(S - 1 - i));
}
// F.affine_grid's base grid differs only for S == 1 (it uses 0 there)
__device__ __forceinline__ float affine_base_coord(int i, int S) {
  if (S <= 1) return 0.f;
  return lin_coord(i, S);
}

// ---------------------------------------------------------------------------------------------
// One axis of a (bi/tri)linear tap: ATen GridSampler.h:26-203 semantics.
// ---------------------------------------------------------------------------------------------
struct AxisTap {
  int i0;       // lower corner index (may be out of range for zeros padding)
  float w0;     // weight of corner i0      = (i0 + 1) - x
  float w1;     // weight of corner i0 + 1  = x - i0
  float mult;   // d(unnormalised, padded coordinate) / d(normalised grid value)
  bool v0, v1;  // corner inside [0, S)
};

__device__ __forceinline__ float reflect_coord(float x, int twice_low, int twice_high, float& g) {
  if (twice_low == twice_high) { g = 0.f; return 0.f; }
  const float mn = (float)twice_low * 0.5f;
  const float span = (float)(twice_high - twice_low) * 0.5f;
  x -= mn;
  float s = 1.f;
  if (x < 0.f) { s = -1.f; x = -x; }
  const float extra = fmodf(x, span);
  const int flips = (int)floorf(x / span);
  if ((flips & 1) == 0) { g = s; return extra + mn; }
  g = -s;
  return span - extra + mn;
}

template <int PAD>
__device__ __forceinline__ float source_index(float coord, int S, float& mult) {
  float x = ((coord + 1.f) * 0.5f) * (float)(S - 1);  // grid_sampler_unnormalize, align_corners
  mult = 0.5f * (float)(S - 1);
  if (PAD == PAD_BORDER) {
    // clip_coordinates_set_grad: zero gradient AT and beyond the border (inclusive)
    if (x <= 0.f) { x = 0.f; mult = 0.f; }
    else if (x >= (float)(S - 1)) { x = (float)(S - 1); mult = 0.f; }
  } else if (PAD == PAD_REFLECTION) {
    float g;
    x = reflect_coord(x, 0, 2 * (S - 1), g);
    mult *= g;
    if (x <= 0.f) { x = 0.f; mult = 0.f; }
    else if (x >= (float)(S - 1)) { x = (float)(S - 1); mult = 0.f; }
  }
  return x;
}

template <int PAD>
__device__ __forceinline__ AxisTap make_tap(float coord, int S) {
  AxisTap t;
  float x = source_index<PAD>(coord, S, t.mult);
  if (!(x > -1.0e9f && x < 1.0e9f)) x = -16.f;  // NaN / huge: every corner out of range
  const float f = floorf(x);
  t.i0 = (int)f;
  t.w1 = x - f;
  t.w0 = (f + 1.f) - x;
  t.v0 = (t.i0 >= 0) && (t.i0 < S);
  t.v1 = (t.i0 + 1 >= 0) && (t.i0 + 1 < S);
  return t;
}

// nearest: nearbyint (round-half-to-even) like ATen
template <int PAD>
__device__ __forceinline__ int nearest_index(float coord, int S, bool& valid) {
  float m;
  float x = source_index<PAD>(coord, S, m);
  if (!(x > -1.0e9f && x < 1.0e9f)) x = -16.f;
  const int i = (int)nearbyintf(x);
  valid = (i >= 0) && (i < S);
  return i;
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum NV values across a 256-thread block.  Result valid in thread 0.  `smem` >= 4*NV floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) smem[wave * NV + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += smem[w * NV + k];
      v[k] = s;
    }
  }
}

// whole-wave lane shifts (gfx9 DPP wave_shl / wave_shr: one VALU op, no LDS); lanes shifted in receive 0
__device__ __forceinline__ float lane_next_f(float v) {   // lane i <- lane i+1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_prev_f(float v) {   // lane i <- lane i-1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// Per-workgroup partial sums go to one of kSumSlots accumulators (slot = workgroup id mod kSumSlots): thousands of
// atomics on ONE address serialise at ~10 ns each on MI355X; the consumer adds the slots up.
constexpr int kSumSlots = 64;
__device__ __forceinline__ int sum_slot() { return (int)((blockIdx.x + blockIdx.y * 7u) % kSumSlots); }

// float -> 32-bit fixed point for the LDS-integer scatters (round 6): v_cvt_rpi_i32_f32 -- floor(x + 0.5) in ONE instruction;
// the compiler only selects it under -ffast-math -- instead of the v_rndne_f32 + v_cvt_i32_f32 pair of __float2int_rn: 8 of
// the 134 VALU instructions of a visited row in the wide march scatter, 2 C per item in the whole-row scatter, 8 C per sample in
// the march / window scatters.  Exact halves round up instead of to even (one unit of the fixed-point resolution, the same on
// every run).  A/B build switch: -DADVCHAIN_FIX_RPI=0.
#ifndef ADVCHAIN_FIX_RPI
#define ADVCHAIN_FIX_RPI 1
#endif
__device__ __forceinline__ int fix_round(float x) {
#if ADVCHAIN_FIX_RPI
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return __float2int_rn(x);
#endif
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  // hardware global_atomic_add_f32 (no CAS loop); device memory only
  unsafeAtomicAdd(p, v);
}

}  // namespace advchain

// ---------------------------------------------------------------------------------------------
// host-side helpers shared by the C-ABI translation units
// ---------------------------------------------------------------------------------------------
#define ADVCHAIN_OK 0
#define ADVCHAIN_ERR_ARG (-1)
#define ADVCHAIN_ERR_UNSUPPORTED (-2)
#define ADVCHAIN_ERR_LAUNCH (-3)

extern "C" void advchain_set_error_(const char* msg);

#define ADVCHAIN_CHECK_ARG(cond, msg)      \
  do {                                     \
    if (!(cond)) {                         \
      advchain_set_error_(msg);            \
      return ADVCHAIN_ERR_ARG;             \
    }                                      \
  } while (0)

#define ADVCHAIN_LAUNCH_CHECK()                          \
  do {                                                   \
    hipError_t e_ = hipGetLastError();                   \
    if (e_ != hipSuccess) {                              \
      advchain_set_error_(hipGetErrorString(e_));        \
      return ADVCHAIN_ERR_LAUNCH;                        \
    }                                                    \
  } while (0)

static inline int advchain_blocks(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

// Zero-fill as a kernel of this library rather than hipMemsetAsync: under stream capture a memset becomes a
// memset node, which the runtime executes outside the kernel queue's own order (LESSONS 66) -- every node of a
// captured ascent is a kernel node.  `bytes` must be a multiple of 4 (every caller clears float arrays).
static __global__ void __launch_bounds__(256) k_zero_fill(float* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && ((uintptr_t)(p + i) & 15) == 0) {
      *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) p[j] = 0.f;
    }
  }
}
static inline void advchain_zero_async(void* p, size_t bytes, hipStream_t st) {
  const int64_t n = (int64_t)(bytes / 4);
  if (n <= 0) return;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, st, (float*)p, n);
}
