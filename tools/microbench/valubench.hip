// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC valubench.hip -o libvalubench.so ; run: python valubench.py
// VALU issue-rate probe: N dependent-free FMA streams per lane, wave64; compare plain vs packed fp32.
#include <hip/hip_runtime.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, int iters, float a, float b) {
  float x[8];
  float2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = float2v{x[i], x[i] + 1.f}; }
  const float2v a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
      else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], a2, b2);
      else if (MODE == 2) x[i] = fmaxf(0.f, 1.f - fabsf(x[i] - a));          // tent: sub, sub|abs|, max
      else if (MODE == 3) x[i] = __builtin_amdgcn_fmed3f(x[i], a, b);
      else if (MODE == 4) x[i] = floorf(x[i]) * a;
      else if (MODE == 5) { int q = __float_as_int(x[i]); q = q * 3 + 1; x[i] = __int_as_float(q); }   // int mul
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += MODE == 1 ? p[i].x + p[i].y : x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int valu_run(int mode, float* out, int blocks, int iters, void* st) {
  hipStream_t s = (hipStream_t)st;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
    case 1: hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
    case 2: hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
    case 3: hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
    case 4: hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
    default: hipLaunchKernelGGL(k_valu<5>, dim3(blocks), dim3(256), 0, s, out, iters, 1.0001f, 0.5f); break;
  }
  return (int)hipGetLastError();
}
