// The two x corners of a trilinear sample as two 4-byte gathers or as ONE 8-byte gather (4-byte aligned address), 4 (z, y) rows per
// thread, 4 x 128 x 128 x 64 threads with offsets of -1 / 0 / +1 voxel:
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/pairbench.hip -o /tmp/pairbench && /tmp/pairbench
// MI355X: 19.6 us (8 dword gathers) against 14.4 us (4 dwordx2 gathers) -> CornerOffsets::load (advchain_amd/csrc/sampler_common.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
// 8 "corner pairs" per thread like a 3D trilinear gather: 4 (z,y) rows x {x, x+1}
template <bool PAIR>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, const int* __restrict__ off, float* __restrict__ out, int n, int S2, int S12, int one) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = off[i];
  float acc = 0.f;
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
      const int a = o + cz * S12 + cy * S2;
      if (PAIR) {
        const f2u v = *reinterpret_cast<const f2u*>(in + a);
        acc += v.x * 0.3f + v.y * 0.7f;
      } else {
        acc += in[a] * 0.3f + in[a + one] * 0.7f;
      }
    }
  out[i] = acc;
}
int main() {
  const int S2 = 64, S1 = 128, S0 = 128, N = 4;
  const int V = S2 * S1 * S0, n = N * V;
  float *in, *out; int* off;
  hipMalloc(&in, (size_t)(n + 2 * S2 * S1 + 8) * 4); hipMalloc(&out, (size_t)n * 4); hipMalloc(&off, (size_t)n * 4);
  int* h = (int*)malloc((size_t)n * 4);
  for (int i = 0; i < n; ++i) { int x = i % S2; int d = (rand() % 3) - 1; int xx = x + d; if (xx < 0) xx = 0; if (xx > S2 - 2) xx = S2 - 2; h[i] = i - x + xx; }
  hipMemcpy(off, h, (size_t)n * 4, hipMemcpyHostToDevice);
  hipMemset(in, 0, (size_t)(n + 2 * S2 * S1 + 8) * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pair = 0; pair < 2; ++pair) {
    for (int rep = 0; rep < 3; ++rep) { if (pair) k<true><<<(n + 255) / 256, 256>>>(in, off, out, n, S2, S2 * S1, atoi(getenv("ONE") ? getenv("ONE") : "1")); else k<false><<<(n + 255) / 256, 256>>>(in, off, out, n, S2, S2 * S1, atoi(getenv("ONE") ? getenv("ONE") : "1")); }
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) { if (pair) k<true><<<(n + 255) / 256, 256>>>(in, off, out, n, S2, S2 * S1, atoi(getenv("ONE") ? getenv("ONE") : "1")); else k<false><<<(n + 255) / 256, 256>>>(in, off, out, n, S2, S2 * S1, atoi(getenv("ONE") ? getenv("ONE") : "1")); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f us per launch\n", pair ? "dwordx2 pairs (4 loads)" : "dword corners (8 loads)", ms / 20 * 1e3);
  }
  return 0;
}
