// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC gbench.hip -o libgbench.so ; run: python gbench.py
// gather cost probe: near-identity 2D bilinear taps, planar dword gathers (8 per voxel) vs interleaved dwordx2 (4 per voxel)
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) k_planar(const float* __restrict__ phi, float* __restrict__ out, int H, int W) {
  const int V = H * W, n = blockIdx.y;
  const float* p = phi + (size_t)n * 2 * V;
  float* o = out + (size_t)n * 2 * V;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    const int y = v / W, x = v - y * W;
    const float fx = p[v], fy = p[V + v];
    const float px = fminf(fmaxf(x + fx, 0.f), W - 1.001f), py = fminf(fmaxf(y + fy, 0.f), H - 1.001f);
    const int ix = (int)px, iy = (int)py;
    const float wx = px - ix, wy = py - iy;
    const int b = iy * W + ix;
    const float a00 = p[b], a01 = p[b + 1], a10 = p[b + W], a11 = p[b + W + 1];
    const float c00 = p[V + b], c01 = p[V + b + 1], c10 = p[V + b + W], c11 = p[V + b + W + 1];
    o[v] = (a00 * (1 - wx) + a01 * wx) * (1 - wy) + (a10 * (1 - wx) + a11 * wx) * wy;
    o[V + v] = (c00 * (1 - wx) + c01 * wx) * (1 - wy) + (c10 * (1 - wx) + c11 * wx) * wy;
  }
}
__global__ void __launch_bounds__(256) k_inter(const float2* __restrict__ phi, float2* __restrict__ out, int H, int W) {
  const int V = H * W, n = blockIdx.y;
  const float2* p = phi + (size_t)n * V;
  float2* o = out + (size_t)n * V;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < V; v += gridDim.x * 256) {
    const int y = v / W, x = v - y * W;
    const float2 f = p[v];
    const float px = fminf(fmaxf(x + f.x, 0.f), W - 1.001f), py = fminf(fmaxf(y + f.y, 0.f), H - 1.001f);
    const int ix = (int)px, iy = (int)py;
    const float wx = px - ix, wy = py - iy;
    const int b = iy * W + ix;
    const float2 a00 = p[b], a01 = p[b + 1], a10 = p[b + W], a11 = p[b + W + 1];
    float2 r;
    r.x = (a00.x * (1 - wx) + a01.x * wx) * (1 - wy) + (a10.x * (1 - wx) + a11.x * wx) * wy;
    r.y = (a00.y * (1 - wx) + a01.y * wx) * (1 - wy) + (a10.y * (1 - wx) + a11.y * wx) * wy;
    o[v] = r;
  }
}
extern "C" int g_run(int mode, const float* phi, float* out, int N, int H, int W, int blocks, void* st) {
  if (mode == 0) hipLaunchKernelGGL(k_planar, dim3(blocks, N), dim3(256), 0, (hipStream_t)st, phi, out, H, W);
  else hipLaunchKernelGGL(k_inter, dim3(blocks, N), dim3(256), 0, (hipStream_t)st, (const float2*)phi, (float2*)out, H, W);
  return (int)hipGetLastError();
}
