// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC dispbench.hip -o libdispbench.so ; run: python dispbench.py
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(1024) k_disp(float* out, int work) {
  extern __shared__ float lds[];
  if (work) { lds[threadIdx.x] = threadIdx.x; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = work ? lds[1] : 1.f;
}
extern "C" int disp_run(float* out, int blocks, int threads, int ldsbytes, int work, void* st) {
  if (ldsbytes > 65536) (void)hipFuncSetAttribute((const void*)k_disp, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbytes);
  hipLaunchKernelGGL(k_disp, dim3(blocks), dim3(threads), ldsbytes, (hipStream_t)st, out, work);
  return (int)hipGetLastError();
}
