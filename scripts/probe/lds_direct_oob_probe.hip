// __builtin_amdgcn_raw_ptr_buffer_load_lds: out-of-range lanes write 0 to LDS (they do not skip the write).
// build: hipcc --offload-arch=gfx950 -O3 <this> -o scripts/probe/lds_direct_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int n) {
  __shared__ __attribute__((aligned(16))) float buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = -7.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n * 4, 0x00020000);
  int voff = (lane & 1) ? 0x7fffffff : lane * 4;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_ptr_t)(buf + wave * 64), 4, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  out[threadIdx.x] = buf[threadIdx.x];
}
int main() {
  float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = 100 + i;
  hipMemcpy(g, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, g, o, 256);
  float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("%g ", r[i]); printf(" | "); for (int i = 64; i < 72; ++i) printf("%g ", r[i]); printf("\n");
  return 0;
}
