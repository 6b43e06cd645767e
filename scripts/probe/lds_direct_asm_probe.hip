// LDS-direct loads (buffer_load_dword ... lds, global_load_lds_dword / _dwordx4) issued from inline assembly: destination =
// M0 + lane * size; out-of-range buffer lanes write 0.  build: hipcc --offload-arch=gfx950 -O3 <this> -o scripts/probe/lds_direct_asm_probe
#include <hip/hip_runtime.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_buf(unsigned lds_addr, int voff, i32x4 rsrc, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_g16(unsigned lds_addr, const void* gptr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(gptr) : "memory");
}
__device__ __forceinline__ void dma_g4(unsigned lds_addr, const void* gptr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(lds_addr), "v"(gptr) : "memory");
}
__global__ void k(const float* g, float* out, int n) {
  __shared__ __attribute__((aligned(16))) float buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = -7.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long ga = (unsigned long long)g;
  i32x4 rsrc = {(int)(unsigned)ga, (int)(unsigned)(ga >> 32), n * 4, 0x00020000};
  int voff = (lane & 1) ? 0x7fffffff : lane * 4;
  typedef __attribute__((address_space(3))) float* lds_f;
  const unsigned base = (unsigned)(size_t)(lds_f)buf;
  dma_buf(base + wave * 256, voff, rsrc, 0);
  dma_g16(base + 1024 * 4 + wave * 1024, g + lane * 4);
  dma_g4(base + 512 * 4 + wave * 256, g + 7 + lane);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = buf[i];
}
#include <stdio.h>
int main() {
  float *g, *o; hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 100 + i;
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, g, o, 256);
  static float r[2048]; hipMemcpy(r, o, 2048 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("%g ", r[i]); printf("| "); for (int i = 64; i < 70; ++i) printf("%g ", r[i]);
  printf("| g4: "); for (int i = 512; i < 516; ++i) printf("%g ", r[i]); printf("| "); for (int i = 576; i < 580; ++i) printf("%g ", r[i]);
  printf("| g16: "); for (int i = 1024; i < 1030; ++i) printf("%g ", r[i]); printf("| "); for (int i = 1280; i < 1286; ++i) printf("%g ", r[i]); printf("\n");
  return 0;
}
