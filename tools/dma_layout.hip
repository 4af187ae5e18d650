// Dev: where does LDS-DMA put each lane's data for the dword / dwordx3 / ubyte forms on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* g, const unsigned char* b, float* out) {
  __shared__ float l[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) l[i] = -1.f;
  __syncthreads();
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) float*)l);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx3 %1, off" ::"s"(la), "v"(g + 3 * threadIdx.x) : "memory");
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_ubyte %1, off" ::"s"(la + 2048), "v"(b + threadIdx.x) : "memory");
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" ::"s"(la + 3072), "v"(g + 3 * threadIdx.x) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = l[i];
}
int main() {
  float h[192]; unsigned char hb[64]; for (int i = 0; i < 192; ++i) h[i] = (float)i; for (int i = 0; i < 64; ++i) hb[i] = (unsigned char)(i + 100);
  float *g, *o; unsigned char* b; hipMalloc(&g, sizeof h); hipMalloc(&b, 64); hipMalloc(&o, 4096);
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(b, hb, 64, hipMemcpyHostToDevice);
  k<<<1, 64>>>(g, b, o); float r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  printf("dwordx3, first 16 floats:"); for (int i = 0; i < 16; ++i) printf(" %g", r[i]); printf("\n ... floats 186..200:"); for (int i = 186; i < 200; ++i) printf(" %g", r[i]); printf("\n");
  printf("ubyte, first 8 dwords:"); for (int i = 0; i < 8; ++i) printf(" %u", ((unsigned*)r)[512 + i]); printf("  dword 63: %u  dword 64: %d\n", ((unsigned*)r)[512 + 63], ((int*)r)[512 + 64]);
  printf("dword, first 8:"); for (int i = 0; i < 8; ++i) printf(" %g", r[768 + i]); printf("\n");
  return 0;
}
