// which lane does a DPP row shift read?  prints lane 5's view
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int x = threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x101, 0xf, 0xf, true);        // row_shl:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x102, 0xf, 0xf, true);  // row_shl:2
}
int main() {
  int* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_shl:1 lane5<-%d lane15<-%d lane0<-%d | row_shr:1 lane5<-%d lane0<-%d | row_shl:2 lane5<-%d\n", h[5], h[15], h[0], h[64 + 5], h[64], h[128 + 5]);
  return 0;
}
