// lane-exchange semantics check for gfx950: v_permlane{16,32}_swap and DPP row mirrors (used by csrc/act_fused.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__global__ void k(float* o) {
  float v = (float)threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  o[threadIdx.x] = __int_as_float(r[0]); o[64 + threadIdx.x] = __int_as_float(r[1]);
  auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  o[128 + threadIdx.x] = __int_as_float(r2[0]); o[192 + threadIdx.x] = __int_as_float(r2[1]);
  o[256 + threadIdx.x] = dpp<0x140>(v); o[320 + threadIdx.x] = dpp<0x141>(v);
  o[384 + threadIdx.x] = dpp<0x4E>(v); o[448 + threadIdx.x] = dpp<0xB1>(v);
}
int main() {
  float* d; hipMalloc(&d, 512 * 4); k<<<1, 64>>>(d); float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"pl32 r0", "pl32 r1", "pl16 r0", "pl16 r1", "row_mirror", "half_mirror", "qp xor2", "qp xor1"};
  for (int j = 0; j < 8; ++j) { printf("%-11s:", names[j]); for (int i = 0; i < 64; ++i) printf(" %d", (int)h[64 * j + i]); printf("\n"); }
  return 0;
}
