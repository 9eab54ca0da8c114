// Micro-benchmark: does independent VALU work issued by the SAME wave overlap with its fp32 MFMAs on gfx950 (one wave per SIMD)?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/debug/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one MFMA of the kind under test: BF = 0 v_mfma_f32_32x32x2_f32 (64 cycles), BF = 1 v_mfma_f32_32x32x16_bf16 (8x the flops)
template <int BF>
__device__ __forceinline__ f32x16 mf(float a, float b, bf16x8 ah, bf16x8 bh, f32x16 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int NM, int NV, int TRANS, int BF = 0>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed) {
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(seed + i); bh[i] = (__bf16)(seed * 0.5f + i); }
  f32x16 acc[3];
  for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m % 3] = mf<BF>(a, b, ah, bh, acc[m % 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        if (TRANS) v[q & 7] = __builtin_amdgcn_rcpf(v[q & 7] + 1.0f);
        else v[q & 7] = v[q & 7] * 1.0001f + 0.5f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// two waves per SIMD (512 threads): waves 0-3 run only the MFMA groups, waves 4-7 only the VALU groups (WHO = 3: both kinds of
// waves active; 1: only the MFMA waves work; 2: only the VALU waves work)
template <int NM, int NV, int WHO, int BF = 0>
__global__ __launch_bounds__(512, 1) void k2(float* out, int iters, float seed) {
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(seed + i); bh[i] = (__bf16)(seed * 0.5f + i); }
  f32x16 acc[3];
  for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    if (WHO & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
          for (int m = 0; m < NM; ++m) acc[m % 3] = mf<BF>(a, b, ah, bh, acc[m % 3]);
        }
      }
  } else {
    if (WHO & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
          for (int q = 0; q < NV; ++q) v[q & 7] = v[q & 7] * 1.0001f + 0.5f;
        }
      }
  }
  float s = 0.f;
  for (int g = 0; g < 3; ++g) for (int i = 0; i < 16; ++i) s += acc[g][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int WHO, int BF = 0>
void run2(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k2<NM, NV, WHO, BF>), dim3(256), dim3(512), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k2<NM, NV, WHO, BF>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("2 waves/SIMD %-22s %s NM=%d NV=%d who=%d: %.3f ms, %.1f ns per group\n", name, BF ? "bf16" : "fp32", NM, NV, WHO, ms, ms * 1e6 / iters / 16);
}

template <int NM, int NV, int TRANS, int BF = 0>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<NM, NV, TRANS, BF>), dim3(256), dim3(256), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, TRANS, BF>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns_group = ms * 1e6 / iters / 16;
  printf("1 wave/SIMD  %-22s %s NM=%d NV=%d trans=%d: %.3f ms, %.1f ns per (MFMA group + VALU group)\n", name, BF ? "bf16" : "fp32", NM, NV, TRANS, ms, ns_group);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<3, 0, 0>("3 MFMA only", out);
  run<0, 12, 0>("12 FMA only", out);
  run<3, 12, 0>("3 MFMA + 12 FMA", out);
  run<3, 24, 0>("3 MFMA + 24 FMA", out);
  run<3, 40, 0>("3 MFMA + 40 FMA", out);
  run<0, 6, 1>("6 (add+rcp) only", out);
  run<3, 6, 1>("3 MFMA + 6 (add+rcp)", out);
  run<12, 0, 0>("12 MFMA only", out);
  run<12, 48, 0>("12 MFMA + 48 FMA", out);
  run<3, 0, 0, 1>("3 MFMA only", out);
  run<3, 6, 0, 1>("3 MFMA + 6 FMA", out);
  run<3, 12, 0, 1>("3 MFMA + 12 FMA", out);
  run<3, 24, 0, 1>("3 MFMA + 24 FMA", out);
  run2<3, 20, 1>("MFMA waves only", out);
  run2<3, 20, 2>("VALU waves only", out);
  run2<3, 20, 3>("both", out);
  run2<3, 60, 2>("VALU waves only", out);
  run2<3, 60, 3>("both", out);
  run2<3, 10, 1, 1>("MFMA waves only", out);
  run2<3, 10, 2, 1>("VALU waves only", out);
  run2<3, 10, 3, 1>("both", out);
  run2<3, 30, 2, 1>("VALU waves only", out);
  run2<3, 30, 3, 1>("both", out);
  return 0;
}
