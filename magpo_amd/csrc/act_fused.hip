// C entry point of the fused acting kernel (the kernel itself: act_fused_kernel.hpp; its instances: act_fused_epw{4,8,16}.hip).
#include "act_fused_kernel.hpp"

using namespace magpo;

#ifndef MAGPO_ACT_PROF   // (the profiling build keeps its counters in one translation unit and instantiates everything here)
extern template void launch_act<4>(const magpo::ActArgs&, hipStream_t);
extern template void launch_act<8>(const magpo::ActArgs&, hipStream_t);
extern template void launch_act<16>(const magpo::ActArgs&, hipStream_t);
#endif

// Envs per wave of the launch magpo_sable_act makes for (N envs, A agents); forced = 0: by size, 4 / 8 / 16: the caller's choice.
// Exported so that tests and benches can NAME the instance they ran (k_sable_act<epw, A <= 4 ? 4 : 8, n_head == 1 && A <= 4>).
// envs per wave, measured per launch on MI355X with device events (scripts/debug/act_epw_small.py, act_epw.py; A = 4, one block, 4 / 8 / 16
// envs per wave: N = 1024 -> 204 / 311 / 335 us, 2048 -> 237 / 331 / 348, 4096 -> 327 / 396 / 361, 8192 -> 606 / 553 / 434, 16384 -> 1360 / 922 / 698;
// A = 8, two blocks (before the one-wave bound of the 4-env waves): N = 1024 -> 1114 / 1363 / 1882, 4096 -> 1691 / 1679 / 2088, 16384 -> 5212 / 4905 / 2340): few envs per wave while the
// waves fit the chip's 1024 SIMDs (a rollout step is a latency chain per wave), full MFMA tiles and less weight traffic once they do not.
// All variants run at one wave per SIMD (512 registers, no scratch).
extern "C" int magpo_sable_act_envs_per_wave(int N, int A, int forced) {
  if (forced) {   // envs per wave forced by the caller (A/B measurements, parity tests of every instance)
    if (forced != 4 && forced != 8 && forced != 16) { set_error("magpo_sable_act: envs per wave must be 0, 4, 8 or 16"); return -1; }
    return forced;
  }
  // Round 4 (staggered / deferred candidate pre-pass; profiles/r04_act_kernel_launch_times.txt, us per launch for 4 / 8 / 16 envs per wave):
  // A = 4, one block: N = 2048 -> 218 / 250 / -, 4096 -> 244 / 260 / 328, 8192 -> - / 309 / 349, 16384 -> 934 / 606 / 482; A = 8, two blocks:
  // 4096 -> 945 / 1080 / 1439, 16384 -> - / - / 1904.  I.e. the fewest envs per wave whose waves still fit the chip's 1024 SIMDs at once.
  (void)A;
  return N <= 4096 ? 4 : (N <= 8192 ? 8 : 16);
}

// Fragment-major copy of a transposed weight for the acting kernel (fm_rows.hpp: wfrag<true>):
//   Wf[g][gk][lane = m + 16 kq][c] = Wt[16 g + m][16 gk + 4 kq + c],   g < nrows / 16, gk / kq / c < 4, m < 16
__global__ void k_act_weight_layout(const float* __restrict__ Wt, float* __restrict__ Wf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = i >> 10, r = i & 1023, gk = r >> 8, lane = (r & 255) >> 2, c = r & 3, m = lane & 15, kq = lane >> 4;
  Wf[i] = Wt[(16 * g + m) * 64 + 16 * gk + 4 * kq + c];
}
extern "C" int magpo_act_weight_layout(const float* Wt, float* Wf, int nrows, hipStream_t st) {
  if (nrows <= 0 || (nrows & 15)) { set_error("magpo_act_weight_layout: nrows must be a positive multiple of 16"); return MAGPO_EINVAL; }
  const int n = nrows * 64;
  hipLaunchKernelGGL(k_act_weight_layout, dim3((n + 255) / 256), dim3(256), 0, st, Wt, Wf, n);
  return check_launch("magpo_act_weight_layout");
}

// Pointer tables (host arrays of device pointers) keep the boundary plain C without a shared struct layout:
//   dims_host[16] = {N, A, K, F, n_block, n_head, hs, gs, npos, value_only, obs row stride, envs per wave, pending, flush, precand, defer}; kappa_host[4];
//   keys_host [A][2] or NULL (then ptrs[3] = device key table);  ptrs_host[49] / blk_ptrs_host[21 * n_block] in the order of the P(...) lists below.
extern "C" int magpo_sable_act(const int* dims_host, const float* kappa_host, const uint32_t* keys_host, const void* const* ptrs,
                               int nptrs, const void* const* blk_ptrs, int nblk_ptrs, hipStream_t st) {
  ActArgs a;
  memset(&a, 0, sizeof(a));
  a.N = dims_host[0]; a.A = dims_host[1]; a.K = dims_host[2]; a.F = dims_host[3]; a.nb = dims_host[4]; a.nh = dims_host[5];
  a.hs = dims_host[6]; a.gs = dims_host[7]; a.npos = dims_host[8]; a.value_only = dims_host[9]; a.ldo = dims_host[10];
  a.pending = dims_host[12] != 0; a.flush = dims_host[13] != 0; a.precand = dims_host[14] != 0; a.defer = dims_host[15] != 0;
  if (a.N <= 0) return MAGPO_OK;
  if (a.A < 1 || a.A > MAXA || a.nb < 1 || a.nb > MAXB || a.nh < 1 || a.nh > 4 || a.K < 1 || a.K > 31 || a.F < 1 || a.hs * a.nh != AE ||
      a.gs < 4 || a.gs > a.hs || (a.gs & (a.gs - 1)) || a.npos < 1 || a.ldo < a.F) {
    set_error("magpo_sable_act: unsupported shape (1 <= A <= 8, n_block <= 4, n_head in {1,2,4}, K <= 31)");
    return MAGPO_EINVAL;
  }
  if (a.defer && (a.flush || a.value_only)) { set_error("magpo_sable_act: defer (candidate pass of the next step at the end of the launch) excludes flush / value_only"); return MAGPO_EINVAL; }
  if (a.precand && !a.pending) { set_error("magpo_sable_act: precand needs pending (the previous launch of the rollout left rows)"); return MAGPO_EINVAL; }
  if (nptrs != 49 || nblk_ptrs != 21 * a.nb) { set_error("magpo_sable_act: pointer table size mismatch"); return MAGPO_EINVAL; }
  for (int i = 0; i < 4; ++i) a.kappa[i] = kappa_host[i];
  if (keys_host) for (int i = 0; i < a.A; ++i) { a.keys[i][0] = keys_host[2 * i]; a.keys[i][1] = keys_host[2 * i + 1]; }
  int p = 0;
#define P(T, f) a.f = (T)ptrs[p++];
  P(const float*, obs) P(const int*, pos) P(const unsigned char*, mask) P(const uint32_t*, keys_dev)
  P(const float*, s_obs) P(const float*, W_obs) P(const float*, s_encln) P(const float*, W_act) P(const float*, s_decln)
  P(const float*, vh0_t) P(const float*, vh0_b) P(const float*, vh_s) P(const float*, vh_w) P(const float*, vh_b1)
  P(const float*, h0_t) P(const float*, h0_b) P(const float*, h_s) P(const float*, h1_t) P(const float*, h1_b)
  P(const float*, pe)
  P(float*, S_enc) P(float*, S_d1) P(float*, S_d2)
  P(float*, xn) P(const unsigned char*, done) P(float*, qkvg) P(float*, u) P(float*, y) P(float*, rep) P(float*, reppe) P(float*, hv)
  P(float*, xa) P(float*, kin1) P(float*, y1) P(float*, c) P(float*, cpe) P(float*, y2) P(float*, xo) P(float*, xope) P(float*, hp)
  P(float*, hn) P(float*, logits) P(float*, u1) P(float*, u2) P(int*, prev)
  P(int*, action) P(float*, logp) P(float*, value) P(float*, ptab)
#undef P
  if (!a.value_only && a.nh == 1 && !a.ptab) { set_error("magpo_sable_act: the candidate table (ptrs[48]) is required for n_head = 1"); return MAGPO_EINVAL; }
  if (!keys_host && !a.keys_dev && !a.value_only) { set_error("magpo_sable_act: no sampling keys"); return MAGPO_EINVAL; }
  for (int b = 0; b < a.nb; ++b) {
    const void* const* q = blk_ptrs + 21 * b;
    ActBlk& B = a.blk[b];
    B.qkvg_t = (const float*)q[0]; B.wo_t = (const float*)q[1]; B.ln1 = (const float*)q[2]; B.ln2 = (const float*)q[3];
    B.gn_g = (const float*)q[4]; B.gn_b = (const float*)q[5];
    B.qkvg1_t = (const float*)q[6]; B.wo1_t = (const float*)q[7]; B.dln1 = (const float*)q[8]; B.gn1_g = (const float*)q[9]; B.gn1_b = (const float*)q[10];
    B.q2_t = (const float*)q[11]; B.kvg2_t = (const float*)q[12]; B.wo2_t = (const float*)q[13]; B.dln2 = (const float*)q[14]; B.dln3 = (const float*)q[15];
    B.gn2_g = (const float*)q[16]; B.gn2_b = (const float*)q[17];
    B.qkvg1 = (float*)q[18]; B.q2 = (float*)q[19]; B.kvg2 = (float*)q[20];
  }
  const int epw = magpo_sable_act_envs_per_wave(a.N, a.A, dims_host[11]);
  if (epw < 0) return MAGPO_EINVAL;
  if (epw == 16) launch_act<16>(a, st); else if (epw == 4) launch_act<4>(a, st); else launch_act<8>(a, st);
  return check_launch("magpo_sable_act");
}

#ifdef MAGPO_ACT_PROF
extern "C" int magpo_debug_act_prof(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(magpo::g_act_prof), sizeof(unsigned long long) * 32) != hipSuccess) return MAGPO_ELAUNCH;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(magpo::g_act_prof), z, sizeof(z)) != hipSuccess) return MAGPO_ELAUNCH; }
  return MAGPO_OK;
}
#endif
