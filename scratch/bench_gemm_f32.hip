// Stand-alone harness for the fp32 (parity mode) GEMMs of k_gemm.hip: the 16x16x4 kernels of rounds 1-4 against the 32x32x2 kernels of
// round 5 on the shapes of one training step at batch 16 (no torch).  Checks both against an fp64 host evaluation on sampled outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_gemm_f32.hip -o scratch/bench_gemm_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../m2trans_amd/csrc/k_gemm.hip"
int m2t_set_hip_error(hipError_t e, const char* f, int l) { fprintf(stderr, "HIP error %d %s at %s:%d\n", (int)e, hipGetErrorString(e), f, l); return (int)e; }
int m2t_set_error(int c, const char* m) { fprintf(stderr, "error %d %s\n", c, m); return c; }
int m2t_ensure_dynamic_lds(const void* k, int b) { return (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, b); }
void m2t_prof_begin(int, hipStream_t) {}
void m2t_prof_end(int, hipStream_t) {}
bool m2t_prof_take(hipEvent_t*, hipEvent_t*) { return false; }
hipEvent_t m2t_fork_take() { return nullptr; }
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand() { return (rand() / (float)RAND_MAX - 0.5f) * 2.f; }

template <typename F> static float time_us(F&& go, int reps = 20) {
  hipEvent_t e0, e1;
  CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) go();
  CKH(hipDeviceSynchronize());
  CKH(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
  CKH(hipEventRecord(e1, 0));
  CKH(hipEventSynchronize(e1));
  float ms = 0;
  CKH(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

static void run_nt(long long M, int N, int K) {
  std::vector<float> ha((size_t)M * K), hw((size_t)N * K), hy((size_t)M * N);
  for (auto& v : ha) v = frand();
  for (auto& v : hw) v = frand() * 0.1f;
  float *da, *dw, *dy;
  CKH(hipMalloc(&da, ha.size() * 4)); CKH(hipMalloc(&dw, hw.size() * 4)); CKH(hipMalloc(&dy, hy.size() * 4));
  CKH(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  m2t_gemm_args a{};
  a.A = da; a.lda = K; a.W = dw; a.Y = dy; a.ldy = N; a.M = M; a.N = N; a.K = K;
  for (int fast = 0; fast < 2; ++fast) {
    g_m2t_f32_fast = fast;
    CKH(hipMemset(dy, 0xff, hy.size() * 4));
    if (launch_gemm_nt(M2T_F32, M2T_A_PLAIN, M2T_E_PLAIN, a, 0)) exit(1);
    CKH(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4000; ++t) {
      const long long m = (t < 200) ? (t < 100 ? t : M - 1 - (t - 100)) : rand() % M;
      const int n = (t < 200) ? (t * 7) % N : rand() % N;
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)ha[m * K + k] * hw[(size_t)n * K + k]; ref += p; mag += fabs(p); }
      worst = std::max(worst, fabs(hy[m * N + n] - ref) / (mag + 1e-30));
    }
    const float us = time_us([&] { launch_gemm_nt(M2T_F32, M2T_A_PLAIN, M2T_E_PLAIN, a, 0); });
    printf("gemm_nt  M %6lld N %4d K %4d  %s  %7.1f us  %6.1f TFLOP/s  worst |err| / sum|a b| %.2e\n", M, N, K, fast ? "32x32x2" : "16x16x4", us,
           2.0 * M * N * K / us * 1e-6, worst);
  }
  CKH(hipFree(da)); CKH(hipFree(dw)); CKH(hipFree(dy));
}

static void run_tn(long long M, int N, int K) {
  std::vector<float> hg((size_t)M * N), hx((size_t)M * K);
  for (auto& v : hg) v = frand() * 0.1f;
  for (auto& v : hx) v = frand();
  const int bound = wgrad_slab_count(M, N, K);
  float *dg, *dx, *ds;
  CKH(hipMalloc(&dg, hg.size() * 4)); CKH(hipMalloc(&dx, hx.size() * 4)); CKH(hipMalloc(&ds, (size_t)bound * N * K * 4));
  CKH(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  m2t_wgrad_args a{};
  a.G = dg; a.ldg = N; a.gmode = M2T_A_PLAIN; a.X = dx; a.ldx = K; a.xmode = M2T_A_PLAIN; a.slabs = ds; a.M = M; a.N = N; a.K = K;
  std::vector<float> hs((size_t)bound * N * K);
  for (int fast = 0; fast < 2; ++fast) {
    g_m2t_f32_fast = fast;
    int ns = 0;
    CKH(hipMemset(ds, 0xff, hs.size() * 4));
    if (launch_wgrad_tn(M2T_F32, a, &ns, 0)) exit(1);
    if (ns > bound) { printf("slab bound exceeded: %d > %d\n", ns, bound); exit(1); }
    CKH(hipMemcpy(hs.data(), ds, hs.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 600; ++t) {
      const int n = t < 64 ? t % N : rand() % N, k = t < 64 ? (t * 5) % K : rand() % K;
      double ref = 0, mag = 0, got = 0;
      for (long long m = 0; m < M; ++m) { const double p = (double)hg[m * N + n] * hx[m * K + k]; ref += p; mag += fabs(p); }
      for (int s = 0; s < ns; ++s) got += hs[(size_t)s * N * K + (size_t)n * K + k];
      worst = std::max(worst, fabs(got - ref) / (mag + 1e-30));
    }
    const float us = time_us([&] { int q; launch_wgrad_tn(M2T_F32, a, &q, 0); });
    printf("wgrad_tn M %6lld N %4d K %4d  %s  %7.1f us  %6.1f TFLOP/s  slabs %3d  worst |err| / sum|g x| %.2e\n", M, N, K,
           fast ? "32x32x2" : "16x16x4", us, 2.0 * M * N * K / us * 1e-6, ns, worst);
  }
  CKH(hipFree(dg)); CKH(hipFree(dx)); CKH(hipFree(ds));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16;
  srand(3);
  // (rows, C) of the step's branches at 64 x 64 LR patches: C = 64 at half resolution, C = 256 at a quarter
  const long long M64 = (long long)B * 32 * 32, M256 = (long long)B * 16 * 16;
  run_nt(M256, 768, 256);      // C = 256 qkv projection
  run_nt(M256, 256, 768);      // its data gradient
  run_nt(M64, 192, 64);        // C = 64 qkv projection
  run_nt(M64, 64, 192);        // its data gradient
  run_nt(M256 + 64, 768, 256); // ragged rows
  run_tn(M256, 768, 256);
  run_tn(M64, 192, 64);
  run_tn(M256 + 64, 768, 256);
  return 0;
}
