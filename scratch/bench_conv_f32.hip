// Stand-alone harness for the fp32 (parity mode) 3x3 conv kernels of k_conv.hip: the 16x16x4 kernels of rounds 1-4 against the
// 32x32x2 kernels of round 5 (forward / data gradient and weight gradient), P64 fp32 feature maps, no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_conv_f32.hip -o scratch/bench_conv_f32
//   ./bench_conv_f32 [B=16] [H=128] [W=128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../m2trans_amd/csrc/k_conv.hip"   // with scratch/k_conv_f32_32x32x2_r05.diff.txt applied (the kernels tied and were not kept)
thread_local int g_m2t_f32_fast = 1;
int m2t_set_hip_error(hipError_t e, const char* f, int l) { fprintf(stderr, "HIP error %d %s at %s:%d\n", (int)e, hipGetErrorString(e), f, l); return (int)e; }
int m2t_set_error(int c, const char* m) { fprintf(stderr, "error %d %s\n", c, m); return c; }
int m2t_ensure_dynamic_lds(const void* k, int b) { return (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, b); }
void m2t_prof_begin(int, hipStream_t) {}
void m2t_prof_end(int, hipStream_t) {}
bool m2t_prof_take(hipEvent_t*, hipEvent_t*) { return false; }
hipEvent_t m2t_fork_take() { return nullptr; }
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float frand() { return (rand() / (float)RAND_MAX - 0.5f) * 2.f; }
template <typename F> static float time_us(F&& go, int reps = 10) {
  hipEvent_t e0, e1;
  CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) go();
  CKH(hipDeviceSynchronize());
  CKH(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
  CKH(hipEventRecord(e1, 0));
  CKH(hipEventSynchronize(e1));
  float ms = 0;
  CKH(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128;
  const long long npix = (long long)B * H * W;
  const size_t n = (size_t)npix * 64;
  srand(5);
  std::vector<float> hx(n), hr(n), hw(9 * 64 * 64), hb(64);
  for (auto& v : hx) v = frand();
  for (auto& v : hr) v = frand();
  for (auto& v : hw) v = frand() * 0.05f;      // packed [tap][oc][ic]
  for (auto& v : hb) v = frand();
  float *dx, *dr, *dw, *db, *dy, *dsl, *dbs;
  CKH(hipMalloc(&dx, n * 4)); CKH(hipMalloc(&dr, n * 4)); CKH(hipMalloc(&dy, n * 4)); CKH(hipMalloc(&dw, hw.size() * 4)); CKH(hipMalloc(&db, 256));
  CKH(hipMalloc(&dsl, (size_t)256 * 9 * 4096 * 4)); CKH(hipMalloc(&dbs, 256 * 64 * 4));
  CKH(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(dr, hr.data(), n * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice));
  const double flop = 2.0 * npix * 64 * 576;
  std::vector<float> y[2], wsum[2], bsum[2];
  for (int fast = 0; fast < 2; ++fast) {
    g_m2t_f32_fast = fast;
    CKH(hipMemset(dy, 0xff, n * 4));
    if (launch_conv3x3_c64(M2T_F32, dx, dw, db, dr, nullptr, dy, B, H, W, 0, nullptr, nullptr, 0, nullptr)) return 1;
    y[fast].resize(n);
    CKH(hipMemcpy(y[fast].data(), dy, n * 4, hipMemcpyDeviceToHost));
    const float us = time_us([&] { launch_conv3x3_c64(M2T_F32, dx, dw, db, dr, nullptr, dy, B, H, W, 0, nullptr, nullptr, 0, nullptr); });
    printf("conv fwd   B %d %dx%d  %s  %7.1f us  %6.1f TFLOP/s\n", B, H, W, fast ? "32x32x2" : "16x16x4", us, flop / us * 1e-6);
    int ns = 0;
    CKH(hipMemset(dsl, 0xff, (size_t)256 * 9 * 4096 * 4));
    if (launch_conv3x3_c64_wgrad(M2T_F32, dx, dr, dsl, dbs, &ns, B, H, W, 0)) return 1;
    std::vector<float> hs((size_t)ns * 9 * 4096), hbs((size_t)ns * 64);
    CKH(hipMemcpy(hs.data(), dsl, hs.size() * 4, hipMemcpyDeviceToHost));
    CKH(hipMemcpy(hbs.data(), dbs, hbs.size() * 4, hipMemcpyDeviceToHost));
    wsum[fast].assign(9 * 4096, 0.f); bsum[fast].assign(64, 0.f);
    std::vector<double> acc(9 * 4096, 0.0), accb(64, 0.0);
    for (int s = 0; s < ns; ++s) {
      for (int i = 0; i < 9 * 4096; ++i) acc[i] += hs[(size_t)s * 9 * 4096 + i];
      for (int i = 0; i < 64; ++i) accb[i] += hbs[(size_t)s * 64 + i];
    }
    for (int i = 0; i < 9 * 4096; ++i) wsum[fast][i] = (float)acc[i];
    for (int i = 0; i < 64; ++i) bsum[fast][i] = (float)accb[i];
    const float us2 = time_us([&] { int q; launch_conv3x3_c64_wgrad(M2T_F32, dx, dr, dsl, dbs, &q, B, H, W, 0); });
    printf("conv wgrad B %d %dx%d  %s  %7.1f us  %6.1f TFLOP/s  slabs %d\n", B, H, W, fast ? "32x32x2" : "16x16x4", us2, flop / us2 * 1e-6, ns);
  }
  double dmax = 0, ymax = 0;
  for (size_t i = 0; i < n; ++i) { dmax = std::max(dmax, (double)fabsf(y[0][i] - y[1][i])); ymax = std::max(ymax, (double)fabsf(y[0][i])); }
  printf("forward: max |new - old| %.3e (max |y| %.3f)\n", dmax, ymax);
  // a few outputs against an fp64 host evaluation
  double worst = 0;
  for (int t = 0; t < 2000; ++t) {
    const int b = rand() % B, yy = (t < 200) ? (t & 1 ? 0 : H - 1) : rand() % H, xx = (t < 400) ? (t & 2 ? 0 : W - 1) : rand() % W, oc = rand() % 64;
    double ref = hb[oc], mag = fabs(hb[oc]);
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
      if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
      for (int ic = 0; ic < 64; ++ic) {
        const double p = (double)hw[(tap * 64 + oc) * 64 + ic] * hx[p64(npix, ((long long)b * H + sy) * W + sx, ic)];
        ref += p; mag += fabs(p);
      }
    }
    const long long o = p64(npix, ((long long)b * H + yy) * W + xx, oc);
    ref += hr[o];
    worst = std::max(worst, fabs(y[1][o] - ref) / (mag + 1.0));
  }
  printf("forward 32x32x2 vs fp64: worst |err| / (sum |w x| + 1) %.2e\n", worst);
  double wd = 0, wm = 0, bd = 0, bm = 0;
  for (int i = 0; i < 9 * 4096; ++i) { wd = std::max(wd, (double)fabsf(wsum[0][i] - wsum[1][i])); wm = std::max(wm, (double)fabsf(wsum[0][i])); }
  for (int i = 0; i < 64; ++i) { bd = std::max(bd, (double)fabsf(bsum[0][i] - bsum[1][i])); bm = std::max(bm, (double)fabsf(bsum[0][i])); }
  printf("weight gradient: max |new - old| %.3e (max %.3f);  bias gradient: %.3e (max %.3f)\n", wd, wm, bd, bm);
  return 0;
}
