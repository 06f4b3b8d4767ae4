// Stand-alone check + timing of the fused conv 64 -> 64 backward (conv3x3_c64_bwd_rows_kernel) against the separate
// data-gradient (row-streaming forward kernel on the flipped weight) and weight-gradient kernels.  No torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_conv_bwd.hip -o scratch/bench_conv_bwd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#ifdef STAMPS
__device__ unsigned long long* g_stamps;
#define C3B_NST 64
#define C3B_STEP_WAIT(n) c3r_wait_vm(0)
#define C3B_STAMP(i) do { if ((i) < C3B_NST && (wv & 3) == 0) { if (lane == 0) g_stamps[((size_t)blockIdx.x * 2 + (wv >> 2)) * C3B_NST + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#endif
#include "../m2trans_amd/csrc/k_conv.hip"
int m2t_set_hip_error(hipError_t e, const char* f, int l) { fprintf(stderr, "HIP error %d %s at %s:%d\n", (int)e, hipGetErrorString(e), f, l); return (int)e; }
int m2t_set_error(int c, const char* m) { fprintf(stderr, "error %d %s\n", c, m); return c; }
int m2t_ensure_dynamic_lds(const void* k, int b) { return (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, b); }
void m2t_prof_begin(int, hipStream_t) {}
void m2t_prof_end(int, hipStream_t) {}
bool m2t_prof_take(hipEvent_t*, hipEvent_t*) { return false; }
hipEvent_t m2t_fork_take() { return nullptr; }
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned short f2bf(float f) { union { float f; unsigned u; } c; c.f = f; unsigned u = c.u; return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

static int run_case(int B, int H, int W, bool time_it) {
  const size_t n = (size_t)B * H * W * 64;
  std::vector<unsigned short> hx(n), hg(n), hw(9 * 64 * 64);
  srand(1 + B + H);
  for (auto& v : hx) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hg) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
  void *dx, *dg, *dw, *dwr, *d0, *d1, *dz; float *s0, *s1, *b0, *b1;
  CKH(hipMalloc(&dx, n * 2)); CKH(hipMalloc(&dg, n * 2)); CKH(hipMalloc(&d0, n * 2)); CKH(hipMalloc(&d1, n * 2));
  CKH(hipMalloc(&dw, hw.size() * 2)); CKH(hipMalloc(&dwr, hw.size() * 2)); CKH(hipMalloc(&dz, 65536)); CKH(hipMemset(dz, 0, 65536));
  const size_t slab = 9 * 64 * 64;
  CKH(hipMalloc(&s0, 1024 * slab * 4)); CKH(hipMalloc(&s1, 256 * slab * 4)); CKH(hipMalloc(&b0, 1024 * 64 * 4)); CKH(hipMalloc(&b1, 256 * 64 * 4));
  std::vector<unsigned short> hwr(hw.size());
  for (size_t e = 0; e < hw.size(); ++e) {
    const int j = e & 7, l = (e >> 3) & 63, f = (int)(e >> 9);
    const int nt = f & 1, kc = (f >> 1) & 1, hh = (f >> 2) & 1, tap = f >> 3;
    const int row = 32 * hh + 8 * ((l & 15) >> 2) + 4 * nt + (l & 3), k = 32 * kc + 8 * (l >> 4) + j;
    hwr[e] = hw[((size_t)tap * 64 + row) * 64 + k];
  }
  CKH(hipMemcpy(dwr, hwr.data(), hwr.size() * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dg, hg.data(), n * 2, hipMemcpyHostToDevice));
  CKH(hipMemset(d0, 0xff, n * 2)); CKH(hipMemset(d1, 0xee, n * 2));
  hipStream_t st; CKH(hipStreamCreate(&st));
  int ns0 = 0, ns1 = 0;
  auto separate = [&]() {
    int rc = launch_conv3x3_c64(M2T_BF16, dg, dw, nullptr, nullptr, nullptr, d0, B, H, W, st, dwr, dz, 0);
    return rc ? rc : launch_conv3x3_c64_wgrad(M2T_BF16, dx, dg, s0, b0, &ns0, B, H, W, st);
  };
  auto fused = [&]() { return launch_conv3x3_c64_bwd_fused(dg, dx, dwr, d1, s1, b1, &ns1, dz, B, H, W, st); };
  if (separate()) return 1;
  if (fused()) return 1;
  CKH(hipStreamSynchronize(st));
  std::vector<unsigned short> y0(n), y1(n);
  CKH(hipMemcpy(y0.data(), d0, n * 2, hipMemcpyDeviceToHost)); CKH(hipMemcpy(y1.data(), d1, n * 2, hipMemcpyDeviceToHost));
  size_t bad = 0, first = (size_t)-1;
  for (size_t i = 0; i < n; ++i) if (y0[i] != y1[i]) { if (!bad) first = i; ++bad; }
  std::vector<float> h0((size_t)ns0 * slab), h1((size_t)ns1 * slab), hb0((size_t)ns0 * 64), hb1((size_t)ns1 * 64);
  CKH(hipMemcpy(h0.data(), s0, h0.size() * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(h1.data(), s1, h1.size() * 4, hipMemcpyDeviceToHost));
  CKH(hipMemcpy(hb0.data(), b0, hb0.size() * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(hb1.data(), b1, hb1.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0, bworst = 0, bscale = 0; size_t wi = 0;
  for (size_t e = 0; e < slab; ++e) {
    double a = 0, c = 0;
    for (int s = 0; s < ns0; ++s) a += h0[(size_t)s * slab + e];
    const size_t et = (e / 4096) * 4096 + (e % 64) * 64 + (e / 64) % 64;        // the fused kernel's slabs are [tap][ic][oc]
    for (int s = 0; s < ns1; ++s) c += h1[(size_t)s * slab + et];
    if (std::fabs(a - c) > worst) { worst = std::fabs(a - c); wi = e; }
    scale = std::max(scale, std::fabs(a));
  }
  for (int e = 0; e < 64; ++e) {
    double a = 0, c = 0;
    for (int s = 0; s < ns0; ++s) a += hb0[(size_t)s * 64 + e];
    for (int s = 0; s < ns1; ++s) c += hb1[(size_t)s * 64 + e];
    bworst = std::max(bworst, std::fabs(a - c)); bscale = std::max(bscale, std::fabs(a));
  }
  const bool wok = worst <= 2e-5 * scale + 1e-3, bok = bworst <= 2e-5 * bscale + 1e-3;
  printf("B=%d %dx%d: data gradient %zu of %zu elements differ; weight gradient (%d vs %d slabs) max |diff| %.3g of max %.3g (tap %zu oc %zu ic %zu) %s; bias %.3g of %.3g %s\n",
         B, H, W, bad, n, ns0, ns1, worst, scale, wi / 4096, (wi / 64) % 64, wi % 64, wok ? "ok" : "BAD", bworst, bscale, bok ? "ok" : "BAD");
  if (bad) {
    const size_t npix = (size_t)B * H * W, pl = first / (npix * 16), pix = (first / 16) % npix, ch = first % 16;
    printf(" (first: plane %zu image %zu row %zu col %zu ch %zu: %04x vs %04x)\n", pl, pix / ((size_t)H * W), (pix / W) % H, pix % W, ch, y0[first], y1[first]);
  }
  if (time_it) {
    hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
      const int N = 40; std::vector<float> ts;
      for (int i = 0; i < 5; ++i) which ? fused() : separate();
      for (int i = 0; i < N; ++i) {
        CKH(hipEventRecord(e0, st)); which ? fused() : separate(); CKH(hipEventRecord(e1, st)); CKH(hipEventSynchronize(e1));
        float ms; CKH(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      printf("   %s: min %.2f us median %.2f us -> %.0f TFLOP/s, %.2f TB/s on 3 tensor passes\n", which ? "fused backward        " : "separate dgrad + wgrad", ts[0], ts[N / 2],
             4.0 * B * H * W * 64 * 576 / ts[N / 2] * 1e-6, (double)n * 2 * 3 / ts[N / 2] * 1e-6);
    }
  }
  CKH(hipFree(dx)); CKH(hipFree(dg)); CKH(hipFree(d0)); CKH(hipFree(d1)); CKH(hipFree(dw)); CKH(hipFree(dwr)); CKH(hipFree(dz));
  CKH(hipFree(s0)); CKH(hipFree(s1)); CKH(hipFree(b0)); CKH(hipFree(b1));
  return (bad || !wok || !bok) ? 2 : 0;
}
#ifdef STAMPS
static void stamp_report(int B, int H, int W) {
  const size_t n = (size_t)B * H * W * 64;
  void *dx, *dg, *dw, *dy, *dz; float *s1, *b1; unsigned long long* dst;
  CKH(hipMalloc(&dx, n * 2)); CKH(hipMalloc(&dg, n * 2)); CKH(hipMalloc(&dy, n * 2)); CKH(hipMalloc(&dw, 9 * 64 * 64 * 2)); CKH(hipMalloc(&dz, 65536));
  CKH(hipMalloc(&s1, 256 * 9 * 64 * 64 * 4)); CKH(hipMalloc(&b1, 256 * 64 * 4));
  CKH(hipMemset(dz, 0, 256)); CKH(hipMemset(dx, 0x3c, n * 2)); CKH(hipMemset(dg, 0x3c, n * 2)); CKH(hipMemset(dw, 0x3c, 9 * 64 * 64 * 2));
  const int NB = 512;
  CKH(hipMalloc(&dst, (size_t)NB * C3B_NST * 8)); CKH(hipMemset(dst, 0, (size_t)NB * C3B_NST * 8));
  CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
  hipStream_t st; CKH(hipStreamCreate(&st));
  int ns = 0;
  launch_conv3x3_c64_bwd_fused(dg, dx, dw, dy, s1, b1, &ns, dz, B, H, W, st);
  CKH(hipStreamSynchronize(st));
  std::vector<unsigned long long> hs((size_t)NB * C3B_NST);
  CKH(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
  for (int role = 0; role < 2; ++role) {
    unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
    for (int b = role; b < NB; b += 2) if (hs[(size_t)b * C3B_NST]) { ++nb; t0 = std::min(t0, hs[(size_t)b * C3B_NST]); for (int i = 0; i < C3B_NST; ++i) t1 = std::max(t1, hs[(size_t)b * C3B_NST + i]); }
    printf("STAMPS B=%d %dx%d %s wave: %d workgroups, first start -> last stamp %llu ticks (s_memtime, 100 MHz)\n", B, H, W, role ? "weight-gradient" : "data-gradient", nb, t1 - t0);
    auto med = [&](int i, int j) { std::vector<long long> d; for (int b = role; b < NB; b += 2) { auto a = hs[(size_t)b * C3B_NST + i], c = hs[(size_t)b * C3B_NST + j]; if (a && c) d.push_back((long long)(c - a)); }
      if (d.empty()) return std::make_pair(-1LL, -1LL); std::sort(d.begin(), d.end()); return std::make_pair(d[d.size() / 2], d.back()); };
    auto p = med(0, 1); printf("  start -> weights + prologue landed : median %lld max %lld\n", p.first, p.second);
    int last = 1;
    for (int s = 0; s < 20; ++s) {
      auto a = med(last, 2 + 3 * s), c = med(2 + 3 * s, 3 + 3 * s), e = med(3 + 3 * s, 4 + 3 * s);
      if (a.first < 0) break;
      printf("  step %2d: wait+barrier %6lld (max %6lld)  dma issue %6lld  products (+ stores) %6lld\n", s, a.first, a.second, c.first, e.first);
      last = 4 + 3 * s;
    }
    auto z = med(last, 63); printf("  last step -> end (drain, slab stores): median %lld max %lld\n", z.first, z.second);
  }
}
#endif
int main() {
  int rc = 0;
#ifdef STAMPS
  stamp_report(16, 128, 128); stamp_report(32, 128, 128);
  return 0;
#endif
  rc |= run_case(1, 32, 32, false);
  rc |= run_case(2, 64, 96, false);
  rc |= run_case(3, 160, 64, false);
  rc |= run_case(16, 128, 128, true);
  rc |= run_case(32, 128, 128, true);
  rc |= run_case(8, 256, 256, true);
  printf(rc ? "FAILED\n" : "ALL OK\n");
  return rc;
}
