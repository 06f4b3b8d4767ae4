// Stand-alone check + timing of the row-streaming 3x3 conv (conv3x3_c64_rows_kernel) against the tile kernel
// (conv3x3_c64_pipe_kernel): bit-identical outputs required.  No torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_conv_rows.hip -o scratch/bench_conv_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef STAMPS
__device__ unsigned long long* g_stamps;
#define C3R_NST 64
#define C3R_STAMP(i) do { if ((i) < C3R_NST && threadIdx.x == 0) { g_stamps[(size_t)blockIdx.x * C3R_NST + (i)] = __builtin_amdgcn_s_memtime(); } if (wv == 0 && (i) < C3R_NST) ++issued; } while (0)
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

static int run_case(int B, int H, int W, int nres, bool time_it) {
  const size_t n = (size_t)B * H * W * 64;
  std::vector<unsigned short> hx(n), hr1(n), hr2(n), hw(9 * 64 * 64);
  srand(1 + B + H);
  for (auto& v : hx) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hr1) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hr2) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
  std::vector<float> hb(64);
  for (int i = 0; i < 64; ++i) hb[i] = 0.01f * (i - 30);
  void *dx, *dw, *dr1, *dr2, *dy0, *dy1, *dz; float* db;
  CKH(hipMalloc(&dx, n * 2)); CKH(hipMalloc(&dr1, n * 2)); CKH(hipMalloc(&dr2, n * 2)); CKH(hipMalloc(&dy0, n * 2)); CKH(hipMalloc(&dy1, n * 2));
  CKH(hipMalloc(&dw, hw.size() * 2)); CKH(hipMalloc(&db, 256)); CKH(hipMalloc(&dz, 256)); CKH(hipMemset(dz, 0, 256));
  // the same weight ([tap][oc][ic], as the tile kernel reads it) in M2T_PACK_CONV3_ROWS order
  std::vector<unsigned short> hwr(hw.size());
  for (size_t e = 0; e < hw.size(); ++e) {
    const int j = e & 7, l = (e >> 3) & 63, f = (int)(e >> 9);
    const int nt = f & 1, kc = (f >> 1) & 1, hh = (f >> 2) & 1, tap = f >> 3;
    const int row = 32 * hh + 8 * ((l & 15) >> 2) + 4 * nt + (l & 3), k = 32 * kc + 8 * (l >> 4) + j;
    hwr[e] = hw[((size_t)tap * 64 + row) * 64 + k];
  }
  void* dwr; CKH(hipMalloc(&dwr, hwr.size() * 2)); CKH(hipMemcpy(dwr, hwr.data(), hwr.size() * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dr1, hr1.data(), n * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dr2, hr2.data(), n * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice));
  CKH(hipMemset(dy0, 0xff, n * 2)); CKH(hipMemset(dy1, 0xee, n * 2));
  hipStream_t st; CKH(hipStreamCreate(&st));
  const float* bias = nres > 0 ? db : nullptr;
  const void* r1 = nres > 0 ? dr1 : nullptr;
  const void* r2 = nres > 1 ? dr2 : nullptr;
  auto run = [&](void* y, int variant) { return launch_conv3x3_c64(M2T_BF16, dx, dw, bias, r1, r2, y, B, H, W, st, dwr, dz, variant); };
  if (run(dy0, 1)) return 1;
  CKH(hipStreamSynchronize(st));
  std::vector<unsigned short> y0(n), y1(n);
  CKH(hipMemcpy(y0.data(), dy0, n * 2, hipMemcpyDeviceToHost));
  size_t bad = 0, first = (size_t)-1;
  for (int variant : {0, 3, 4}) {
    CKH(hipMemset(dy1, 0xee, n * 2));
    if (run(dy1, variant)) return 1;
    CKH(hipStreamSynchronize(st));
    CKH(hipMemcpy(y1.data(), dy1, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) if (y0[i] != y1[i]) { if (!bad) first = i; ++bad; }
  }
  printf("B=%d %dx%d nres=%d (variants 0, 3, 4 against the tile kernel): %zu of %zu elements differ", B, H, W, nres, bad, 3 * n);
  if (bad) {
    const size_t npix = (size_t)B * H * W, pl = first / (npix * 16), pix = (first / 16) % npix, ch = first % 16;
    printf(" (first: plane %zu image %zu row %zu col %zu ch %zu: %04x vs %04x)", pl, pix / ((size_t)H * W), (pix / W) % H, pix % W, ch, y0[first], y1[first]);
  }
  printf("\n");
  if (time_it) {
    hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
    for (int variant : {1, 0, 3, 4}) {
      const int N = 40; std::vector<float> ts;
      for (int i = 0; i < 5; ++i) run(dy1, variant);
      for (int i = 0; i < N; ++i) {
        CKH(hipEventRecord(e0, st)); run(dy1, variant); CKH(hipEventRecord(e1, st)); CKH(hipEventSynchronize(e1));
        float ms; CKH(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      const double bytes = (double)n * 2 * (2 + nres);
      printf("   %s: min %.2f us median %.2f us -> %.2f TB/s algorithmic, %.0f TFLOP/s\n", variant == 1 ? "tile kernel (pipe)    " : (variant == 0 ? "row-streaming D=2     " : (variant == 3 ? "row-streaming D=3     " : "row-streaming D=2 pipe")), ts[0], ts[N / 2],
             bytes / ts[N / 2] * 1e-6, 2.0 * B * H * W * 64 * 576 / ts[N / 2] * 1e-6);
    }
  }
  CKH(hipFree(dx)); CKH(hipFree(dr1)); CKH(hipFree(dr2)); CKH(hipFree(dy0)); CKH(hipFree(dy1)); CKH(hipFree(dw)); CKH(hipFree(db)); CKH(hipFree(dz));
  return bad ? 2 : 0;
}
#ifdef STAMPS
static void stamp_report(int B, int H, int W, int nres) {
  const size_t n = (size_t)B * H * W * 64;
  void *dx, *dw, *dr1, *dy, *dz; float* db; unsigned long long* dst;
  CKH(hipMalloc(&dx, n * 2)); CKH(hipMalloc(&dr1, n * 2)); CKH(hipMalloc(&dy, n * 2)); CKH(hipMalloc(&dw, 9 * 64 * 64 * 2)); CKH(hipMalloc(&db, 256)); CKH(hipMalloc(&dz, 256));
  CKH(hipMemset(dz, 0, 256)); CKH(hipMemset(dx, 0x3c, n * 2)); CKH(hipMemset(dr1, 0x3c, n * 2)); CKH(hipMemset(dw, 0x3c, 9 * 64 * 64 * 2)); CKH(hipMemset(db, 0, 256));
  const int NB = 4096;
  // (weights are a constant pattern here: the fragment order does not matter for timing)
  CKH(hipMalloc(&dst, (size_t)NB * C3R_NST * 8)); CKH(hipMemset(dst, 0, (size_t)NB * C3R_NST * 8));
  CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
  hipStream_t st; CKH(hipStreamCreate(&st));
  for (int i = 0; i < 1; ++i) launch_conv3x3_c64(M2T_BF16, dx, dw, nres ? db : nullptr, nres ? dr1 : nullptr, nullptr, dy, B, H, W, st, dw, dz, 0);
  CKH(hipStreamSynchronize(st));
  std::vector<unsigned long long> hs((size_t)NB * C3R_NST);
  CKH(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
  for (int b = 0; b < NB; ++b) if (hs[(size_t)b * C3R_NST]) { ++nb; t0 = std::min(t0, hs[(size_t)b * C3R_NST]); for (int i = 0; i < C3R_NST; ++i) t1 = std::max(t1, hs[(size_t)b * C3R_NST + i]); }
  printf("STAMPS B=%d %dx%d nres=%d: %d workgroups, first start -> last stamp %llu ticks (s_memtime, 100 MHz?)\n", B, H, W, nres, nb, t1 - t0);
  auto med = [&](int i, int j) { std::vector<long long> d; for (int b = 0; b < NB; ++b) { auto a = hs[(size_t)b * C3R_NST + i], c = hs[(size_t)b * C3R_NST + j]; if (a && c) d.push_back((long long)(c - a)); }
    if (d.empty()) return std::make_pair(-1LL, -1LL); std::sort(d.begin(), d.end()); return std::make_pair(d[d.size() / 2], d.back()); };
  auto p = med(0, 1); printf("  start -> weights + prologue landed : median %lld max %lld\n", p.first, p.second);
  for (int s = 0; s < 20; ++s) {
    auto a = med(s == 0 ? 1 : 4 + 3 * (s - 1), 2 + 3 * s), c = med(2 + 3 * s, 3 + 3 * s), e = med(3 + 3 * s, 4 + 3 * s);
    if (a.first < 0) break;
    printf("  step %2d: wait+barrier %6lld (max %6lld)  dma issue + products %6lld  epilogue %6lld\n", s, a.first, a.second, c.first, e.first);
  }
  std::vector<long long> starts; for (int b = 0; b < NB; ++b) if (hs[(size_t)b * C3R_NST]) starts.push_back((long long)(hs[(size_t)b * C3R_NST] - t0));
  std::sort(starts.begin(), starts.end());
  printf("  workgroup start times: 50%% %lld, 90%% %lld, last %lld\n", starts[starts.size() / 2], starts[starts.size() * 9 / 10], starts.back());
}
#endif
int main(int argc, char** argv) {
  int rc = 0;
#ifdef STAMPS
  stamp_report(16, 128, 128, 1); stamp_report(16, 128, 128, 0); stamp_report(32, 128, 128, 1);
  return 0;
#endif
  if (argc > 1) {                      // sweep: segment lengths at the benchmark sizes
    for (int rs : {16, 32, 64}) {
      c3r_force_rs = rs;
      printf("== rows per segment %d ==\n", rs);
      rc |= run_case(16, 128, 128, 1, true);
      rc |= run_case(32, 128, 128, 1, true);
    }
    return rc;
  }
  rc |= run_case(2, 64, 96, 1, false);
  rc |= run_case(1, 32, 32, 0, false);
  rc |= run_case(3, 160, 64, 2, false);
  for (int nres = 0; nres <= 2; ++nres) rc |= run_case(16, 128, 128, nres, true);
  rc |= run_case(32, 128, 128, 1, true);
  rc |= run_case(8, 256, 256, 1, true);
  printf(rc ? "FAILED\n" : "ALL IDENTICAL\n");
  return rc;
}
