// Round 6: the x4 tail backward tile kernel on 8 waves (rounds 2-5) against its 16-wave partition (k_tail_bwd.hip), stand-alone:
// g(t1) must agree bit for bit, the three parameter-gradient slab sums to fp32 summation order; median time of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_tail_bwd16.hip -o scratch/bench_tail_bwd16 [-DSTAMPS -DSTAMP_WAVE=n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#ifdef STAMPS
#ifndef STAMP_WAVE
#define STAMP_WAVE 0
#endif
__device__ unsigned long long* g_stamps = nullptr;
#define M2T_TAIL_STAMP(i) do { const long long it__ = (t - t0) / tstep; if (g_stamps && it__ >= 2 && it__ < 6 && threadIdx.x == 64 * STAMP_WAVE) g_stamps[((size_t)blockIdx.x * 4 + (it__ - 2)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define M2T_TAIL_STAMP2(i) do { if (tile_it >= 2 && tile_it < 6 && threadIdx.x == 64 * STAMP_WAVE && g_stamps) g_stamps[((size_t)blockIdx.x * 4 + (tile_it - 2)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "../m2trans_amd/csrc/k_tail_bwd.hip"
int m2t_set_hip_error(hipError_t e, const char* f, int l) { fprintf(stderr, "HIP error %d %s at %s:%d\n", (int)e, hipGetErrorString(e), f, l); return (int)e; }
int m2t_set_error(int c, const char* m) { fprintf(stderr, "error %d %s\n", c, m); return c; }
int m2t_ensure_dynamic_lds(const void* k, int b) { return (int)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, b); }
void m2t_prof_begin(int, hipStream_t) {}
void m2t_prof_end(int, hipStream_t) {}
bool m2t_prof_take(hipEvent_t*, hipEvent_t*) { return false; }
hipEvent_t m2t_fork_take() { return nullptr; }
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned short f2bf(float f) { union { float f; unsigned u; } c; c.f = f; unsigned u = c.u; return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float frand() { return rand() / (float)RAND_MAX - 0.5f; }
template <typename F> static float time_it(hipStream_t st, int n, F f) {
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  std::vector<float> ts;
  for (int i = 0; i < n; ++i) {
    CKH(hipEventRecord(e0, st)); f(); CKH(hipEventRecord(e1, st)); CKH(hipEventSynchronize(e1));
    float ms; CKH(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1000.f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

static int run_case(int B, int Hlr, int Wlr, bool timing, bool l1) {
  const int H = 4 * Hlr, W = 4 * Wlr, Hm = H / 2, Wm = W / 2;
  const int Hs = l1 ? H - 5 : H, Ws = l1 ? W - 3 : W;                 // the L1 arm crops (reflect-padded sizes)
  const size_t nmid = (size_t)B * Hm * Wm * 64, nhr = (size_t)B * 3 * H * W, ncrop = (size_t)B * 3 * Hs * Ws;
  srand(7 + B + Hlr);
  std::vector<unsigned short> ha1(nmid), hd1(nmid), hw3t(64 * 256);
  for (auto& v : ha1) { float t = 3.f * frand(); v = f2bf(0.5f * t * (1.f + erff(t * 0.70710678f))); }
  for (auto& v : hd1) v = f2bf(0.5f + frand());
  for (int n = 0; n < 256; ++n) for (int k = 0; k < 64; ++k) hw3t[k * 256 + n] = f2bf(0.25f * frand());
  std::vector<float> hb3(256), hwf(3 * 64 * 9), hg(nhr), hhr(ncrop);
  for (auto& v : hb3) v = 0.2f * frand();
  for (auto& v : hwf) v = 0.1f * frand();
  for (auto& v : hg) v = l1 ? (0.5f + 1.2f * frand()) : ((rand() % 7 == 0) ? 0.f : 1e-3f * frand());     // L1 arm: the pre-clamp output
  for (auto& v : hhr) v = 0.5f + frand();
  void *da1, *dd1, *dw3t, *dgt1; float *db3, *dwf, *dg, *dhr, *swf, *sw3, *sb3, *lp;
  CKH(hipMalloc(&da1, nmid * 2)); CKH(hipMalloc(&dd1, nmid * 2)); CKH(hipMalloc(&dgt1, nmid * 2)); CKH(hipMalloc(&dw3t, 256 * 64 * 2));
  CKH(hipMalloc(&db3, 1024)); CKH(hipMalloc(&dwf, hwf.size() * 4)); CKH(hipMalloc(&dg, nhr * 4)); CKH(hipMalloc(&dhr, ncrop * 4)); CKH(hipMalloc(&lp, 4096));
  const int nbmax = 512;
  CKH(hipMalloc(&swf, (size_t)nbmax * 32 * 64 * 4)); CKH(hipMalloc(&sw3, (size_t)nbmax * 256 * 64 * 4)); CKH(hipMalloc(&sb3, (size_t)nbmax * 256 * 4));
  CKH(hipMemcpy(da1, ha1.data(), nmid * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dd1, hd1.data(), nmid * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw3t, hw3t.data(), 256 * 64 * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(db3, hb3.data(), 1024, hipMemcpyHostToDevice)); CKH(hipMemcpy(dwf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dg, hg.data(), nhr * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(dhr, hhr.data(), ncrop * 4, hipMemcpyHostToDevice));
  hipStream_t st; CKH(hipStreamCreate(&st));
  int ns = 0;
  auto bwd = [&](int waves) {
    if (launch_tail_bwd_fused(dg, dwf, nullptr, nullptr, da1, dd1, dw3t, db3, dgt1, swf, sw3, sb3, &ns, B, H, W, st, l1 ? dg : nullptr, l1 ? dhr : nullptr,
                              l1 ? lp : nullptr, Hs, Ws, 1.0f, 1e-3f, waves)) exit(2);
  };
  struct Res { std::vector<unsigned short> gt; std::vector<double> s1, s2, s3; double loss; };
  auto collect = [&](int waves) {
    CKH(hipMemset(dgt1, 0xff, nmid * 2));
    bwd(waves);
    CKH(hipStreamSynchronize(st));
    Res r; r.gt.resize(nmid);
    CKH(hipMemcpy(r.gt.data(), dgt1, nmid * 2, hipMemcpyDeviceToHost));
    std::vector<float> h1((size_t)ns * 32 * 64), h2((size_t)ns * 256 * 64), h3((size_t)ns * 256), hl(ns);
    CKH(hipMemcpy(h1.data(), swf, h1.size() * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(h2.data(), sw3, h2.size() * 4, hipMemcpyDeviceToHost));
    CKH(hipMemcpy(h3.data(), sb3, h3.size() * 4, hipMemcpyDeviceToHost));
    auto fold = [&](const std::vector<float>& s, size_t n) { std::vector<double> o(n, 0.0); for (int k = 0; k < ns; ++k) for (size_t i = 0; i < n; ++i) o[i] += s[(size_t)k * n + i]; return o; };
    r.s1 = fold(h1, 32 * 64); r.s2 = fold(h2, 256 * 64); r.s3 = fold(h3, 256);
    r.loss = 0;
    if (l1) { CKH(hipMemcpy(hl.data(), lp, ns * 4, hipMemcpyDeviceToHost)); for (float v : hl) r.loss += v; }
    return r;
  };
  const Res r8 = collect(16), r16 = collect(32);
  size_t bad = 0;
  double gmax = 0, gdiff = 0, gsq = 0, dsq = 0;
  auto bf2f = [](unsigned short v) { union { unsigned u; float f; } c; c.u = (unsigned)v << 16; return (double)c.f; };
  for (size_t i = 0; i < nmid; ++i) {
    bad += r8.gt[i] != r16.gt[i];
    const double x = bf2f(r8.gt[i]), y = bf2f(r16.gt[i]);
    gmax = std::max(gmax, fabs(x)); gdiff = std::max(gdiff, fabs(x - y)); gsq += x * x; dsq += (x - y) * (x - y);
  }
  auto cmp = [&](const std::vector<double>& x, const std::vector<double>& y) { double d = 0, m = 0; for (size_t i = 0; i < x.size(); ++i) { d = std::max(d, fabs(x[i] - y[i])); m = std::max(m, fabs(x[i])); } return d / (m + 1e-300); };
  printf("B=%d LR %dx%d%s: g(t1) %zu of %zu differ (max %.2e of %.2e, rms rel %.2e); slabs %d; rel diff dWf %.2e dW3 %.2e db3 %.2e", B, Hlr, Wlr, l1 ? " (L1 inside)" : "", bad, nmid,
         gdiff, gmax, sqrt(dsq / (gsq + 1e-300)), ns, cmp(r8.s1, r16.s1), cmp(r8.s2, r16.s2), cmp(r8.s3, r16.s3));
  if (l1) printf("; loss %.9e vs %.9e", r8.loss, r16.loss);
  printf("\n");
  if (timing) {
    const float t8 = time_it(st, 30, [&]() { bwd(16); }), t16 = time_it(st, 30, [&]() { bwd(32); });
    printf("   tail_bwd_fused: 16x16x32 kernel %.1f us   32x32x16 kernel %.1f us\n", t8, t16);
#ifdef STAMPS
    for (int waves : {16, 32}) {
      unsigned long long* ds; const size_t nst = (size_t)4096 * 4 * 16;
      CKH(hipMalloc(&ds, nst * 8)); CKH(hipMemset(ds, 0, nst * 8));
      CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
      bwd(waves); CKH(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs(nst);
      CKH(hipMemcpy(hs.data(), ds, nst * 8, hipMemcpyDeviceToHost));
      double acc[17] = {0}; int cnt = 0;
      for (int blk = 0; blk < ns; ++blk) for (int it = 0; it < 3; ++it) {
        const unsigned long long* a = &hs[((size_t)blk * 4 + it) * 16], *nx = &hs[((size_t)blk * 4 + it + 1) * 16];
        if (!a[0] || !nx[0]) continue;
        for (int i = 1; i < 16; ++i) acc[i] += a[i] ? (double)(a[i] - a[0]) : 0.0;      // offset of stamp i from the top of the tile
        acc[16] += (double)(nx[0] - a[0]); ++cnt;
      }
      printf("   variant %2d, wave %d, stamp offsets from the top of the tile (mean cycles over %d tiles):", waves, STAMP_WAVE, cnt);
      for (int i = 1; i < 17; ++i) printf(" %d:%.0f", i, acc[i] / std::max(cnt, 1));
      printf("\n");
      ds = nullptr; CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
    }
    printf("   [variant 16: 0 stage+bar1 | 1 fetchA+recompute+gelu | 2 Geff | 3 bar2 | 4 g(t2) | 5 dWf | 6 bar3 | 7 g(t1) | 8 fetchB+dW3 | 9 bar4 | 10 loop | 11 total]\n   [variant 32 stamps: 1 after bar1 | 2 fetchA issued | 3 Geff done | 4 after bar2 | 11 R: first mid tile done | 5 R done | 6 after bar3 | 7 F done | 12 D: MFMA chain issued | 13 D: stores issued | 8 D / W done | 9 after bar4 | 16 next tile]\n");
#endif
  }
  for (void* q : {da1, dd1, dgt1, dw3t, (void*)db3, (void*)dwf, (void*)dg, (void*)dhr, (void*)swf, (void*)sw3, (void*)sb3, (void*)lp}) (void)hipFree(q);
  return 0;
}

int main() {
  run_case(2, 40, 56, false, false);       // border tiles on every side, 140 tiles over 140 workgroups
  run_case(3, 72, 200, false, true);       // L1 seed inside, cropped
  run_case(1, 8, 8, false, true);
  run_case(16, 128, 128, true, false);
  run_case(16, 128, 128, true, true);
  run_case(32, 128, 128, true, true);
  return 0;
}
