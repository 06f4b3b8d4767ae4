// Stand-alone timing + output hashes of the x4 tail's two fused high-resolution kernels (k_tail_fwd.hip, k_tail_bwd.hip).
// No torch.  Two builds of this file (e.g. the committed kernels and a work-in-progress copy) print hashes that must
// agree when the change is meant to be bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scratch/bench_tail.hip -o scratch/bench_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#ifdef STAMPS
// per-phase s_memtime of wave STAMP_WAVE of every workgroup, tiles 2 .. 5 of its strip (11 stamps per tile)
#ifndef STAMP_WAVE
#define STAMP_WAVE 0
#endif
__device__ unsigned long long* g_stamps = nullptr;
__device__ int g_stamp_tile;
#define M2T_TAIL_STAMP(i) do { const long long it__ = (t - t0) / tstep; if (g_stamps && it__ >= 2 && it__ < 6 && threadIdx.x == 64 * STAMP_WAVE) g_stamps[((size_t)blockIdx.x * 4 + (it__ - 2)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifdef BSTAMPS
__device__ unsigned long long* g_bstamps = nullptr;
#ifndef BSTAMP_WAVE
#define BSTAMP_WAVE 0
#endif
// steps 5 .. 8 of the first task of every workgroup, wave BSTAMP_WAVE
#define BS_STAMP(i) do { if (g_bstamps && task == (int)xcd_block_index() && s >= 5 && s < 9 && threadIdx.x == 64 * BSTAMP_WAVE) g_bstamps[((size_t)blockIdx.x * 4 + (s - 5)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifndef TAIL_FWD_SRC
#define TAIL_FWD_SRC "../m2trans_amd/csrc/k_tail_fwd.hip"
#endif
#ifndef TAIL_BWD_SRC
#define TAIL_BWD_SRC "../m2trans_amd/csrc/k_tail_bwd.hip"
#endif
#include TAIL_FWD_SRC
#include TAIL_BWD_SRC
#ifndef NO_STREAM
#define TS_BENCH_HOOKS
#include "../m2trans_amd/csrc/k_tail_stream.hip"
#define WITH_BWD_STREAM
#include "../m2trans_amd/csrc/k_tail_bwd_stream.hip"
#endif
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
static unsigned long long fnv(const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
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

static int run_case(int B, int Hlr, int Wlr, bool timing) {
  const int H = 4 * Hlr, W = 4 * Wlr, Hm = H / 2, Wm = W / 2;
  const size_t nmid = (size_t)B * Hm * Wm * 64, nhr = (size_t)B * 3 * H * W;
  srand(7 + B + Hlr);
  std::vector<unsigned short> ha1(nmid), hd1(nmid), hw3(256 * 64), hw3t(64 * 256);
  for (auto& v : ha1) { float t = 3.f * frand(); v = f2bf(0.5f * t * (1.f + erff(t * 0.70710678f))); }
  for (auto& v : hd1) v = f2bf(0.5f + frand());
  for (int n = 0; n < 256; ++n) for (int k = 0; k < 64; ++k) { unsigned short v = f2bf(0.25f * frand()); hw3[n * 64 + k] = v; hw3t[k * 256 + n] = v; }
  std::vector<float> hb3(256), hwf(3 * 64 * 9), hg(nhr);
  for (auto& v : hb3) v = 0.2f * frand();
  for (auto& v : hwf) v = 0.1f * frand();
  for (auto& v : hg) v = (rand() % 7 == 0) ? 0.f : 1e-3f * frand();
  void *da1, *dd1, *dw3, *dw3t, *dgt1; float *db3, *dwf, *dg, *dout, *swf, *sw3, *sb3;
  CKH(hipMalloc(&da1, nmid * 2)); CKH(hipMalloc(&dd1, nmid * 2)); CKH(hipMalloc(&dgt1, nmid * 2));
  CKH(hipMalloc(&dw3, 256 * 64 * 2)); CKH(hipMalloc(&dw3t, 256 * 64 * 2));
  CKH(hipMalloc(&db3, 1024)); CKH(hipMalloc(&dwf, hwf.size() * 4)); CKH(hipMalloc(&dg, nhr * 4)); CKH(hipMalloc(&dout, nhr * 4));
  const int nb = tail_bwd_fused_blocks(B, H, W);
  const int nbmax = 2048;
  CKH(hipMalloc(&swf, (size_t)nbmax * 32 * 64 * 4)); CKH(hipMalloc(&sw3, (size_t)nbmax * 256 * 64 * 4)); CKH(hipMalloc(&sb3, (size_t)nbmax * 256 * 4));
  CKH(hipMemcpy(da1, ha1.data(), nmid * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dd1, hd1.data(), nmid * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw3, hw3.data(), 256 * 64 * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dw3t, hw3t.data(), 256 * 64 * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(db3, hb3.data(), 1024, hipMemcpyHostToDevice)); CKH(hipMemcpy(dwf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dg, hg.data(), nhr * 4, hipMemcpyHostToDevice));
  CKH(hipMemset(dout, 0xff, nhr * 4)); CKH(hipMemset(dgt1, 0xff, nmid * 2));
  hipStream_t st; CKH(hipStreamCreate(&st));
  float* dsink; CKH(hipMalloc(&dsink, 4096 + 8192 * 4 * 16));
  int ns = 0;
  auto fwd = [&]() { if (launch_tail_fwd_fused(da1, dw3, db3, dwf, dout, B, H, W, st)) exit(2); };
  auto bwd = [&]() { if (launch_tail_bwd_fused(dg, dwf, nullptr, nullptr, da1, dd1, dw3t, db3, dgt1, swf, sw3, sb3, &ns, B, H, W, st)) exit(2); };
  fwd(); bwd();
  CKH(hipStreamSynchronize(st));
  (void)nb;
  std::vector<float> ho(nhr); std::vector<unsigned short> hgt(nmid);
  CKH(hipMemcpy(ho.data(), dout, nhr * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(hgt.data(), dgt1, nmid * 2, hipMemcpyDeviceToHost));
  // slab sums in a fixed order (the slab COUNT may differ between builds; fp64 sums compared to ~1e-6)
  std::vector<float> h1((size_t)ns * 32 * 64), h2((size_t)ns * 256 * 64), h3((size_t)ns * 256);
  CKH(hipMemcpy(h1.data(), swf, h1.size() * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(h2.data(), sw3, h2.size() * 4, hipMemcpyDeviceToHost));
  CKH(hipMemcpy(h3.data(), sb3, h3.size() * 4, hipMemcpyDeviceToHost));
  auto slabsum = [&](const std::vector<float>& s, size_t n, double& l1, double& probe) {
    l1 = 0; probe = 0;
    for (size_t i = 0; i < n; ++i) { double a = 0; for (int k = 0; k < ns; ++k) a += s[(size_t)k * n + i]; l1 += fabs(a); probe += a * ((i * 2654435761u % 1000) / 1000.0 - 0.5); }
  };
  double a1, p1, a2, p2, a3, p3;
  slabsum(h1, 32 * 64, a1, p1); slabsum(h2, 256 * 64, a2, p2); slabsum(h3, 256, a3, p3);
  double osum = 0; for (float v : ho) osum += fabs(v);
  printf("B=%d LR %dx%d: out hash %016llx (|out| %.6e)  gt1 hash %016llx  slabs=%d dWf %.9e/%.9e dW3 %.9e/%.9e db3 %.9e/%.9e\n", B, Hlr, Wlr, fnv(ho.data(), nhr * 4), osum,
         fnv(hgt.data(), nmid * 2), ns, a1, p1, a2, p2, a3, p3);
#ifdef WITH_BWD_STREAM
  {
    // streaming backward against the tile kernel: g(t1) bit for bit, the three parameter gradients to fp32 summation order
    std::vector<unsigned short> hgt2(nmid);
    CKH(hipMemset(dgt1, 0xee, nmid * 2));
    int ns2 = 0;
    if (launch_tail_bwd_stream(dg, dwf, da1, dd1, dw3t, db3, dgt1, swf, sw3, sb3, &ns2, B, Hm, Wm, 2, 0, st)) exit(2);
    CKH(hipStreamSynchronize(st));
    CKH(hipMemcpy(hgt2.data(), dgt1, nmid * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < nmid; ++i) if (hgt[i] != hgt2[i]) { if (!bad) first = i; ++bad; }
    std::vector<float> k1((size_t)ns2 * 32 * 64), k2((size_t)ns2 * 256 * 64), k3((size_t)ns2 * 256);
    CKH(hipMemcpy(k1.data(), swf, k1.size() * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(k2.data(), sw3, k2.size() * 4, hipMemcpyDeviceToHost));
    CKH(hipMemcpy(k3.data(), sb3, k3.size() * 4, hipMemcpyDeviceToHost));
    auto cmp = [&](const std::vector<float>& o, const std::vector<float>& n_, size_t n) {
      double num = 0, den = 0;
      for (size_t i = 0; i < n; ++i) {
        double a = 0, b = 0;
        for (int k = 0; k < ns; ++k) a += o[(size_t)k * n + i];
        for (int k = 0; k < ns2; ++k) b += n_[(size_t)k * n + i];
        num += (a - b) * (a - b); den += a * a;
      }
      return sqrt(num / (den + 1e-300));
    };
    printf("   stream bwd: g(t1) %zu of %zu differ", bad, nmid);
    if (bad) printf(" (first %zu: pixel %zu ch %zu: %04x vs %04x)", first, first / 64, first % 64, hgt[first], hgt2[first]);
    printf("; slabs %d; rel diff dWf %.2e dW3 %.2e db3 %.2e\n", ns2, cmp(h1, k1, 32 * 64), cmp(h2, k2, 256 * 64), cmp(h3, k3, 256));
    if (timing) {
      auto bs = [&]() { int q; if (launch_tail_bwd_stream(dg, dwf, da1, dd1, dw3t, db3, dgt1, swf, sw3, sb3, &q, B, Hm, Wm, 2, 0, st)) exit(2); };
      printf("   tail_bwd_stream %.1f us\n", time_it(st, 30, bs));
#ifdef BSTAMPS
      {
        unsigned long long* ds; const size_t nst = (size_t)512 * 4 * 16;
        CKH(hipMalloc(&ds, nst * 8)); CKH(hipMemset(ds, 0, nst * 8));
        CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &ds, sizeof(ds)));
        bs(); CKH(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs(nst);
        CKH(hipMemcpy(hs.data(), ds, nst * 8, hipMemcpyDeviceToHost));
        double acc[16] = {0}; int cnt = 0;
        for (int blk = 0; blk < 512; ++blk) for (int it = 0; it < 3; ++it) {
          const unsigned long long* a = &hs[((size_t)blk * 4 + it) * 16], *nx = &hs[((size_t)blk * 4 + it + 1) * 16];
          if (!a[0] || !nx[0]) continue;
          for (int i = 0; i < 8; ++i) acc[i] += (double)(a[i + 1] - a[i]);
          acc[8] += (double)(nx[0] - a[8]); acc[9] += (double)(nx[0] - a[0]); ++cnt;
        }
        printf("   bwd stream stamps (wave %d, mean cycles over %d steps): ", BSTAMP_WAVE, cnt);
        for (int i = 0; i < 10; ++i) printf("%s%.0f", i ? " | " : "", acc[i] / std::max(cnt, 1));
        printf("\n   [0 loads+MFMA t2 | 1 Geff gather | 2 GELU, g(a2), g(t2), stores | 3 barrier 1 | 4 ring write, dW3 | 5 dWf | 6 g(t1) | 7 barrier 2 | 8 loop | 9 total]\n");
        unsigned long long* z = nullptr; CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &z, sizeof(z)));
      }
#endif
    }
  }
#endif
#ifndef NO_STREAM
  for (int segr : {0, 24, 64}) {
    CKH(hipMemset(dout, 0xff, nhr * 4));
    if (launch_tail_fwd_stream(da1, 0, dw3, db3, dwf, dout, B, Hm, Wm, 2, segr, st)) exit(2);
    CKH(hipStreamSynchronize(st));
    std::vector<float> ho2(nhr);
    CKH(hipMemcpy(ho2.data(), dout, nhr * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < nhr; ++i) if (memcmp(&ho[i], &ho2[i], 4)) { if (!bad) first = i; ++bad; }
    printf("   stream fwd (seg rows %d): %zu of %zu outputs differ from the tile kernel", segr, bad, nhr);
    if (bad) printf(" (first: image %zu ch %zu row %zu col %zu: %g vs %g)", first / ((size_t)3 * H * W), (first / ((size_t)H * W)) % 3, (first / W) % H, first % W, ho[first], ho2[first]);
    printf("\n");
  }
  if (timing) {
    int nblk = -1;
    CKH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, tail_fwd_stream_kernel<2, false>, 256, TSCfg<2>::total));
    printf("   occupancy API: %d workgroups of 256 threads per CU (LDS %zu B)\n", nblk, TSCfg<2>::total);
  }
  if (timing)
    for (int wgs : {1, 2, 3, 4}) {
      g_ts_pad_lds = (wgs == 4) ? 0 : (160 * 1024 / wgs - (int)TSCfg<2>::total - 512);
      auto fs = [&]() { if (launch_tail_fwd_stream(da1, 0, dw3, db3, dwf, dout, B, Hm, Wm, 2, 52, st)) exit(2); };
      printf("   tail_fwd_stream, %d workgroups per CU (LDS pad %d): %.1f us\n", wgs, g_ts_pad_lds, time_it(st, 20, fs));
#ifdef TS_STAMPS
      {
        const int ntask = B * ((Wm - 16 + 14) / 15 + 1) * 5, nw = ntask * 4;    // (upper bound on tasks is fine: zeros are skipped)
        std::vector<unsigned long long> hq((size_t)8192 * 8);
        CKH(hipMemcpy(hq.data(), (char*)dsink + 4096, hq.size() * 8 > 8192 * 4 * 16 ? 8192 * 4 * 16 : hq.size() * 8, hipMemcpyDeviceToHost));
        double c = 0, r = 0; int n = 0;
        for (int i = 0; i < 8000; ++i) if (hq[2 * i] && hq[2 * i + 1]) { c += hq[2 * i]; r += hq[2 * i + 1]; ++n; }
        (void)nw;
        if (n) printf("      per wave: %.0f shader cycles per task (%d waves), clock %.3f GHz\n", c / n, n, c / r * 0.1);
      }
#endif
      g_ts_pad_lds = 0;
    }
  if (timing)
    for (int segr : {0, 24, 32, 52, 64, 128}) {
      auto fs = [&]() { if (launch_tail_fwd_stream(da1, 0, dw3, db3, dwf, dout, B, Hm, Wm, 2, segr, st)) exit(2); };
      printf("   tail_fwd_stream seg rows %3d: %.1f us\n", segr, time_it(st, 30, fs));
    }
#endif
  if (timing) {
    const float tf = time_it(st, 30, fwd), tb = time_it(st, 30, bwd);
    printf("   tail_fwd_fused %.1f us   tail_bwd_fused %.1f us\n", tf, tb);
#ifdef STAMPS
    {
      unsigned long long* ds; const size_t nst = (size_t)4096 * 4 * 16;
      CKH(hipMalloc(&ds, nst * 8)); CKH(hipMemset(ds, 0, nst * 8));
      CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &ds, sizeof(ds)));
      bwd(); CKH(hipStreamSynchronize(st));
      std::vector<unsigned long long> hs(nst);
      CKH(hipMemcpy(hs.data(), ds, nst * 8, hipMemcpyDeviceToHost));
      double acc[16] = {0}; int cnt = 0;
      for (int blk = 0; blk < ns; ++blk) for (int it = 0; it < 3; ++it) {
        const unsigned long long* a = &hs[((size_t)blk * 4 + it) * 16], *nx = &hs[((size_t)blk * 4 + it + 1) * 16];
        if (!a[0] || !nx[0]) continue;
        for (int i = 0; i < 10; ++i) acc[i] += (double)(a[i + 1] - a[i]);
        acc[10] += (double)(nx[0] - a[10]); acc[11] += (double)(nx[0] - a[0]); ++cnt;
      }
      printf("   bwd stamps (wave %d, mean cycles over %d tiles): ", STAMP_WAVE, cnt);
      for (int i = 0; i < 12; ++i) printf("%s%.0f", i ? " | " : "", acc[i] / std::max(cnt, 1));
      printf("\n   [0 stage->bar1 | 1 prefetch issue | 2 recompute+gelu | 3 Geff | 4 bar2 | 5 g(t2) | 6 dWf | 7 bar3 | 8 g(t1) | 9 fetchB+dW3 | 10 bar4 | 11 total]\n");
    }
#endif
  }
  for (void* q : {da1, dd1, dgt1, dw3, dw3t, (void*)db3, (void*)dwf, (void*)dg, (void*)dout, (void*)swf, (void*)sw3, (void*)sb3}) (void)hipFree(q);
  return 0;
}

#ifndef NO_STREAM
// x3 (R = 3, P64 input): timing of the forward / backward row-streaming pair on the BASELINE configs[4] shape; no reference here (the
// library's -m gpu tests compare them with the plain kernels)
static void run_r3(int B, int Hlr, int Wlr) {
  const int R = 3, H = R * Hlr, W = R * Wlr;
  const size_t nlr = (size_t)B * Hlr * Wlr * 64, nhr = (size_t)B * 3 * H * W;
  std::vector<unsigned short> ha(nlr), hw(576 * 64), hwt(64 * 576);
  for (auto& v : ha) v = f2bf(2.f * frand());
  for (int n = 0; n < 576; ++n) for (int k = 0; k < 64; ++k) { unsigned short v = f2bf(0.25f * frand()); hw[n * 64 + k] = v; hwt[k * 576 + n] = v; }
  std::vector<float> hb(576), hwf(3 * 64 * 9), hg(nhr);
  for (auto& v : hb) v = 0.2f * frand();
  for (auto& v : hwf) v = 0.1f * frand();
  for (auto& v : hg) v = 1e-3f * frand();
  void *da, *dw, *dwt, *dga; float *db, *dwf, *dg, *dout, *s1, *s2, *s3;
  CKH(hipMalloc(&da, nlr * 2)); CKH(hipMalloc(&dga, nlr * 2)); CKH(hipMalloc(&dw, hw.size() * 2)); CKH(hipMalloc(&dwt, hwt.size() * 2));
  CKH(hipMalloc(&db, 576 * 4)); CKH(hipMalloc(&dwf, hwf.size() * 4)); CKH(hipMalloc(&dg, nhr * 4)); CKH(hipMalloc(&dout, nhr * 4));
  CKH(hipMalloc(&s1, (size_t)512 * 32 * 64 * 4)); CKH(hipMalloc(&s2, (size_t)512 * 576 * 64 * 4)); CKH(hipMalloc(&s3, (size_t)512 * 576 * 4));
  CKH(hipMemcpy(da, ha.data(), nlr * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dwt, hwt.data(), hwt.size() * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(db, hb.data(), 576 * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dwf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice)); CKH(hipMemcpy(dg, hg.data(), nhr * 4, hipMemcpyHostToDevice));
  hipStream_t st; CKH(hipStreamCreate(&st));
  auto fwd = [&]() { if (launch_tail_fwd_stream(da, 1, dw, db, dwf, dout, B, Hlr, Wlr, R, 0, st)) exit(2); };
  auto bwd = [&]() { int q; if (launch_tail_bwd_stream(dg, dwf, da, nullptr, dwt, db, dga, s1, s2, s3, &q, B, Hlr, Wlr, R, 1, st)) exit(2); };
  printf("x3 B=%d LR %dx%d: tail_fwd_stream<3> %.1f us   tail_bwd_stream<3> %.1f us\n", B, Hlr, Wlr, time_it(st, 20, fwd), time_it(st, 20, bwd));
#ifdef BSTAMPS
  {
    unsigned long long* ds; const size_t nst = (size_t)512 * 4 * 16;
    CKH(hipMalloc(&ds, nst * 8)); CKH(hipMemset(ds, 0, nst * 8));
    CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &ds, sizeof(ds)));
    bwd(); CKH(hipStreamSynchronize(st));
    std::vector<unsigned long long> hs(nst);
    CKH(hipMemcpy(hs.data(), ds, nst * 8, hipMemcpyDeviceToHost));
    double acc[16] = {0}; int cnt = 0;
    for (int blk = 0; blk < 512; ++blk) for (int it = 0; it < 3; ++it) {
      const unsigned long long* a = &hs[((size_t)blk * 4 + it) * 16], *nx = &hs[((size_t)blk * 4 + it + 1) * 16];
      if (!a[0] || !nx[0]) continue;
      for (int i = 0; i < 8; ++i) acc[i] += (double)(a[i + 1] - a[i]);
      acc[8] += (double)(nx[0] - a[8]); acc[9] += (double)(nx[0] - a[0]); ++cnt;
    }
    printf("   x3 bwd stamps (wave %d, mean cycles over %d steps): ", BSTAMP_WAVE, cnt);
    for (int i = 0; i < 10; ++i) printf("%s%.0f", i ? " | " : "", acc[i] / std::max(cnt, 1));
    printf("\n   [0 loads+MFMA t | 1 Geff gather | 2 GELU, g(act), g(t), stores | 3 barrier 1 | 4 ring write, db | 5 dWf | 6 g(a) | 7 barrier 2 | 8 loop | 9 total]\n");
    unsigned long long* z = nullptr; CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), &z, sizeof(z)));
  }
#endif
  for (void* q : {da, dga, dw, dwt, (void*)db, (void*)dwf, (void*)dg, (void*)dout, (void*)s1, (void*)s2, (void*)s3}) (void)hipFree(q);
}
#endif

int main(int argc, char** argv) {
#ifndef NO_STREAM
  if (argc > 1 && argv[1][0] == '3') { run_r3(8, 256, 256); return 0; }
#endif
  run_case(2, 16, 24, false);       // 64 x 96 HR: border tiles only
  run_case(3, 32, 24, false);
  run_case(1, 40, 64, false);
  run_case(16, 128, 128, true);     // BASELINE configs[1]
  if (argc > 1) run_case(32, 128, 128, true);
  return 0;
}
