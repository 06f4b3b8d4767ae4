// Stand-alone timing harness for the 3x3 conv kernels of k_conv.hip (no torch): B=16, 128x128, P64 bf16 feature maps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DSTAMPS] scratch/bench_conv.hip -o scratch/bench_conv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#ifdef STAMPS
__device__ unsigned long long* g_stamps;
#define M2T_CONV_STAMP(i) do { if (threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
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
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, H = 128, W = 128;
  const int variant = argc > 2 ? atoi(argv[2]) : 1;
  const bool with_res = argc > 3 ? atoi(argv[3]) != 0 : true;
  const bool persistent = argc > 4 ? atoi(argv[4]) != 0 : false;
  const size_t n = (size_t)B * H * W * 64;
  std::vector<unsigned short> hx(n), hw(9 * 64 * 64);
  srand(1);
  for (auto& v : hx) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
  std::vector<float> hb(64, 0.1f);
  void *dx, *dw, *dr, *dy; float* db;
  CKH(hipMalloc(&dx, n * 2)); CKH(hipMalloc(&dr, n * 2)); CKH(hipMalloc(&dy, n * 2)); CKH(hipMalloc(&dw, hw.size() * 2)); CKH(hipMalloc(&db, 256));
  CKH(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(dr, hx.data(), n * 2, hipMemcpyHostToDevice));
  CKH(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CKH(hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice));
  const int ntiles = B * (H / 8) * (W / 16);
#ifdef STAMPS
  unsigned long long* dst;
  CKH(hipMalloc(&dst, (size_t)4096 * 8 * 8)); CKH(hipMemset(dst, 0, (size_t)4096 * 8 * 8));
  CKH(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
#endif
  hipStream_t st; CKH(hipStreamCreate(&st));
  auto launch = [&]() { return launch_conv3x3_c64(M2T_BF16, dx, dw, with_res ? db : nullptr, with_res ? dr : nullptr, nullptr, dy, B, H, W, st, persistent, variant); };
  for (int i = 0; i < 5; ++i) if (launch()) return 1;
  CKH(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  const int N = 40; std::vector<float> ts;
  for (int i = 0; i < N; ++i) {
    CKH(hipEventRecord(e0, st)); if (launch()) return 1; CKH(hipEventRecord(e1, st)); CKH(hipEventSynchronize(e1));
    float ms; CKH(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)n * 2 * (with_res ? 3 : 2);
  printf("conv3x3 B=%d variant=%d res=%d tiles=%d: event-bracketed min %.2f us median %.2f us -> %.2f TB/s algorithmic, %.0f TFLOP/s\n", B, variant, (int)with_res, ntiles,
         ts[0], ts[N / 2], bytes / ts[N / 2] * 1e-6, 2.0 * B * H * W * 64 * 576 / ts[N / 2] * 1e-6);
#ifdef STAMPS
  std::vector<unsigned long long> hs((size_t)4096 * 8);
  CKH(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
  const char* names[4] = {"start -> 2nd pair staged", "nine taps", "epilogue (res wait, stores)", "barrier"};
  unsigned long long tmin = ~0ull, tmax = 0;
  int nb = 0;
  for (int b = 0; b < 4096; ++b) if (hs[(size_t)b * 8]) { ++nb; tmin = std::min(tmin, hs[(size_t)b * 8]); tmax = std::max(tmax, hs[(size_t)b * 8 + 3]); }
  if (persistent) names[0] = "start -> 2nd pair staged";
  printf("%d workgroups stamped; first start -> last end: %llu cycles\n", nb, tmax - tmin);
  for (int s = 0; s < ((persistent || variant == 2) ? 4 : 3); ++s) {
    std::vector<long long> d;
    for (int b = 0; b < 4096; ++b) { const unsigned long long a = hs[(size_t)b * 8 + s], c = hs[(size_t)b * 8 + s + 1]; if (a && c) d.push_back((long long)(c - a)); }
    std::sort(d.begin(), d.end());
    printf("  %-30s median %8lld (min %lld max %lld)\n", names[s], d[d.size() / 2], d.front(), d.back());
  }
  // start-time distribution: how many rounds?
  std::vector<long long> starts;
  for (int b = 0; b < 4096; ++b) if (hs[(size_t)b * 8]) starts.push_back((long long)(hs[(size_t)b * 8] - tmin));
  std::sort(starts.begin(), starts.end());
  printf("  workgroup start times (cycles after the first): 10%% %lld, 50%% %lld, 90%% %lld, last %lld\n", starts[starts.size() / 10], starts[starts.size() / 2], starts[starts.size() * 9 / 10], starts.back());
#endif
  return 0;
}
