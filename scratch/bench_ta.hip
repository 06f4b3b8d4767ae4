// What one vector-memory load instruction costs a CU (gfx950): 256 workgroups of NW waves, every thread issues NL loads of width W bytes in a
// row (no arithmetic between them), s_memtime from the first issue to the last arrival.  Patterns:
//   0  contiguous over the wave (lane stride = W), instruction stride = 64 W
//   1  "pixel" pattern of the P64 planes: 4 lanes cover 4 W contiguous bytes, the next 4 lanes are 16 W bytes further (128 at W = 8)
//   2  every lane the same address (L1 broadcast)
//   3  contiguous, but every wave of the workgroup reads the same bytes (L1 hits after the first wave)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/bench_ta.hip -o scratch/bench_ta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W> struct Vec;
template <> struct Vec<4> { typedef float t; static __device__ float sum(float v) { return v; } };
template <> struct Vec<8> { typedef f32x2 t; static __device__ float sum(f32x2 v) { return v[0] + v[1]; } };
template <> struct Vec<16> { typedef f32x4 t; static __device__ float sum(f32x4 v) { return v[0] + v[1] + v[2] + v[3]; } };

template <int W, int NL, int PAT>
__global__ void __launch_bounds__(512) ta_kernel(const char* __restrict__ src, size_t wg_bytes, int hot, float* out, unsigned long long* stamps) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const char* base = src + (size_t)(hot ? (blockIdx.x % hot) : blockIdx.x) * wg_bytes;
  unsigned off;
  if (PAT == 0) off = (wv * NL * 64 + lane) * W;
  else if (PAT == 1) off = wv * NL * 64 * W + (lane >> 2) * (16 * W) + (lane & 3) * W;
  else if (PAT == 2) off = wv * NL * 64 * W;
  else off = lane * W;
  typename Vec<W>::t v[NL];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    unsigned o;
    if (PAT == 1) o = off + (i & 3) * (4 * W) + (i >> 2) * (16 * 16 * W);        // 4 instructions fill the 16 lines they touch (at W = 8), then the next 16 lines
    else o = off + i * 64 * W;
    v[i] = *reinterpret_cast<const typename Vec<W>::t*>(base + o);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += Vec<W>::sum(v[i]);
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (s == 12345.678f) out[0] = s;
  if (lane == 0) { stamps[((size_t)blockIdx.x * 8 + wv) * 2] = t1 - t0; stamps[((size_t)blockIdx.x * 8 + wv) * 2 + 1] = t2 - t0; }
}

template <int W, int NL, int PAT>
static void run(const char* name, int nw, const char* dsrc, size_t wg_bytes, int hot, float* dout, unsigned long long* dst) {
  const int nwg = 256;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ta_kernel<W, NL, PAT>), dim3(nwg), dim3(nw * 64), 0, 0, dsrc, wg_bytes, hot, dout, dst);
  CKH(hipDeviceSynchronize());
  hipEvent_t e0, e1; CKH(hipEventCreate(&e0)); CKH(hipEventCreate(&e1));
  CKH(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((ta_kernel<W, NL, PAT>), dim3(nwg), dim3(nw * 64), 0, 0, dsrc, wg_bytes, hot, dout, dst);
  CKH(hipEventRecord(e1, 0)); CKH(hipEventSynchronize(e1));
  float ms; CKH(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> hs((size_t)nwg * 8 * 2);
  CKH(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
  std::vector<long long> issue, land;
  for (int b = 0; b < nwg; ++b) { long long mi = 0, ml = 0; for (int w = 0; w < nw; ++w) { mi = std::max(mi, (long long)hs[((size_t)b * 8 + w) * 2]); ml = std::max(ml, (long long)hs[((size_t)b * 8 + w) * 2 + 1]); } issue.push_back(mi); land.push_back(ml); }
  std::sort(issue.begin(), issue.end()); std::sort(land.begin(), land.end());
  const double instr = (double)nw * NL;
  printf("%-34s W=%2d NL=%2d waves=%d hot=%3d: all issued %6lld, all landed %6lld cycles = %5.1f clk / wave-instruction, %5.1f B/clk per CU  (kernel %.1f us)\n", name, W, NL, nw, hot,
         issue[nwg / 2], land[nwg / 2], land[nwg / 2] / instr, instr * 64 * W / land[nwg / 2], ms * 1e3);
}

int main() {
  const size_t wg_bytes = 8 * 64 * 64 * 16 + 65536;      // the largest footprint of one workgroup (8 waves x 64 loads x 1 KB) + slack
  char* dsrc; float* dout; unsigned long long* dst;
  CKH(hipMalloc(&dsrc, wg_bytes * 256)); CKH(hipMemset(dsrc, 0, wg_bytes * 256)); CKH(hipMalloc(&dout, 16)); CKH(hipMalloc(&dst, 256 * 8 * 2 * 8));
  for (int hot : {0, 4}) {
    for (int nw : {8, 4}) {
      run<16, 32, 0>("contiguous", nw, dsrc, wg_bytes, hot, dout, dst);
      run<8, 32, 0>("contiguous", nw, dsrc, wg_bytes, hot, dout, dst);
      run<4, 32, 0>("contiguous", nw, dsrc, wg_bytes, hot, dout, dst);
      run<8, 32, 1>("pixel pattern (4 lanes / 128 B)", nw, dsrc, wg_bytes, hot, dout, dst);
      run<16, 32, 1>("pixel pattern (4 lanes / 128 B)", nw, dsrc, wg_bytes, hot, dout, dst);
      run<16, 32, 2>("one address per wave", nw, dsrc, wg_bytes, hot, dout, dst);
      run<16, 32, 3>("same KB in every wave", nw, dsrc, wg_bytes, hot, dout, dst);
      run<8, 32, 3>("same KB in every wave", nw, dsrc, wg_bytes, hot, dout, dst);
    }
    run<16, 8, 0>("contiguous, short", 8, dsrc, wg_bytes, hot, dout, dst);
    run<16, 60, 0>("contiguous, long", 8, dsrc, wg_bytes, hot, dout, dst);
    run<8, 60, 1>("pixel pattern, long", 4, dsrc, wg_bytes, hot, dout, dst);
  }
  return 0;
}
