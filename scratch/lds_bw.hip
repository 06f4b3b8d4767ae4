// LDS read throughput per CU for ds_read_b128 / b64 / b32 (conflict-free and the kernels' 144-byte row stride).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, int iters, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 256) ((float*)lds)[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float acc = 0.f;
  if (MODE == 0) {        // b128: lane (lr, g): row lr stride, 16 B at g*16
    const int off = ((lane & 15) * stride_bytes + (lane >> 4) * 16 + wv * 4096) & 0xFFF0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const f32x4 v = *(const f32x4*)(lds + ((off + u * 2304 + it * 4608) & 0xFFF0)); acc += v[0] + v[3]; }
    }
  } else if (MODE == 1) { // b64
    const int off = ((lane & 15) * stride_bytes + (lane >> 4) * 8 + wv * 4096) & 0xFFF8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { const f32x2 v = *(const f32x2*)(lds + ((off + u * 1160 + it * 4640) & 0xFFF8)); acc += v[0] + v[1]; }
    }
  } else {                // b32 linear
    const int off = (lane * 4 + wv * 4096) & 0xFFFC;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) { acc += *(const float*)(lds + ((off + u * 260 + it * 4160) & 0xFFFC)); }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int MODE> void run(const char* name, int stride) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(d, 10, stride); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters, stride); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 256 * iters * 128.0;   // every mode reads 128 B per thread per iteration
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("%-28s stride %4d: %.2f TB/s aggregate = %.1f B/clk/CU (at %.2f GHz, 256 CUs)\n", name, stride, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / (clk * 1e3), clk / 1e6);
  hipFree(d);
}
int main() {
  run<0>("ds_read_b128", 16); run<0>("ds_read_b128", 144); run<0>("ds_read_b128", 272);
  run<1>("ds_read_b64", 8); run<1>("ds_read_b64", 144);
  run<2>("ds_read_b32 linear", 4);
  return 0;
}
