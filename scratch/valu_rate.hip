// VALU issue-rate probe (gfx950): cycles per wave-instruction for v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_med3_f32,
// v_cvt_pk_bf16_f32 with 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 scratch/valu_rate.hip -o scratch/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP> __global__ void probe(unsigned long long* out, float seed) {
  float a[8]; f32x2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (f32x2){seed + i, seed - i}; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
        if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
        if (OP == 7) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  if (threadIdx.x % 64 == 0) out[blockIdx.x * 32 + threadIdx.x / 64] = t1 - t0;
  if (s == 12345.678f) out[0] = 0;
}
template <int OP> void run(const char* name) {
  unsigned long long* d; hipMalloc(&d, 256 * 32 * 8);
  printf("%-20s", name);
  for (int waves : {4, 8, 16}) {          // waves per workgroup = per CU (one workgroup per CU): 1, 2, 4 per SIMD
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 32);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { m += h[b * 32 + w]; ++n; }
    // cycles per wave-instruction as seen by ONE wave, and SIMD cycles per instruction (divide by waves per SIMD)
    const double per = m / n / (64.0 * REP);
    printf("  %dw/SIMD: %.2f cyc/inst/wave = %.2f SIMD-cyc/inst", waves / 4, per, per / (waves / 4));
  }
  printf("\n");
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<7>("v_mul_f32"); run<6>("v_pk_mul_f32"); run<8>("v_pk_add_f32"); run<2>("v_exp_f32"); run<3>("v_rcp_f32"); run<4>("v_med3_f32"); run<5>("v_cvt_pk_bf16_f32");
  return 0;
}
