// IO floor of the C=256 qkv projection: read A [16384][256] bf16 tile-wise, write Y [16384][768] with the GEMM's store pattern, no math.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) io_kernel(const bf16_t* __restrict__ A, bf16_t* __restrict__ Y, int mode) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;
  f32x4 acc = {0, 0, 0, 0};
  if (mode & 1) {
    for (int k0 = 0; k0 < 256; k0 += 64)
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + (m0 + (idx >> 3)) * 256 + k0 + (idx & 7) * 8);
        acc += v;
      }
  }
  for (int mt = 0; mt < 2; ++mt) {
    const long long m = m0 + 32 * wv + 16 * mt + lr;
    for (int h = 0; h < 2; ++h) {
      f32x4* p = reinterpret_cast<f32x4*>(Y + m * 768 + n0 + 64 * h + 16 * g);
      p[0] = acc; p[1] = acc;
    }
  }
}
int main() {
  bf16_t *A, *Y;
  hipMalloc(&A, 16384LL * 256 * 2); hipMalloc(&Y, 16384LL * 768 * 2);
  hipMemset(A, 0, 16384LL * 256 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(io_kernel, dim3(128, 6), dim3(256), 0, 0, A, Y, mode);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(io_kernel, dim3(128, 6), dim3(256), 0, 0, A, Y, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s): %.2f us per launch\n", mode, mode ? "read A tile + write Y" : "write Y only", ms / 50 * 1e3);
  }
  return 0;
}
