// Fork a side stream behind a main-stream kernel WITHOUT recording an event on the main stream: the NEXT main-stream kernel writes a
// sequence number to signal memory (hipMallocSignalMemory) as its first action, the side stream waits for it with hipStreamWaitValue32.
// Measures the idle time between two dependent main-stream kernels for (a) nothing in between, (b) an event record + side wait,
// (c) the flag scheme; checks that the side kernel starts after the first main kernel has finished.
//   hipcc --offload-arch=gfx950 -O3 scratch/test_waitvalue.hip -o scratch/test_waitvalue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work(unsigned long long* stamps, int slot, float* buf, int iters, unsigned* flag, unsigned seq) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * slot] = wall_clock64();
  float v = buf[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  buf[blockIdx.x * blockDim.x + threadIdx.x] = v;
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}
int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned* flag = nullptr; CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory)); CK(hipMemset(flag, 0, 8));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 8 * 64)); CK(hipMemset(stamps, 0, 8 * 64));
  float *ba, *bb; CK(hipMalloc(&ba, 4 * 256 * 1024)); CK(hipMalloc(&bb, 4 * 256 * 1024)); CK(hipMemset(ba, 0, 4 * 256 * 1024)); CK(hipMemset(bb, 0, 4 * 256 * 1024));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  unsigned seq = 0;
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<double> gaps, side_lag; int bad = 0;
    for (int rep = 0; rep < 40; ++rep) {
      ++seq;
      CK(hipMemsetAsync(stamps, 0, 8 * 64, a)); CK(hipStreamSynchronize(a));
      hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, a, stamps, 0, ba, 60000, (unsigned*)nullptr, 0u);           // K1 (main)
      if (mode == 1) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); }
      if (mode == 2) CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(work, dim3(1024), dim3(256), 0, a, stamps, 1, ba, 4000, mode == 2 ? flag : nullptr, seq);    // K2 (main): writes the flag
      if (mode > 0) hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, b, stamps, 2, bb, 1000, (unsigned*)nullptr, 0u);  // K3 (side)
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      unsigned long long h[6]; CK(hipMemcpy(h, stamps, 48, hipMemcpyDeviceToHost));
      gaps.push_back((double)(h[2] - h[1]) / 100.0);                 // wall_clock64: 100 MHz -> us
      if (mode > 0) { side_lag.push_back((double)((long long)h[4] - (long long)h[1]) / 100.0); if (h[4] < h[1]) ++bad; }
    }
    std::sort(gaps.begin(), gaps.end()); std::sort(side_lag.begin(), side_lag.end());
    printf("mode %d (%s): idle between the two main kernels: median %.2f us (min %.2f, max %.2f)", mode,
           mode == 0 ? "nothing in between" : (mode == 1 ? "event record + side wait" : "flag written by the next kernel + hipStreamWaitValue32"), gaps[20], gaps[0], gaps[39]);
    if (mode > 0) printf("; side kernel starts %.2f us (median) after the first main kernel ends, %d of 40 started EARLY", side_lag[20], bad);
    printf("\n");
  }
  return 0;
}
