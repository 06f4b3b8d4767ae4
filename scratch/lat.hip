#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void addk(const float4* a, const float4* b, float4* o, long long n4){
  for (long long t = blockIdx.x*(long long)blockDim.x+threadIdx.x; t<n4; t+=(long long)gridDim.x*blockDim.x){ float4 x=a[t], y=b[t]; x.x+=y.x;x.y+=y.y;x.z+=y.z;x.w+=y.w; o[t]=x; }
}
__global__ void chain(const int* idx, float* o, int hops){ int i=threadIdx.x+blockIdx.x*blockDim.x; int j=i; for(int h=0;h<hops;++h) j=idx[j]; o[i]=(float)j; }
int main(){
  const long long N = 64ll<<20; float *a,*b,*o; hipMalloc(&a,N*4);hipMalloc(&b,N*4);hipMalloc(&o,N*4); hipMemset(a,0,N*4);hipMemset(b,0,N*4);
  hipEvent_t e0,e1; hipEventCreate(&e0);hipEventCreate(&e1);
  for (long long n : {256ll, 1ll<<16, 1ll<<20, 1ll<<22, 1ll<<24, 1ll<<26}) {
    long long n4=n/4; int grid=(int)std::min<long long>((n4+255)/256, 4096); if(grid<1)grid=1;
    for(int i=0;i<20;i++) addk<<<grid,256>>>((float4*)a,(float4*)b,(float4*)o,n4);
    hipEventRecord(e0); for(int i=0;i<200;i++) addk<<<grid,256>>>((float4*)a,(float4*)b,(float4*)o,n4); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1); printf("add n=%lld floats (%6.1f MB moved): %.2f us/launch  -> %.0f GB/s\n", n, n*12/1e6, ms*1000/200, n*12/(ms/200*1e-3)/1e9);
  }
  // dependent-load chain latency: 1 block of 64 threads, idx[i]=i (L2/HBM resident)
  int* idx; hipMalloc(&idx, 1<<24); std::vector<int> h(1<<22); for(int i=0;i<(1<<22);++i) h[i]=(i*9973+12345)&((1<<22)-1); hipMemcpy(idx,h.data(),1<<24,hipMemcpyHostToDevice);
  for (int hops : {1, 8, 32}) {
    for(int i=0;i<5;i++) chain<<<256,256>>>(idx,o,hops);
    hipEventRecord(e0); for(int i=0;i<50;i++) chain<<<256,256>>>(idx,o,hops); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1); printf("chain hops=%d: %.2f us/launch\n", hops, ms*1000/50);
  }
  return 0;
}
