#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out){
  __shared__ __attribute__((aligned(16))) short s[16][72];
  for(int i=threadIdx.x;i<16*72;i+=64) s[i/72][i%72]=(short)((i/72)*100+(i%72));
  __syncthreads();
  int l=threadIdx.x; int g=l>>4, i=l&15; int q=i>>2,p=i&3;
  typedef s16x4 __attribute__((address_space(3))) * lp;
  // every 16-lane group reads the same 4x16 block at rows 0..3 (+4*g to distinguish), cols 0..15
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(&s[4*g+q][4*p]));
  for(int e=0;e<4;e++) out[l*4+e]= v[e];
}
int main(){ short* d; hipMalloc(&d,64*4*2); k<<<1,64>>>(d); short h[256]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost);
 for(int l=0;l<64;l++){ printf("lane %2d:",l); for(int e=0;e<4;e++) printf(" %4d",h[l*4+e]); printf("\n"); } return 0; }
