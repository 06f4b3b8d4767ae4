#include "../m2trans_amd/csrc/m2t_common.h"
#include <cstdio>
int m2t_set_hip_error(hipError_t e, const char* file, int line){return 1;}
int m2t_set_error(int code, const char* msg){return code;}
__global__ void k(float* out, float* out2){
  __shared__ __attribute__((aligned(16))) bf16_t s[128][72];
  for(int i=threadIdx.x;i<128*72;i+=64) s[i/72][i%72]=(bf16_t)(float)((i/72) + 0.0f);
  __syncthreads();
  int lane=threadIdx.x, g=lane>>4;
  int c4=1, mt=2;
  Frag8<bf16_t> f = load8_tr(&s[32*c4+4*g][16*mt], &s[32*c4+16+4*g][16*mt], 72, lane);
  for(int e=0;e<8;e++) out[lane*8+e]=f.get(e);
  for(int e=0;e<4;e++) out2[lane*4+e]=(float)s[32*c4+4*g+e][16*mt+(lane&15)];
}
int main(){ float* d,*d2; (void)hipMalloc(&d,64*8*4);(void)hipMalloc(&d2,64*4*4); k<<<1,64>>>(d,d2); float h[512],h2[256]; (void)hipMemcpy(h,d,2048,hipMemcpyDeviceToHost);(void)hipMemcpy(h2,d2,1024,hipMemcpyDeviceToHost);
 for(int l=0;l<64;l+=7){ printf("lane %2d (g=%d): tr",l,l>>4); for(int e=0;e<8;e++) printf(" %4.0f",h[l*8+e]); printf(" | direct"); for(int e=0;e<4;e++) printf(" %4.0f",h2[l*4+e]); printf("\n"); } return 0; }
