#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float* o){ 
  float xs[4]={-3.f,-2.f,-1.f,0.f};
  for(int i=0;i<4;i++) o[i]=expf(xs[i]);
  float s=((o[0]+o[1])+o[2])+o[3]; o[4]=s;
  for(int i=0;i<4;i++) o[5+i]=o[i]/s;
  volatile float one=1.0f, d=0x1.8d9186p+0f; o[9]=one/d;
  o[10]=__expf(-3.f);
}
int main(){ float* d; hipMalloc(&d,64*4); k<<<1,1>>>(d); float h[16]; hipMemcpy(h,d,16*4,hipMemcpyDeviceToHost);
 for(int i=0;i<11;i++) printf("%d %a | host expf %a\n",i,h[i], i<4? expf(-3.f+i):0.f); return 0;}
