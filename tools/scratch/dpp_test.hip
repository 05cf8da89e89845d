#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../nyxus_amd/csrc/device_math.h"
using namespace nyxhip;
__global__ void k(double* out, unsigned* o2){
  int lane = threadIdx.x;
  double v = (double)(lane*lane+1);
  out[lane] = wave_sum(v);
  o2[lane] = wave_max_u32((unsigned)((lane*37)%61));
  o2[64+lane] = lane_plus1(lane, 999);
  o2[128+lane] = lane_minus1(lane, 777);
  out[64+lane] = wave_max_nonneg((double)((lane*13)%50));
}
int main(){ double* d; unsigned* u; hipMalloc(&d, 128*8); hipMalloc(&u, 192*4); k<<<1,64>>>(d,u); double h[128]; unsigned hu[192]; hipMemcpy(h,d,sizeof(h),hipMemcpyDeviceToHost); hipMemcpy(hu,u,sizeof(hu),hipMemcpyDeviceToHost);
 double ref=0; for(int i=0;i<64;i++) ref += i*i+1; int ok=1; for(int i=0;i<64;i++) if(h[i]!=ref) ok=0; printf("sum ok=%d (%g vs %g)\n", ok, h[5], ref);
 unsigned m=0; for(int i=0;i<64;i++){unsigned x=(i*37)%61; if(x>m)m=x;} ok=1; for(int i=0;i<64;i++) if(hu[i]!=m) ok=0; printf("max ok=%d (%u vs %u)\n", ok, hu[3], m);
 ok=1; for(int i=0;i<64;i++){ unsigned e = i<63? i+1:999; if(hu[64+i]!=e) ok=0; unsigned e2 = i>0? i-1:777; if(hu[128+i]!=e2) ok=0;} printf("shift ok=%d (%u %u %u %u)\n", ok, hu[64], hu[64+63], hu[128], hu[128+63]);
 double dm=0; for(int i=0;i<64;i++){double x=(i*13)%50; if(x>dm)dm=x;} ok=1; for(int i=0;i<64;i++) if(h[64+i]!=dm) ok=0; printf("dmax ok=%d\n", ok);
 return 0;}
