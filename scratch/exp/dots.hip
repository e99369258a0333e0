#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__device__ inline double wave_sum(double v){ for(int o=32;o>0;o>>=1) v+=__shfl_down(v,o,64); return v; }
// variant A: current structure (waves split columns), column stride ld
template<int ROWS>
__global__ __launch_bounds__(256) void dotsA(const double* __restrict__ B, const double* __restrict__ r, const double* __restrict__ z, double* __restrict__ partial, long d, long ld, int m){
  const int lane=threadIdx.x&63, w=threadIdx.x>>6; const long r0=(long)blockIdx.x*ROWS; constexpr int kIter=ROWS/128;
  double2 rz[kIter];
  for(int it=0;it<kIter;++it){ long i=r0+it*128+2*lane; double2 zz=*(const double2*)(z+i), rr=*(const double2*)(r+i); rz[it]=make_double2(zz.x*rr.x, zz.y*rr.y);}
  for(int q0=0; w+4*q0<m; q0+=8){ double acc[8]; for(int q=0;q<8;++q) acc[q]=0;
    #pragma unroll
    for(int it=0;it<kIter;++it){ long i=r0+it*128+2*lane;
      #pragma unroll
      for(int q=0;q<8;++q){ int j=w+4*(q0+q); int jc = j<m? j : m-1; double2 b=*(const double2*)(B+(long)jc*ld+i); acc[q]=fma(b.x,rz[it].x,fma(b.y,rz[it].y,acc[q])); } }
    for(int q=0;q<8;++q){ double v=wave_sum(acc[q]); int j=w+4*(q0+q); if(lane==0&&j<m) partial[(long)blockIdx.x*256+j]=v; } }
}
// variant C: pure streaming read (sum everything) to find the ceiling
__global__ __launch_bounds__(256) void stream_sum(const double* __restrict__ B, double* __restrict__ out, long n){
  double2 a=make_double2(0,0);
  for(long i=((long)blockIdx.x*256+threadIdx.x)*2; i<n; i+=(long)gridDim.x*512){ double2 b=*(const double2*)(B+i); a.x+=b.x; a.y+=b.y; }
  double v=wave_sum(a.x+a.y); if((threadIdx.x&63)==0) out[blockIdx.x*4+(threadIdx.x>>6)]=v;
}
// variant B: blockIdx.y = group of 8 columns; all 256 threads along rows; ITER row-pairs per thread
template<int ITER>
__global__ __launch_bounds__(256) void dotsB(const double* __restrict__ B, const double* __restrict__ r, const double* __restrict__ z, double* __restrict__ partial, long d, long ld, int m){
  __shared__ double red[4][8];
  const int lane=threadIdx.x&63, w=threadIdx.x>>6; const long r0=(long)blockIdx.x*(512*ITER); const int j0=blockIdx.y*8;
  double acc[8]; for(int q=0;q<8;++q) acc[q]=0;
  #pragma unroll
  for(int it=0;it<ITER;++it){ long i=r0+it*512+2*threadIdx.x; double2 zz=*(const double2*)(z+i), rr=*(const double2*)(r+i); double2 rz=make_double2(zz.x*rr.x, zz.y*rr.y);
    #pragma unroll
    for(int q=0;q<8;++q){ int j=j0+q; if(j<m){ double2 b=*(const double2*)(B+(long)j*ld+i); acc[q]=fma(b.x,rz.x,fma(b.y,rz.y,acc[q])); } } }
  for(int q=0;q<8;++q){ double v=wave_sum(acc[q]); if(lane==0) red[w][q]=v; }
  __syncthreads();
  if(threadIdx.x<8 && j0+threadIdx.x<m) partial[(long)blockIdx.x*256+j0+threadIdx.x]=red[0][threadIdx.x]+red[1][threadIdx.x]+red[2][threadIdx.x]+red[3][threadIdx.x];
}
int main(){ long d=196608; int m=32; 
  for(long pad : {0L}){ long ld=d+pad; double *B,*r,*z,*p; CK(hipMalloc(&B,sizeof(double)*ld*64)); CK(hipMalloc(&r,8*d)); CK(hipMalloc(&z,8*d)); CK(hipMalloc(&p,8*2048*256));
    CK(hipMemset(B,0,sizeof(double)*ld*64)); CK(hipMemset(r,0,8*d)); CK(hipMemset(z,0,8*d));
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for(int rows : {128, 256, 384, 768}){
      auto run=[&](int it){ for(int i=0;i<it;++i){ if(rows==768) hipLaunchKernelGGL(dotsA<768>, dim3(d/768), dim3(256),0,0,B,r,z,p,d,ld,m); else if(rows==384) hipLaunchKernelGGL(dotsA<384>, dim3(d/384), dim3(256),0,0,B,r,z,p,d,ld,m); else if(rows==256) hipLaunchKernelGGL(dotsA<256>, dim3(d/256), dim3(256),0,0,B,r,z,p,d,ld,m); else hipLaunchKernelGGL(dotsA<128>, dim3(d/128), dim3(256),0,0,B,r,z,p,d,ld,m);} };
      run(5); hipDeviceSynchronize(); hipEventRecord(e0); run(50); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
      printf("pad %5ld rows %4d: %.2f us  %.2f TB/s\n", pad, rows, ms*1e3/50, 8.0*d*m/(ms/50*1e-3)/1e12);
    }
    for(int iter : {1,2,4}){ auto runB=[&](int n){ for(int i=0;i<n;++i){ if(iter==1) hipLaunchKernelGGL(dotsB<1>, dim3(d/512, (m+7)/8), dim3(256),0,0,B,r,z,p,d,ld,m); else if(iter==2) hipLaunchKernelGGL(dotsB<2>, dim3(d/1024,(m+7)/8), dim3(256),0,0,B,r,z,p,d,ld,m); else hipLaunchKernelGGL(dotsB<4>, dim3(d/2048,(m+7)/8), dim3(256),0,0,B,r,z,p,d,ld,m);} };
      runB(5); hipDeviceSynchronize(); hipEventRecord(e0); runB(50); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
      printf("B iter %d: %.2f us  %.2f TB/s\n", iter, ms*1e3/50, 8.0*d*m/(ms/50*1e-3)/1e12); }
    hipLaunchKernelGGL(stream_sum, dim3(2048), dim3(256),0,0,B,p,(long)ld*m); hipDeviceSynchronize();
    hipEventRecord(e0); for(int i=0;i<50;++i) hipLaunchKernelGGL(stream_sum, dim3(2048), dim3(256),0,0,B,p,(long)ld*m); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
    printf("pad %5ld stream: %.2f us %.2f TB/s\n", pad, ms*1e3/50, 8.0*ld*m/(ms/50*1e-3)/1e12);
    hipFree(B);hipFree(r);hipFree(z);hipFree(p);
  }
  return 0; }
