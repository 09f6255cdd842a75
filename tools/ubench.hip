// Calibration micro-benchmarks (not product code): how fast can config 2 run when fully specialised?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__device__ inline uint64_t splitmix64(uint64_t x){uint64_t z=x+0x9E3779B97F4A7C15ull; z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
__global__ void gen(uint32_t* cat, uint8_t* tag, uint64_t* pay, uint64_t n, uint32_t k){
  uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  uint64_t r=splitmix64(i*8); cat[i]=(uint32_t)__umul64hi(r,k);
  uint64_t sel=splitmix64(i*8+1)%1000, r2=splitmix64(i*8+2);
  if(sel<800){uint64_t c=r2%10000; if(c%100==0){tag[i]=4;pay[i]=c/100;} else {tag[i]=5; double d=(double)c/100.0; pay[i]=__double_as_longlong(d);}}
  else if(sel<980){tag[i]=4;pay[i]=r2%101;} else if(sel<990){tag[i]=1;pay[i]=0;} else if(sel<995){tag[i]=0;pay[i]=0;} else {tag[i]=6;pay[i]=k;}
}

// (1) pure streaming read of the three arrays, 32-bit indexing, R rows per lane
template<int R> __global__ __launch_bounds__(1024) void k_read(const uint32_t* __restrict__ cat,const uint8_t* __restrict__ tag,const uint64_t* __restrict__ pay,uint32_t n,unsigned long long* out){
  unsigned long long acc=0; uint32_t tile=1024*R;
  for(uint32_t base=blockIdx.x*tile; base<n; base+=gridDim.x*tile){
    #pragma unroll
    for(int j=0;j<R;j++){uint32_t i=base+j*1024+threadIdx.x; if(i<n){acc+=cat[i]+tag[i]+pay[i];}}
  }
  for(int o=32;o>0;o>>=1) acc+=__shfl_down(acc,o,64);
  if((threadIdx.x&63)==0) atomicAdd(out,acc);
}
// (2) + predicate price>50 with N1QL tag rules, count survivors
template<int R> __global__ __launch_bounds__(1024) void k_pred(const uint32_t* __restrict__ cat,const uint8_t* __restrict__ tag,const uint64_t* __restrict__ pay,uint32_t n,unsigned long long* out){
  unsigned long long acc=0; uint32_t tile=1024*R;
  for(uint32_t base=blockIdx.x*tile; base<n; base+=gridDim.x*tile){
    #pragma unroll
    for(int j=0;j<R;j++){uint32_t i=base+j*1024+threadIdx.x; if(i<n){uint32_t t=tag[i]; uint64_t p=pay[i];
      bool pass = t==4 ? (int64_t)p>50 : (t==5 ? __longlong_as_double(p)>50.0 : t>=6);
      acc+=pass? cat[i]:0;}}
  }
  for(int o=32;o>0;o>>=1) acc+=__shfl_down(acc,o,64);
  if((threadIdx.x&63)==0) atomicAdd(out,acc);
}
// (3) full config 2: direct LDS table [isum][fsum][flags] word-major, global merge by atomics
template<int R> __global__ __launch_bounds__(1024) void k_full(const uint32_t* __restrict__ cat,const uint8_t* __restrict__ tag,const uint64_t* __restrict__ pay,uint32_t n,uint32_t S,unsigned long long* gtab){
  extern __shared__ unsigned long long lds[];
  for(uint32_t s=threadIdx.x;s<3*S;s+=1024) lds[s]=0;
  __syncthreads();
  uint32_t tile=1024*R;
  for(uint32_t base=blockIdx.x*tile; base<n; base+=gridDim.x*tile){
    uint32_t t[R]; uint64_t p[R]; uint32_t c[R]; bool v[R];
    #pragma unroll
    for(int j=0;j<R;j++){uint32_t i=base+j*1024+threadIdx.x; v[j]=i<n; t[j]=v[j]?tag[i]:0; p[j]=v[j]?pay[i]:0; c[j]=v[j]?cat[i]:0;}
    #pragma unroll
    for(int j=0;j<R;j++){
      bool pass = t[j]==4 ? (int64_t)p[j]>50 : (t[j]==5 ? __longlong_as_double(p[j])>50.0 : t[j]>=6);
      if(pass && v[j]){ uint32_t s=c[j]+2;
        if(t[j]==4){ atomicAdd(&lds[s],(unsigned long long)p[j]); if(!(lds[2*S+s]&1)) atomicOr(&lds[2*S+s],1ull);}
        else if(t[j]==5){ atomicAdd((double*)&lds[S+s],__longlong_as_double(p[j])); if(!(lds[2*S+s]&4)) atomicOr(&lds[2*S+s],4ull);}
      }
    }
  }
  __syncthreads();
  for(uint32_t s=threadIdx.x;s<S;s+=1024){ unsigned long long fl=lds[2*S+s]; if(!fl) continue;
    if(fl&1) atomicAdd(&gtab[s*4],lds[s]); if(fl&4) atomicAdd((double*)&gtab[s*4+1],__longlong_as_double(lds[S+s])); atomicOr(&gtab[s*4+2],fl);}
}
// (4) same as (3) but 2 adjacent rows per 16-byte load
__global__ __launch_bounds__(1024) void k_full_wide(const uint2* __restrict__ cat2,const uint16_t* __restrict__ tag2,const ulonglong2* __restrict__ pay2,uint32_t npairs,uint32_t S,unsigned long long* gtab){
  extern __shared__ unsigned long long lds[];
  for(uint32_t s=threadIdx.x;s<3*S;s+=1024) lds[s]=0;
  __syncthreads();
  constexpr int R=2; uint32_t tile=1024*R;
  for(uint32_t base=blockIdx.x*tile; base<npairs; base+=gridDim.x*tile){
    uint16_t tt[R]; ulonglong2 pp[R]; uint2 cc[R]; bool v[R];
    #pragma unroll
    for(int j=0;j<R;j++){uint32_t i=base+j*1024+threadIdx.x; v[j]=i<npairs; if(v[j]){tt[j]=tag2[i]; pp[j]=pay2[i]; cc[j]=cat2[i];} else {tt[j]=0; pp[j]=make_ulonglong2(0,0); cc[j]=make_uint2(0,0);} }
    #pragma unroll
    for(int j=0;j<R;j++){
      #pragma unroll
      for(int h=0;h<2;h++){ uint32_t t= h? (tt[j]>>8):(tt[j]&255); uint64_t p= h? pp[j].y:pp[j].x; uint32_t c= h? cc[j].y:cc[j].x;
        bool pass = t==4 ? (int64_t)p>50 : (t==5 ? __longlong_as_double(p)>50.0 : t>=6);
        if(pass && v[j]){ uint32_t s=c+2;
          if(t==4){ atomicAdd(&lds[s],(unsigned long long)p); if(!(lds[2*S+s]&1)) atomicOr(&lds[2*S+s],1ull);}
          else if(t==5){ atomicAdd((double*)&lds[S+s],__longlong_as_double(p)); if(!(lds[2*S+s]&4)) atomicOr(&lds[2*S+s],4ull);}
        }
      }
    }
  }
  __syncthreads();
  for(uint32_t s=threadIdx.x;s<S;s+=1024){ unsigned long long fl=lds[2*S+s]; if(!fl) continue;
    if(fl&1) atomicAdd(&gtab[s*4],lds[s]); if(fl&4) atomicAdd((double*)&gtab[s*4+1],__longlong_as_double(lds[S+s])); atomicOr(&gtab[s*4+2],fl);}
}

int main(int argc,char**argv){
  uint64_t n= argc>1? strtoull(argv[1],0,10):100000000ull; uint32_t k= argc>2? atoi(argv[2]):1000;
  uint32_t *cat; uint8_t* tag; uint64_t* pay; unsigned long long *out,*gtab;
  CK(hipMalloc(&cat,n*4)); CK(hipMalloc(&tag,n)); CK(hipMalloc(&pay,n*8)); CK(hipMalloc(&out,8)); CK(hipMalloc(&gtab,(k+2)*32));
  gen<<<(n+255)/256,256>>>(cat,tag,pay,n,k); CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t S=k+2;
  auto timeit=[&](const char* name, auto launch, double bytes){ float best=1e9; for(int r=0;r<6;r++){ hipMemset(out,0,8); hipMemset(gtab,0,(k+2)*32); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best) best=ms;} printf("%-28s %8.3f ms  %7.0f GB/s\n",name,best,bytes/best/1e6); fflush(stdout);};
  for(int grid: {256,512,1024,2048}){
    printf("grid %d\n",grid);
    timeit("read R4", [&]{k_read<4><<<grid,1024>>>(cat,tag,pay,(uint32_t)n,out);}, 13.0*n);
    timeit("read R8", [&]{k_read<8><<<grid,1024>>>(cat,tag,pay,(uint32_t)n,out);}, 13.0*n);
    timeit("pred R4", [&]{k_pred<4><<<grid,1024>>>(cat,tag,pay,(uint32_t)n,out);}, 13.0*n);
    timeit("full R4", [&]{k_full<4><<<grid,1024,3*S*8>>>(cat,tag,pay,(uint32_t)n,S,gtab);}, 13.0*n);
    timeit("full R8", [&]{k_full<8><<<grid,1024,3*S*8>>>(cat,tag,pay,(uint32_t)n,S,gtab);}, 13.0*n);
    timeit("full wide(2x16B)", [&]{k_full_wide<<<grid,1024,3*S*8>>>((const uint2*)cat,(const uint16_t*)tag,(const ulonglong2*)pay,(uint32_t)(n/2),S,gtab);}, 13.0*n);
  }
  std::vector<unsigned long long> h((k+2)*4); hipMemcpy(h.data(),gtab,(k+2)*32,hipMemcpyDeviceToHost);
  printf("check slot2: isum=%llu flags=%llu\n",h[2*4],h[2*4+2]);
  return 0;
}
