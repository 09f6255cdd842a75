// Calibration (not product code): LDS atomic rates on gfx950 — what one CU sustains for the accumulator updates of the
// workgroup tables.  Every thread does N atomics on a table of S slots in LDS, slot = random / same-per-wave / lane-linear;
// kinds: u32 add, u64 add, f64 add, u64 add with return, u64 CAS, plain u64 store, plain u64 load.
// Output: lane-atomics per clock per CU (2.4 GHz nominal) for 1/2/3 workgroups of 512 threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) double lds_f64;
__device__ inline uint32_t mix(uint32_t x){x^=x>>16; x*=0x7feb352du; x^=x>>15; x*=0x846ca68bu; x^=x>>16; return x;}

template <int KIND, int PAT>
__global__ __launch_bounds__(512) void k(uint32_t S, uint32_t iters, unsigned long long* sink){
  extern __shared__ uint64_t lds[];
  for(uint32_t i=threadIdx.x;i<S;i+=512) lds[i]=0;
  __syncthreads();
  uint32_t r = mix(blockIdx.x*512u+threadIdx.x+1u);
  unsigned long long acc=0;
  for(uint32_t it=0; it<iters; it++){
    r = r*1664525u+1013904223u;
    uint32_t slot;
    if(PAT==0) slot = (uint32_t)(((uint64_t)mix(r)*S)>>32);              // random per lane
    else if(PAT==1) slot = (threadIdx.x + it*64u) % S;                      // lane-linear (conflict free)
    else slot = (uint32_t)(((uint64_t)mix((threadIdx.x>>6)+it*977u)*S)>>32); // same slot for the whole wave
    lds_u64* p = (lds_u64*)lds + slot;
    if(KIND==0) (void)__hip_atomic_fetch_add((lds_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if(KIND==1) (void)__hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if(KIND==2) (void)__hip_atomic_fetch_add((lds_f64*)p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if(KIND==3) acc += __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if(KIND==4){ unsigned long long e=0; __hip_atomic_compare_exchange_strong(p,&e,(unsigned long long)r,__ATOMIC_RELAXED,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); acc+=e; }
    else if(KIND==5) *(volatile lds_u64*)p = r;
    else if(KIND==6) acc += *(volatile lds_u64*)p;
    else if(KIND==7) acc += __hip_atomic_fetch_add((lds_u32*)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if(acc==0x123456789ull || lds[threadIdx.x % S]==0x987654321ull) *sink = acc;
}
template <int KIND, int PAT>
int run(const char* name, uint32_t S, int wg_per_cu, unsigned long long* sink){
  const uint32_t iters=4096;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t shmem = (size_t)S*8;
  if(shmem > 48*1024) (void)hipFuncSetAttribute((const void*)k<KIND,PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((k<KIND,PAT>),dim3(256*wg_per_cu),dim3(512),shmem,0,S,64u,sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<KIND,PAT>),dim3(256*wg_per_cu),dim3(512),shmem,0,S,iters,sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double per_cu = (double)wg_per_cu*512*iters/(ms*1e-3);  // lane-atomics per second per CU
  printf("%-22s S=%5u wg/CU=%d : %7.3f ms  %6.2f G lane-ops/s/CU = %5.2f per clock @2.4GHz\n",name,S,wg_per_cu,ms,per_cu/1e9,per_cu/2.4e9);
  return 0;
}
int main(){
  unsigned long long* sink; CK(hipMalloc(&sink,8));
  const uint32_t S=1024;
  for(int w=1;w<=3;w+=2){
    run<0,0>("u32 add random",S,w,sink); run<1,0>("u64 add random",S,w,sink); run<2,0>("f64 add random",S,w,sink);
    run<3,0>("u64 add ret random",S,w,sink); run<7,0>("u32 add ret random",S,w,sink); run<4,0>("u64 cas random",S,w,sink);
    run<5,0>("u64 store random",S,w,sink); run<6,0>("u64 load random",S,w,sink);
    run<0,1>("u32 add linear",S,w,sink); run<1,1>("u64 add linear",S,w,sink); run<2,1>("f64 add linear",S,w,sink);
    run<1,2>("u64 add same/wave",S,w,sink); run<0,2>("u32 add same/wave",S,w,sink);
  }
  run<1,0>("u64 add random",8192,2,sink); run<0,0>("u32 add random",8192,2,sink); run<0,0>("u32 add random",256,2,sink);
  return 0;
}
