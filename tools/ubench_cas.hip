// Calibration (not product code): random 64-bit CAS inserts vs table size and XCD locality on gfx950.
// Answers: where is the knee (L2 4 MB/XCD, Infinity Cache 256 MB, HBM) for the DISTINCT set build?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__device__ inline uint64_t mix(uint64_t x){uint64_t z=x+0x9E3779B97F4A7C15ull; z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}

// mode 0: every workgroup hits the whole table.  mode 1: workgroups of XCD x (blockIdx%8) hit sub-table x only.
// `phases`: the table is covered in `phases` consecutive windows (all workgroups work on window p at the same time,
// roughly): window size = slots/phases.
__global__ __launch_bounds__(512) void k_cas(unsigned long long* tab, uint64_t slots, uint64_t n, int mode, uint32_t phases, unsigned long long* fresh){
  uint64_t per = n / gridDim.x;
  uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per;
  uint64_t wslots = slots / phases;
  unsigned long long f = 0;
  uint64_t per_phase = per / phases;
  for(uint32_t p=0;p<phases;p++){
    uint64_t base = (uint64_t)p * wslots;
    uint64_t sub = wslots, off = 0;
    if(mode==1){ sub = wslots/8; off = (blockIdx.x & 7) * sub; }
    for(uint64_t i=lo + p*per_phase + threadIdx.x; i<lo+(p+1)*per_phase; i+=blockDim.x){
      uint64_t v = mix(i) | 1ull;
      uint64_t h = base + off + (mix(v ^ 0x1234567ull) & (sub-1));
      for(int probe=0; probe<64; probe++){
        if(mode==2){ unsigned long long old = atomicCAS(&tab[h], 0ull, (unsigned long long)v); if(old==0ull){f++; break;} if(old==v) break; }
        else {
        unsigned long long cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(cur == v) break;
        if(cur == 0ull){ unsigned long long old = atomicCAS(&tab[h], 0ull, (unsigned long long)v); if(old==0ull){f++; break;} if(old==v) break; }
        }
        h = base + off + ((h - base - off + 1) & (sub-1));
      }
    }
  }
  for(int o=32;o>0;o>>=1) f+=__shfl_down(f,o,64);
  if((threadIdx.x&63)==0 && f) atomicAdd(fresh,f);
}
int main(int argc,char**argv){
  uint64_t n = 100000000ull;
  unsigned long long* d_fresh; CK(hipMalloc(&d_fresh,8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // (slots, phases, mode): total inserts always n; each window sees n/phases distinct values in wslots slots
  struct Cfg{uint64_t slots; uint32_t phases; int mode;} cfgs[] = {
    {1ull<<28,1,0},      // 2 GB table, one window (today's situation), load 0.37
    {1ull<<28,1,2},      // same, CAS without the load in front
    {1ull<<28,16,2},
    {1ull<<28,16,0},     // 16 windows of 128 MB
    {1ull<<28,64,1},     // 64 windows of 32 MB, each XCD its own 4 MB eighth
    {1ull<<28,512,1},    // 512 windows of 4 MB, each XCD 512 KB
  };
  unsigned long long* tab; CK(hipMalloc(&tab,(1ull<<28)*8));
  for(auto c: cfgs){
    CK(hipMemset(tab,0,c.slots*8)); CK(hipMemset(d_fresh,0,8)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_cas,dim3(2048),dim3(512),0,0,tab,c.slots,n,c.mode,c.phases,d_fresh);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    unsigned long long f; CK(hipMemcpy(&f,d_fresh,8,hipMemcpyDeviceToHost));
    printf("slots=2^28 phases=%4u window=%6.1f MB mode=%d : %7.3f ms  %6.1f G inserts/s fresh=%llu\n",c.phases,(double)c.slots*8/c.phases/1e6,c.mode,ms,n/ms/1e6,f);
  }
  return 0;
}
