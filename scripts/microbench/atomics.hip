// Microbenchmarks that size the batched SAGA design on gfx950:
//  (1) scattered fp64 global atomic adds into a table of T doubles (contention vs table size)
//  (2) random 8B / record gathers from a large array (latency-bound gather ceiling)
//  (3) kernel launch boundary cost for back-to-back tiny kernels (eager vs graph)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s failed %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__device__ inline unsigned hash32(unsigned x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

__global__ void k_atomic(double* D, unsigned table, int per_thread, int replicas){
  unsigned tid = blockIdx.x*blockDim.x+threadIdx.x;
  double* base = D + (size_t)(blockIdx.x % replicas) * table;
  for(int i=0;i<per_thread;++i){
    unsigned j = hash32(tid*977u + i*131071u) % table;
    __hip_atomic_fetch_add(base + j, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_atomic_ret(double* D, unsigned table, int per_thread, double* sink){
  unsigned tid = blockIdx.x*blockDim.x+threadIdx.x; double acc=0;
  for(int i=0;i<per_thread;++i){
    unsigned j = hash32(tid*977u + i*131071u) % table;
    acc += __hip_atomic_exchange(D + j, (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if(acc==12345.678) sink[0]=acc;
}
__global__ void k_gather(const double* A, size_t n, int per_thread, int rec_doubles, double* sink){
  unsigned tid = blockIdx.x*blockDim.x+threadIdx.x; double acc=0;
  int gl = threadIdx.x & 15; unsigned grp = tid>>4;
  for(int i=0;i<per_thread;++i){
    size_t r = ((size_t)hash32(grp*977u + i*131071u)*2654435761ull) % (n/rec_doubles);
    if(gl < rec_doubles) acc += A[r*rec_doubles + gl];
  }
  if(acc==12345.678) sink[0]=acc;
}
__global__ void k_tiny(double* p){ if(threadIdx.x==0 && blockIdx.x==0) p[0]+=1.0; }

int main(){
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double* D; size_t maxT = 1<<24; CK(hipMalloc(&D, maxT*8)); CK(hipMemset(D,0,maxT*8));
  double* sink; CK(hipMalloc(&sink,8));
  printf("== scattered fp64 atomic add: 1024 blocks x 256 thr x 16 each = 4.19M atomics ==\n");
  for(unsigned table : {125u*8, 1250u*8, 12500u*8, 125000u*8, 1u<<22}){
    for(int rep : {1, 8, 64}){
      if((size_t)table*rep > maxT) continue;
      k_atomic<<<1024,256,0,st>>>(D,table,16,rep); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0,st)); k_atomic<<<1024,256,0,st>>>(D,table,16,rep); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      printf("table %8u doubles (%7u lines) x %2d replicas: %8.3f ms  %7.2f G atomics/s\n", table, table/8, rep, ms, 4.194304/ms);
    }
  }
  printf("== small launches (like one 20K-draw batch: 200K atomics) ==\n");
  for(unsigned table : {1250u*8}){ for(int rep: {1,4,16,64}){
      int blocks = 200000/256/1; 
      k_atomic<<<blocks,256,0,st>>>(D,table,1,rep); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0,st)); for(int r=0;r<20;++r) k_atomic<<<blocks,256,0,st>>>(D,table,1,rep); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      printf("200K atomics/launch, table %u x %2d replicas: %7.2f us per launch\n", table, rep, ms*1000/20);
  }}
  printf("== returning atomic exchange on random 8B in 80MB (g_memory claim) ==\n");
  { size_t T=10000000; k_atomic_ret<<<1024,256,0,st>>>(D,T,4,sink); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0,st)); k_atomic_ret<<<1024,256,0,st>>>(D,T,4,sink); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("1.05M exch: %.3f ms %.2f G/s\n", ms, 1.048576/ms); }
  printf("== random record gather from 1 GiB (16 lanes per record) ==\n");
  { size_t n = (1ull<<30)/8; double* A; CK(hipMalloc(&A,n*8)); CK(hipMemset(A,0,n*8));
    for(int rec : {1,2,4,8,16}){
      k_gather<<<4096,256,0,st>>>(A,n,8,rec,sink); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0,st)); k_gather<<<4096,256,0,st>>>(A,n,8,rec,sink); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      double recs = 4096.0*256/16*8;
      printf("record %3d B: %.3f ms, %.2f G records/s, %.1f GB/s useful\n", rec*8, ms, recs/ms/1e6, recs*rec*8/ms/1e6);
    }
    printf("== small gather launches: 20K records of 128 B ==\n");
    { int blocks=20000*16/256; k_gather<<<blocks,256,0,st>>>(A,n,1,16,sink); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0,st)); for(int r=0;r<20;++r) k_gather<<<blocks,256,0,st>>>(A,n,1,16,sink); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("20K x 128B gather: %.2f us per launch (incl. boundary)\n", ms*1000/20); }
    CK(hipFree(A)); }
  printf("== launch boundary ==\n");
  { CK(hipEventRecord(e0,st)); for(int r=0;r<1000;++r) k_tiny<<<1,64,0,st>>>(sink); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("eager 1000 tiny kernels: %.2f us each\n", ms);
    hipGraph_t g; hipGraphExec_t ge; CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for(int r=0;r<1000;++r) k_tiny<<<1,64,0,st>>>(sink);
    CK(hipStreamEndCapture(st,&g)); CK(hipGraphInstantiate(&ge,g,nullptr,nullptr,0));
    CK(hipGraphLaunch(ge,st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0,st)); CK(hipGraphLaunch(ge,st)); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms,e0,e1)); printf("graph of 1000 tiny kernels: %.2f us each\n", ms); }
  return 0;
}
