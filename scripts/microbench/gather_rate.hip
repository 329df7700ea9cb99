// Random record gather ceilings on gfx950: how many random fixed-size records per second can
// the chip fetch as a function of footprint (L2 / Infinity Cache / HBM), record size and
// loads in flight per lane.  Sizes the packed-record design of saga_batched.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s failed %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__device__ inline unsigned hash32(unsigned x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

// 16 lanes per record, lane gl reads 8 B at offset 8*gl (+128 for the second line of a 256-B record)
template <int U, int LINES, bool RMW>
__global__ __launch_bounds__(1024) void k_rec(char* A, unsigned nrec, int stride, int iters, double* sink){
  const unsigned tid = blockIdx.x*blockDim.x+threadIdx.x;
  const int gl = threadIdx.x & 15; const unsigned grp = tid>>4;
  double acc=0;
  for(int it=0; it<iters; ++it){
    double v[U][LINES];
    char* base[U];
#pragma unroll
    for(int u=0;u<U;++u){
      unsigned r = hash32(grp*977u + (it*U+u)*131071u) % nrec;
      base[u] = A + (size_t)r*stride;
#pragma unroll
      for(int l=0;l<LINES;++l) v[u][l] = *reinterpret_cast<const double*>(base[u] + 128*l + 8*gl);
    }
#pragma unroll
    for(int u=0;u<U;++u){
#pragma unroll
      for(int l=0;l<LINES;++l) acc += v[u][l];
      if(RMW && gl==0) *reinterpret_cast<double*>(base[u] + 8) = acc;
    }
  }
  if(acc==12345.678) sink[0]=acc;
}

template <int U, int LINES, bool RMW>
void run(const char* name, char* A, size_t bytes, int stride, double* sink, hipStream_t st){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned nrec = (unsigned)(bytes/stride);
  const int blocks = 256*2, threads = 1024;           // 2 workgroups of 16 waves per CU
  const long groups = (long)blocks*threads/16;
  const int iters = (int)(16L*1024*1024/(groups*U));   // 16M records per launch
  k_rec<U,LINES,RMW><<<blocks,threads,0,st>>>(A,nrec,stride,iters,sink); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0,st)); k_rec<U,LINES,RMW><<<blocks,threads,0,st>>>(A,nrec,stride,iters,sink); CK(hipEventRecord(e1,st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  const double recs = (double)groups*U*iters;
  printf("%-34s footprint %8.1f MB stride %3d U=%d: %7.3f ms  %6.2f G rec/s  %6.2f G lines/s  %7.1f GB/s\n", name, bytes/1e6, stride, U, ms,
         recs/ms/1e6, recs*LINES/ms/1e6, recs*LINES*128/ms/1e6);
}

int main(){
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t maxb = 2560ull<<20;
  char* A; CK(hipMalloc(&A, maxb)); CK(hipMemset(A,0,maxb));
  double* sink; CK(hipMalloc(&sink,8));
  for(size_t mb : {16ul, 128ul, 2560ul}){
    const size_t b = mb<<20;
    run<1,1,false>("128B rec", A,b,128,sink,st);
    run<4,1,false>("128B rec", A,b,128,sink,st);
    run<8,1,false>("128B rec", A,b,128,sink,st);
    run<4,2,false>("256B rec (2 lines)", A,b,256,sink,st);
    run<8,2,false>("256B rec (2 lines)", A,b,256,sink,st);
    run<4,1,true >("128B rec + 8B store into it", A,b,128,sink,st);
    run<4,2,true >("256B rec + 8B store into it", A,b,256,sink,st);
  }
  return 0;
}
