// Where do the workgroups of a CU-masked stream land on gfx950?  (hipExtStreamCreateWithCUMask)
//   side stream: the first R mask bits; main stream: the other 256 - R.  Every workgroup records the XCC, shader
//   engine and CU it ran on; the main kernel takes the whole LDS of a CU like the LDS gather, so a second round
//   (two workgroups on one CU) shows up in its duration.
//   hipcc -O3 --offload-arch=gfx950 cu_mask.hip -o cu_mask && ./cu_mask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s failed %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void where(unsigned* out, int spin_us) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc & 0xf;
    out[2 * blockIdx.x + 1] = hw;
    lds[0] = (char)xcc;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (long long)spin_us * 100) {}   // 100 MHz counter
}

static std::set<unsigned> g_last;
static void report(const char* name, const std::vector<unsigned>& h, int blocks) {
  std::set<unsigned> places;
  int per_xcc[16] = {0};
  for (int b = 0; b < blocks; ++b) {
    const unsigned xcc = h[2 * b], hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    places.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    per_xcc[xcc]++;
  }
  printf("%s: %d workgroups on %zu distinct CUs; per XCC:", name, blocks, places.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  if (places.size() <= 16) {
    printf("    CUs (xcc.se.sh.cu):");
    for (unsigned q : places) printf(" %u.%u.%u.%u", q >> 16, (q >> 8) & 0xff, (q >> 4) & 0xf, q & 0xf);
    printf("\n");
  }
  g_last = places;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 8;
  const int first = argc > 2 ? atoi(argv[2]) : 0;       // first mask bit of the side stream
  unsigned side_mask[8] = {0}, main_mask[8];
  for (int i = first; i < first + R; ++i) side_mask[i >> 5] |= 1u << (i & 31);
  for (int i = 0; i < 8; ++i) main_mask[i] = ~side_mask[i];
  hipStream_t side, mainst, plain;
  CK(hipExtStreamCreateWithCUMask(&side, 8, side_mask));
  CK(hipExtStreamCreateWithCUMask(&mainst, 8, main_mask));
  CK(hipStreamCreate(&plain));
  unsigned* d;
  CK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned)));
  std::vector<unsigned> h(2 * 4096);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Case { const char* name; hipStream_t st; int blocks; size_t lds; } cases[] = {
      {"side stream, small workgroups", side, 64, 0},
      {"main stream, whole-LDS workgroups", mainst, 256 - R, 160 * 1024 - 64},
      {"main stream, 256 whole-LDS workgroups (one too many CUs)", mainst, 256, 160 * 1024 - 64},
      {"unmasked stream, 256 whole-LDS workgroups", plain, 256, 160 * 1024 - 64}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, c.st));
      where<<<c.blocks, 1024, c.lds, c.st>>>(d, 20);
      CK(hipEventRecord(e1, c.st));
      CK(hipStreamSynchronize(c.st));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, 2 * c.blocks * sizeof(unsigned), hipMemcpyDeviceToHost));
    report(c.name, h, c.blocks);
    printf("    kernel %.1f us (a 20 us spin per workgroup)\n", ms * 1e3);
  }
  // both at once: the side kernel must not delay the main kernel
  CK(hipEventRecord(e0, mainst));
  where<<<64, 256, 0, side>>>(d + 2 * 1024, 200);
  where<<<256 - R, 1024, 160 * 1024 - 64, mainst>>>(d, 20);
  CK(hipEventRecord(e1, mainst));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("main kernel beside a 200 us side kernel: %.1f us\n", ms * 1e3);
  return 0;
}
