// Which compute units does a bit of hipExtStreamCreateWithCUMask's mask select?  Launches workgroups that hold a
// whole CU's LDS on a masked stream and counts, per XCD (the XCC_ID hardware register), how many distinct
// (shader engine, CU) places they ran on.  Usage: cu_mask_probe <32-bit word, hex> (the word is repeated eight times).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ __launch_bounds__(64) void where_kernel(unsigned* out) {
  extern __shared__ unsigned hold[];
  hold[threadIdx.x] = threadIdx.x;
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);   // 20 us: every masked CU gets a workgroup
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const unsigned word = argc > 1 ? static_cast<unsigned>(std::strtoul(argv[1], nullptr, 16)) : 0x0f0f0f0fu;
  unsigned mask[8];
  for (unsigned& m : mask) m = word;
  hipStream_t s;
  CHECK(hipSetDevice(0));
  std::fprintf(stderr, "creating the stream\n");
  CHECK(hipExtStreamCreateWithCUMask(&s, 8, mask));
  std::fprintf(stderr, "stream created\n");
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(where_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int wgs = 2048;
  unsigned* out;
  CHECK(hipMalloc(&out, wgs * 2 * sizeof(unsigned)));
  CHECK(hipMemsetAsync(out, 0xff, wgs * 2 * sizeof(unsigned), s));
  hipLaunchKernelGGL(where_kernel, dim3(wgs), dim3(64), 100 * 1024, s, out);
  CHECK(hipGetLastError());
  CHECK(hipStreamSynchronize(s));
  std::fprintf(stderr, "kernel done\n");
  std::vector<unsigned> h(wgs * 2);
  CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
  std::set<unsigned> places[16];
  for (int b = 0; b < wgs; ++b) {
    const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    places[xcc].insert((hw >> 8) & 0xf | ((hw >> 13) & 0x7) << 4 | ((hw >> 12) & 1) << 7);   // CU_ID, SE_ID, SH_ID
  }
  std::printf("mask word %08x:", word);
  int total = 0;
  for (int x = 0; x < 8; ++x) {
    std::printf(" xcd%d=%zu", x, places[x].size());
    total += static_cast<int>(places[x].size());
  }
  std::printf(" total=%d\n", total);
  return 0;
}
