// Test helper, not part of the product library: a kernel that holds a chosen number of compute units for a chosen
// time, the stand-in for another tenant's kernel (a collective's, say) that is resident while the cooperative
// recurrent kernels of libvts_hip.so are launched.  Every workgroup takes 64 KiB of LDS, so a CU that runs one has
// no room for a 160 KiB workgroup until it ends, and spins on the constant 100 MHz clock.
#include <hip/hip_runtime.h>

extern "C" __global__ __launch_bounds__(256) void vts_test_occupy_kernel(long long ticks, unsigned* sink) {
  extern __shared__ unsigned hold[];
  hold[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {       // bounded: ends `ticks` after the workgroup started, whatever else runs
    acc += hold[(threadIdx.x + acc) & 255];
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0xffffffffu) sink[0] = acc;      // keeps the loop's loads alive
}

// workgroups: how many CUs to hold (one workgroup each while CUs are free); microseconds: for how long (at most 100 ms).
extern "C" int vts_test_occupy(int workgroups, long long microseconds, unsigned* sink, void* stream) {
  if (workgroups < 1 || workgroups > 256 || microseconds < 0 || microseconds > 100000 || sink == nullptr) return 1;
  hipLaunchKernelGGL(vts_test_occupy_kernel, dim3(workgroups), dim3(256), 64 * 1024, reinterpret_cast<hipStream_t>(stream),
                     microseconds * 100, sink);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
