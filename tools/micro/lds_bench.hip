// Micro-benchmark: LDS read rate of ds_read_b64 / ds_read_b128 (16-B aligned) / ds_read_b128 at an
// address = 8 (mod 16), all CUs busy, `waves` wavefronts per CU.  Prints bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_bench tools/micro/lds_bench.hip && ./lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, int stride_bytes, long long* cyc, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) ((int*)smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // consecutive lanes read consecutive 8- or 16-byte words (conflict-free), rows `stride_bytes` apart
  unsigned addr = (unsigned)(size_t)smem + (MODE == 0 ? lane * 8 : lane * 16) + (MODE == 2 ? 8 : 0);
  int acc = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      v2i r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(r[u]) : "v"(addr + u * stride_bytes));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += r[u].x;
    } else {
      v4i r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(r[u]) : "v"(addr + u * stride_bytes));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += r[u].x;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x7fffffff) sink[0] = acc;
}

int main() {
  long long* d;
  int* s;
  hipMalloc(&d, 4096 * 8);
  hipMalloc(&s, 4);
  const int iters = 4000;
  for (int waves : {4, 8, 16}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int stride : {8 * 1031, 8 * 1032}) {  // odd / even row stride in elements
        if (mode == 2 && false) continue;
        const int blocks = 256, threads = waves * 64;
        auto fn = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 160 * 1024, 0, iters, stride, d, s);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 160 * 1024, 0, iters, stride, d, s);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += v;
        mean /= blocks;
        const double bytes = (double)iters * 8 * waves * 64 * (mode == 0 ? 8 : 16);
        // s_memtime counts at a fixed 100 MHz-class clock on gfx9? report raw ticks and derived figure
        printf("waves=%2d mode=%s stride=%5d ticks=%.0f bytes/tick/CU=%.1f\n", waves,
               mode == 0 ? "b64      " : mode == 1 ? "b128     " : "b128+8   ", stride / 8, mean, bytes / mean);
      }
    }
  }
  return 0;
}
