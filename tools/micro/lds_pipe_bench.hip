// How much LDS read bandwidth do the load/wait/add patterns of the fold reach at 8 waves per SIMD?
//   S: 16 ds_read_b64 per iteration, no wait inside the loop (streaming upper bound)
//   A: 16 ds_read_b64, s_waitcnt lgkmcnt(0), 16 v_add_f64            (the fold's group: drain, then consume)
//   B: two sets of 8 registers, never drained: issue set1, wait lgkmcnt(8), consume set0, issue set0, wait(8), consume set1
// Reports TB/s chip-wide from hipEvent time (bytes = loads x 512 B).
//   hipcc --offload-arch=gfx950 -O3 lds_pipe_bench.hip -o lds_pipe_bench && ./lds_pipe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v1d;
#define RD(dst, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, int rowbytes, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((double*)smem)[i] = 1e-9 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned addr = (unsigned)(size_t)smem + lane * 8 + (threadIdx.x >> 6) * 64;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  double a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
  if (MODE == 2) { RD(a0, 0); RD(a1, 512); RD(a2, 1024); RD(a3, 1536); RD(a4, 2048); RD(a5, 2560); RD(a6, 3072); RD(a7, 3584); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      RD(a0, 0); RD(a1, 512); RD(a2, 1024); RD(a3, 1536); RD(a4, 2048); RD(a5, 2560); RD(a6, 3072); RD(a7, 3584);
      RD(b0, 4096); RD(b1, 4608); RD(b2, 5120); RD(b3, 5632); RD(b4, 6144); RD(b5, 6656); RD(b6, 7168); RD(b7, 7680);
    } else if (MODE == 1) {
      RD(a0, 0); RD(a1, 512); RD(a2, 1024); RD(a3, 1536); RD(a4, 2048); RD(a5, 2560); RD(a6, 3072); RD(a7, 3584);
      RD(b0, 4096); RD(b1, 4608); RD(b2, 5120); RD(b3, 5632); RD(b4, 6144); RD(b5, 6656); RD(b6, 7168); RD(b7, 7680);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc0 += a0; acc1 += a1; acc2 += a2; acc3 += a3; acc0 += a4; acc1 += a5; acc2 += a6; acc3 += a7;
      acc0 += b0; acc1 += b1; acc2 += b2; acc3 += b3; acc0 += b4; acc1 += b5; acc2 += b6; acc3 += b7;
    } else {
      RD(b0, 4096); RD(b1, 4608); RD(b2, 5120); RD(b3, 5632); RD(b4, 6144); RD(b5, 6656); RD(b6, 7168); RD(b7, 7680);
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)::"memory");
      acc0 += a0; acc1 += a1; acc2 += a2; acc3 += a3; acc0 += a4; acc1 += a5; acc2 += a6; acc3 += a7;
      RD(a0, 0); RD(a1, 512); RD(a2, 1024); RD(a3, 1536); RD(a4, 2048); RD(a5, 2560); RD(a6, 3072); RD(a7, 3584);
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)::"memory");
      acc0 += b0; acc1 += b1; acc2 += b2; acc3 += b3; acc0 += b4; acc1 += b5; acc2 += b6; acc3 += b7;
    }
    addr ^= (it & 1) ? 8u : 0u;  // keep the address a run-time value
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (MODE == 0) { acc0 = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7; }
  if (MODE == 2) { acc0 += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
int main(int argc, char** argv) {
  // argv: threads per workgroup, LDS bytes per workgroup (occupancy), iterations: `1024 163840` = one 16-wave workgroup per CU (4 waves per SIMD)
  const int threads = argc > 1 ? atoi(argv[1]) : 512, lds = argc > 2 ? atoi(argv[2]) : 16384, iters = argc > 3 ? atoi(argv[3]) : 5000;
  const int blocks = 1024;
  setvbuf(stdout, NULL, _IONBF, 0);
  double* o; hipMalloc(&o, (size_t)blocks * threads * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"S stream, no wait      ", "A 16 loads/drain/16 adds", "B 8+8 never drained     "};
  for (int mode = 0; mode < 3; ++mode) {
    auto fn = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, 0, iters, 4096, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double bytes = (double)blocks * (threads / 64) * iters * 16 * 512;
    printf("threads %d lds %d | %s: %.3f ms  %.1f TB/s\n", threads, lds, names[mode], best, bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
