// What does a vector / scalar instruction cost at 8 waves per SIMD on gfx950?  Every kernel below runs 32 waves per CU
// (1024 workgroups of 512 threads, 8 waves per SIMD, two rounds over the chip) through a loop of independent
// instructions of one kind and reports cycles per instruction per SIMD (VALU) or per CU (SALU) from the hipEvent
// time and the in-kernel clock (s_memtime / s_memrealtime).  The mixes at the end are the fold's group:
// ds_read_b64 + v_pk_add_f32 (+ address adds + scalar loop code).  (s_add_u32 writes SCC: without the "scc" clobber the
// loop around the scalar modes never ends.)
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(int iters, float* out, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float*)smem)[i] = 1e-6f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned addr = (unsigned)(size_t)smem + lane * 8;
  f2 a0 = {1.0f, 2.0f}, a1 = {3.0f, 4.0f}, a2 = {5.0f, 6.0f}, a3 = {7.0f, 8.0f}, inc = {1e-3f, 2e-3f};
  float s0 = 1.0f, s1 = 2.0f, s2 = 3.0f, s3 = 4.0f, si = 1e-3f;
  int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
  double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0, di = 1e-3;
  int sc = 0;
  f2 l0, l1, l2, l3;
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 16 v_pk_add_f32, 4 independent chains
      REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(inc));)
    } else if (MODE == 1) {  // v_add_f32
      REP16(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(si));)
    } else if (MODE == 2) {  // v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane));)
    } else if (MODE == 3) {  // v_add_f64
      REP16(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(di));)
    } else if (MODE == 4) {  // v_add_f32 with DPP
      REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));)
    } else if (MODE == 5) {  // s_add_u32 (64 per iteration)
      REP16(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(sc) : : "scc");)
    } else if (MODE == 6) {  // 2 v_pk_add_f32 + 2 s_add_u32 interleaved (x16): do the scalar ones ride along?
      REP16(asm volatile("v_pk_add_f32 %0, %0, %3\n\ts_add_u32 %2, %2, 1\n\tv_pk_add_f32 %1, %1, %3\n\ts_add_u32 %2, %2, 3"
                         : "+v"(a0), "+v"(a1), "+s"(sc) : "v"(inc) : "scc");)
    } else if (MODE == 7) {  // the fold's group: 16 ds_read_b64, wait, 16 v_pk_add_f32
      REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(addr)); a0 += l0;)
    } else if (MODE == 8) {  // group of 4 loads issued together, one wait, 4 pk adds, one address add, scalar loop code (x4)
      for (int g = 0; g < 4; ++g) {
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3) : "v"(addr));
        a0 += l0; a1 += l1; a2 += l2; a3 += l3;
        addr += 2048;
        addr &= 0x3fff;
      }
    } else if (MODE == 9) {  // 12 loads (3 rows x 4 chunks), one wait, 8 pk adds + 4 pk fma + 3 address adds: the NR=3 group
      f2 m0, m1, m2, m3, n0, n1, n2, n3;
      unsigned ad1 = addr + 4104, ad2 = addr + 8208;
      asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:512\n\tds_read_b64 %2, %12 offset:1024\n\tds_read_b64 %3, %12 offset:1536\n\t"
                   "ds_read_b64 %4, %13\n\tds_read_b64 %5, %13 offset:512\n\tds_read_b64 %6, %13 offset:1024\n\tds_read_b64 %7, %13 offset:1536\n\t"
                   "ds_read_b64 %8, %14\n\tds_read_b64 %9, %14 offset:512\n\tds_read_b64 %10, %14 offset:1024\n\tds_read_b64 %11, %14 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3), "=v"(m0), "=v"(m1), "=v"(m2), "=v"(m3), "=v"(n0), "=v"(n1), "=v"(n2), "=v"(n3)
                   : "v"(addr), "v"(ad1), "v"(ad2));
      l0 += m0; l1 += m1; l2 += m2; l3 += m3;
      l0 += n0; l1 += n1; l2 += n2; l3 += n3;
      a0 = __builtin_elementwise_fma(l0, l0, a0); a1 = __builtin_elementwise_fma(l1, l1, a1);
      a2 = __builtin_elementwise_fma(l2, l2, a2); a3 = __builtin_elementwise_fma(l3, l3, a3);
      addr = (addr + 2048) & 0x1fff;
    }
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 5) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + s0 + s1 + s2 + s3 + (float)(i0 + i1 + i2 + i3) + (float)(d0 + d1 + d2 + d3) + (float)sc;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const int blocks = 2048, threads = 512, iters = argc > 2 ? atoi(argv[2]) : 2000;
  float* o;
  long long* clk;
  hipMalloc(&o, (size_t)blocks * threads * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct Row { const char* name; int per_iter; bool scalar; } rows[10] = {
      {"v_pk_add_f32            ", 64, false}, {"v_add_f32               ", 64, false}, {"v_add_u32               ", 64, false},
      {"v_add_f64               ", 64, false}, {"v_add_f32_dpp           ", 64, false}, {"s_add_u32               ", 64, true},
      {"2 v_pk_add + 2 s_add x16", 32, false}, {"ds_read_b64+wait+pk_add ", 16, false}, {"4 loads,wait,4 pk_add x4", 16, false},
      {"NR=3 group: 12 loads    ", 12, false}};
  for (int mode = 0; mode < 10; ++mode) {
    if (only >= 0 && mode != only) continue;
    void (*fn)(int, float*, long long*) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4>
                                        : mode == 5 ? k<5> : mode == 6 ? k<6> : mode == 7 ? k<7> : mode == 8 ? k<8> : k<9>;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 32768, 0, iters, o, clk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = 100.0 * (double)h[0] / (double)h[1];
    // instructions per SIMD: 8 waves per SIMD per round, blocks / (256 CUs x 4 workgroups per CU) rounds
    const double rounds = (double)blocks / 1024.0;
    const double per_simd = rounds * 8.0 * iters * rows[mode].per_iter;
    const double cyc = best * 1e-3 * mhz * 1e6;
    printf("%s: %.3f ms  clock %.0f MHz  %.2f cycles per instruction per %s\n", rows[mode].name, best, mhz,
           rows[mode].scalar ? cyc / (per_simd * 4.0) : cyc / per_simd, rows[mode].scalar ? "CU" : "SIMD");
  }
  return 0;
}
