// Test bed for the single-period screen pass of the window-pair kernels (ph_pair.h): every wavefront of a workgroup
// walks the periods q_lo + wave, q_lo + wave + 8, ... < q_hi over one LDS window pair, as the screens of
// k_small_to_large_pair / k_mbest_step1_pair do, at their occupancy (4 workgroups of 8 wavefronts per CU).  Variant 0 is
// pair_pass_seg<1>, 1 is pair_pass_single, 2 / 3 are the two- and four-class passes (period q, 2q, 4q from one fold); the values of all variants are compared per period.  Time from
// hipEvents; instruction counts per pass with
//   rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES -- ./pair_pass_bench <variant>
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../pyperiod_amd/csrc pair_pass_bench.hip -o pair_pass_bench
#include "ph_pair.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

using namespace ph;

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(const PGeomF* __restrict__ geom, int N, int q_lo,
                                                                                   int q_hi, int reps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f2* xs = (f2*)smem;
  for (int i = threadIdx.x; i < N + 64; i += blockDim.x) {
    const float v = i < N ? __sinf(0.37f * i + 0.001f * blockIdx.x) + 0.25f * __cosf(1.1f * i) : 0.0f;
    xs[i] = f2_make(v, 0.5f * v + 0.125f);
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform, as the pass queues of the kernels deliver it
  for (int r = 0; r < reps; ++r) {
    for (int q = q_lo + wave; q < q_hi; q += 8) {
      f2 tot;
      if (V == 0) {
        f2 part[3];
        pair_pass_seg<1, false>(xs, q, geom, part);
        tot = pair_wave_sum(part[0]);
      } else if (V == 1) {
        tot = pair_wave_sum(pair_pass_single<false>(xs, q, geom[q]));
      } else if (V == 2 || V == 3) {  // the multi-class passes as k_mbest_step1_pair runs them (q, 2q[, 4q] from one fold)
        f2 part[3];
        pair_pass_seg<V == 2 ? 2 : 4, false>(xs, q, geom, part);
        tot = pair_wave_red2<false>(part[0], part[1]);
        if (V == 3) tot += pair_wave_sum(part[2]);
      } else {  // 4 / 5: pair_pass_multi, the straight-line version of 2 / 3
        f2 part[3];
        pair_pass_multi<V == 4 ? 2 : 4, false>(xs, q, geom, part);
        tot = pair_wave_red2<false>(part[0], part[1]);
        if (V == 5) tot += pair_wave_sum(part[2]);
      }
      if (blockIdx.x == 0 && r == 0 && pair_lane() == 0) {
        out[2 * q] = tot.x;
        out[2 * q + 1] = tot.y;
      }
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const int N = 4096, q_lo = argc > 2 ? atoi(argv[2]) : 64, q_hi = argc > 3 ? atoi(argv[3]) : 2048, reps = 2, blocks = 2048;
  std::vector<PGeomF> g(N + 1);
  for (int p = 1; p <= N; ++p) {
    const int rows = (N + p - 1) / p, shortn = rows * p - N;
    g[p] = PGeomF{rows, p - shortn, (float)(1.0 / rows), rows > 1 ? (float)(1.0 / (rows - 1)) : 0.0f};
  }
  PGeomF* dg;
  float* out[2];
  hipMalloc(&dg, g.size() * sizeof(PGeomF));
  hipMemcpy(dg, g.data(), g.size() * sizeof(PGeomF), hipMemcpyHostToDevice);
  for (int v = 0; v < 2; ++v) {
    hipMalloc(&out[v], 2 * (N + 1) * sizeof(float));
    hipMemset(out[v], 0, 2 * (N + 1) * sizeof(float));
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = (size_t)(N + 64) * sizeof(f2);
  for (int v = 0; v < 6; ++v) {
    if (only >= 0 ? v != only && v != only + 2 : v >= 2) continue;
    void (*fn)(const PGeomF*, int, int, int, int, float*) = v == 0 ? k<0> : v == 1 ? k<1> : v == 2 ? k<2> : v == 3 ? k<3> : v == 4 ? k<4> : k<5>;
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds, 0, dg, N, q_lo, q_hi, reps, out[(v == 1 || v >= 4) ? 1 : 0]);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double passes = (double)blocks * reps * (q_hi - q_lo);
    // per CU: passes / 256; cycles at 2.4 GHz nominal (the counters give the exact instruction counts)
    printf("variant %d: %.3f ms for %.0f passes (q in [%d, %d)): %.1f ns per pass per CU\n", v, best, passes, q_lo, q_hi,
           best * 1e6 / (passes / 256.0));
  }
  if (only < 0 || only == 2 || only == 3) {
    std::vector<float> h0(2 * (N + 1)), h1(2 * (N + 1));
    hipMemcpy(h0.data(), out[0], h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), out[1], h1.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    int at = 0;
    for (int q = q_lo; q < q_hi; ++q)
      for (int s = 0; s < 2; ++s) {
        const double d = fabs((double)h0[2 * q + s] - (double)h1[2 * q + s]) / fmax(1e-30, fabs((double)h0[2 * q + s]));
        if (d > worst) worst = d, at = q;
      }
    printf("largest relative difference between the variants: %.3g (q = %d)\n", worst, at);
  }
  return 0;
}
