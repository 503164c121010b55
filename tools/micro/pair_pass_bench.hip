// Test bed for the screen passes of the window-pair kernels (ph_pair.h): every wavefront of a workgroup walks the
// periods q_lo + wave, q_lo + wave + 8, ... < q_hi over one LDS window pair, as the screens of k_small_to_large_pair /
// k_mbest_step1_pair do, at their occupancy (4 workgroups of 8 wavefronts per CU).  Variant 1 is pair_pass_single,
// 2 / 4 are pair_pass_multi with two / four classes (period q, 2q[, 4q] from one fold).  Every value is checked against
// the fp64 fold of the same float samples in units of the rigorous radius of the screen (pair_radius).  Time from hipEvents;
// instruction counts per pass with tools/micro/pair_pass_pmc.sh.  profiles/r4_study/pair_pass_bench.txt holds the
// numbers of the round-3 passes (one loop over the two segments) next to these.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../pyperiod_amd/csrc pair_pass_bench.hip -o pair_pass_bench
#include "ph_pair.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

using namespace ph;

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(const PGeomF* __restrict__ geom, const f2* __restrict__ win,
                                                                                   int N, int q_lo, int q_hi, int reps, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f2* xs = (f2*)smem;
  for (int i = threadIdx.x; i < N + 64; i += blockDim.x) xs[i] = i < N ? win[i] : f2_zero();  // the pad is read (straddling chunks)
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform, as the pass queues of the kernels deliver it
  for (int r = 0; r < reps; ++r) {
    for (int q = q_lo + wave; q < q_hi; q += 8) {
      f2* vals = reinterpret_cast<f2*>(out);
      const bool keep = blockIdx.x == 0 && r == 0;
      if (V == 1) {
        const float z = pair_reduce1<false>(pair_pass_single<false>(xs, q, geom[q]));
        if (keep) pair_store1(vals, z, q, 0);
      } else {
        f2 part[3];
        pair_pass_multi<(V == 1 ? 2 : V), false>(xs, q, geom, part);
        const float z = pair_reduce2<false>(part[0], part[1]);
        if (keep) pair_store2(vals, z, q, 2 * q, 0);
        if (V == 4) {
          const float z4 = pair_reduce1<false>(part[2]);
          if (keep) pair_store1(vals, z4, 4 * q, 0);
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int v = argc > 1 ? atoi(argv[1]) : 1;
  const int N = 4096, q_lo = argc > 2 ? atoi(argv[2]) : 64, q_hi = argc > 3 ? atoi(argv[3]) : 2048, reps = 2, blocks = 2048;
  if (v != 1 && v != 2 && v != 4) return 2;
  if (q_lo < 64 || v * (q_hi - 1) > N) return 2;  // the passes need q >= 64 and whole class cycles inside the window
  std::vector<PGeomF> g(N + 1);
  for (int p = 1; p <= N; ++p) {
    const int rows = (N + p - 1) / p, shortn = rows * p - N;
    g[p] = PGeomF{rows, p - shortn, (float)(1.0 / rows), rows > 1 ? (float)(1.0 / (rows - 1)) : 0.0f};
  }
  // the window pair of every workgroup: two sinusoids plus a ramp, values of order 1 (the kernels scale to RMS ~ 1)
  std::vector<float> hw(2 * (size_t)N);
  for (int i = 0; i < N; ++i) {
    const float va = sinf(0.37f * i) + 0.25f * cosf(1.1f * i);
    hw[2 * i] = va;
    hw[2 * i + 1] = 0.5f * va + 0.125f + 1e-4f * (i % 97);
  }
  f2* dwin;
  hipMalloc(&dwin, hw.size() * sizeof(float));
  hipMemcpy(dwin, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
  PGeomF* dg;
  float* out;
  hipMalloc(&dg, g.size() * sizeof(PGeomF));
  hipMemcpy(dg, g.data(), g.size() * sizeof(PGeomF), hipMemcpyHostToDevice);
  hipMalloc(&out, 2 * (N + 1) * sizeof(float));
  hipMemset(out, 0, 2 * (N + 1) * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = (size_t)(N + 64) * sizeof(f2);
  void (*fn)(const PGeomF*, const f2*, int, int, int, int, float*) = v == 1 ? k<1> : v == 2 ? k<2> : k<4>;
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds, 0, dg, dwin, N, q_lo, q_hi, reps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double passes = (double)blocks * reps * (q_hi - q_lo);
  {  // every stored total against the double-precision fold of the same float samples, in units of the rigorous radius
    std::vector<float> h(2 * (N + 1));
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    double ssq[2] = {0, 0};
    for (int i = 0; i < N; ++i)
      for (int w = 0; w < 2; ++w) ssq[w] += (double)hw[2 * i + w] * hw[2 * i + w];
    double worst = 0, worst_rel = 0;
    int at = 0, checked = 0;
    for (int q = q_lo; q < q_hi; ++q)
      for (int m = 1; m <= v; m *= 2) {
        const int Q = m * q, rows = (N + Q - 1) / Q;
        const double radius = 1.5 * (2.0 * rows + (double)(Q >> 6) + 32.0) * 5.9604644775390625e-08;  // pair_radius(rows, Q)
        for (int w = 0; w < 2; ++w) {
          double ss = 0;
          for (int j = 0; j < Q; ++j) {
            double sj = 0;
            int cnt = 0;
            for (int i = j; i < N; i += Q) sj += (double)hw[2 * i + w], cnt += 1;
            if (cnt) ss += sj * sj / cnt;
          }
          const double err = fabs(ss - (double)h[2 * Q + w]);
          worst_rel = fmax(worst_rel, err / fmax(1e-30, ss));
          if (err / (radius * ssq[w]) > worst) worst = err / (radius * ssq[w]), at = Q;
          checked += 1;
        }
      }
    printf("%d screen values against the fp64 fold: largest |error| / (pair_radius x sum of squares) = %.3f (period %d; must be < 1), largest relative error %.2e\n",
           checked, worst, at, worst_rel);
  }
  printf("%d-class pass: %.3f ms for %.0f passes (q in [%d, %d)): %.1f ns per pass per CU\n", v, best, passes, q_lo, q_hi,
         best * 1e6 / (passes / 256.0));
  return 0;
}
