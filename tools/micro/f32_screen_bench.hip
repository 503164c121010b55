// Micro-benchmark for a mixed-precision sweep screen: how fast is the wave-per-period fold
// when the LDS-resident window is kept as fp32 pairs (two shifted copies so that every row is 8-byte aligned),
// read with ds_read_b64 (2 residues per lane per load) and summed with v_pk_add_f32 -- against the same loop
// on the fp64 window (1 residue per lane per load, v_add_f64)?  Both variants use the simplified geometry
// "every residue has R = ceil(N / p) rows, zeros behind the window", single-period passes, p = 683..1365.
//   hipcc --offload-arch=gfx950 -O3 f32_screen_bench.hip -o f32_screen_bench && ./f32_screen_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
constexpr int N = 4096, PAD = 2048, WAVES = 8;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const volatile double lds_cd;
typedef __attribute__((address_space(3))) const volatile v2f lds_cv2;

template <int NR, bool MASK>
__device__ __forceinline__ void group_f64(const double* xs, int p, int c0, int lane, double& acc) {
  double v[NR][4];
  lds_cd* ptr = (lds_cd*)(xs + 64 * c0 + lane);
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) v[r][c] = ptr[r * p + 64 * c];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double t = v[0][c];
#pragma unroll
    for (int r = 1; r < NR; ++r) t += v[r][c];
    if (MASK) t = (64 * (c0 + c) + lane < p) ? t : 0.0;
    acc = fma(t, t, acc);
  }
}
template <int NR>
__device__ __forceinline__ double fold_f64(const double* xs, int p, int lane) {
  const int nch = (p + 63) >> 6, whole = p >> 6;
  double acc = 0.0;
  int c0 = 0;
  for (; c0 + 4 <= whole; c0 += 4) group_f64<NR, false>(xs, p, c0, lane, acc);
  if (c0 < nch) group_f64<NR, true>(xs, p, c0, lane, acc);  // reads zeros / other rows behind p: masked
  return acc;
}

// copies: A[n] = x[n], B[n] = x[n + 1] (floats); row r of period p starts at element r p: aligned pair in A when r p is
// even, in B at index r p - 1 when odd
template <int NR, bool MASK>
__device__ __forceinline__ void group_f32(const float* A, const float* B, int p, bool oddp, int c0, int lane, v2f& acc) {
  v2f v[NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const bool useB = oddp && (r & 1);
    const float* base = useB ? B + r * p - 1 : A + r * p;
    lds_cv2* ptr = (lds_cv2*)(base + 128 * c0 + 2 * lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[r][c] = ptr[64 * c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v2f t = v[0][c];
#pragma unroll
    for (int r = 1; r < NR; ++r) t += v[r][c];
    if (MASK) {
      const int j = 128 * (c0 + c) + 2 * lane;
      t.x = j < p ? t.x : 0.f;
      t.y = j + 1 < p ? t.y : 0.f;
    }
    acc = __builtin_elementwise_fma(t, t, acc);
  }
}
template <int NR>
__device__ __forceinline__ double fold_f32(const float* A, const float* B, int p, int lane) {
  const int nch = (p + 127) >> 7, whole = p >> 7;
  v2f acc = {0.f, 0.f};
  const bool oddp = p & 1;
  int c0 = 0;
  for (; c0 + 4 <= whole; c0 += 4) group_f32<NR, false>(A, B, p, oddp, c0, lane, acc);
  if (c0 < nch) group_f32<NR, true>(A, B, p, oddp, c0, lane, acc);
  return (double)acc.x + (double)acc.y;
}

template <bool F32>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(const double* x, double* out, int p_lo, int p_hi, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* xs = (double*)smem;
  float* A = (float*)smem;
  float* B = A + N + PAD;
  const double* g = x + (size_t)blockIdx.x * N;
  if (F32) {
    for (int n = threadIdx.x; n < N + PAD; n += blockDim.x) {
      A[n] = n < N ? (float)g[n] : 0.f;
      B[n] = n + 1 < N ? (float)g[n + 1] : 0.f;
    }
  } else {
    for (int n = threadIdx.x; n < N + PAD; n += blockDim.x) xs[n] = n < N ? g[n] : 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double tot = 0.0;
  for (int rep = 0; rep < reps; ++rep)
    for (int p = p_lo + wv; p <= p_hi; p += WAVES) {
      const int R = (N + p - 1) / p;
      double a;
      if (F32) {
        switch (R) { case 4: a = fold_f32<4>(A, B, p, lane); break; case 5: a = fold_f32<5>(A, B, p, lane); break; default: a = fold_f32<6>(A, B, p, lane); }
      } else {
        switch (R) { case 4: a = fold_f64<4>(xs, p, lane); break; case 5: a = fold_f64<5>(xs, p, lane); break; default: a = fold_f64<6>(xs, p, lane); }
      }
      tot += a * (1.0 / R);
    }
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  if (lane == 0) atomicAdd(out + blockIdx.x, tot);
}

int main() {
  const int W = 1024, p_lo = 683, p_hi = 1365, reps = 4;
  std::vector<double> h((size_t)W * N);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
  double *d, *o;
  hipMalloc(&d, h.size() * 8); hipMalloc(&o, W * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double res[2];
  for (int f32 = 0; f32 < 2; ++f32) {
    const size_t lds = f32 ? 2 * (N + PAD) * 4 : (N + PAD) * 8;
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      hipMemset(o, 0, W * 8);
      hipEventRecord(e0);
      if (f32) hipLaunchKernelGGL(k<true>, dim3(W), dim3(64 * WAVES), lds, 0, d, o, p_lo, p_hi, reps);
      else hipLaunchKernelGGL(k<false>, dim3(W), dim3(64 * WAVES), lds, 0, d, o, p_lo, p_hi, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    std::vector<double> r(W); hipMemcpy(r.data(), o, W * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (double v : r) sum += v;
    res[f32] = sum;
    const double passes = (double)W * reps * (p_hi - p_lo + 1);
    printf("%s: %.3f ms  %.2f G window-passes/s  logical LDS %.1f TB/s (N x 8 B per pass)  checksum %.9e\n", f32 ? "fp32 pairs" : "fp64      ", best, passes / best / 1e6,
           passes * N * 8 / (best * 1e-3) / 1e12, sum);
  }
  printf("relative difference of the checksums %.2e\n", fabs(res[1] - res[0]) / fabs(res[0]));
  return 0;
}
