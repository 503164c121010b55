// Checks the semantics of __builtin_amdgcn_global_load_lds on gfx950: per-lane global source,
// LDS destination = wave-uniform base + lane * size.  hipcc --offload-arch=gfx950 -O3 glds_test.hip -o glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;
__global__ void k(const double* __restrict__ src, const int* __restrict__ isrc, double* out, int* iout, int n) {
  __shared__ __attribute__((aligned(16))) double buf[1024];
  __shared__ int ibuf[128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = -1.0;
  if (threadIdx.x < 128) ibuf[threadIdx.x] = -1;
  __syncthreads();
  // wave wv copies elements [128 wv, 128 wv + 128) of src + 6 (16-byte aligned: element offset even)
  __builtin_amdgcn_global_load_lds((glb_vp)(src + 6 + 128 * wv + 2 * lane), (lds_vp)(buf + 128 * wv), 16, 0, 0);
  if (wv == 0) __builtin_amdgcn_global_load_lds((glb_vp)(isrc + 3 + lane), (lds_vp)ibuf, 4, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = buf[i];
  if (threadIdx.x < 128) iout[threadIdx.x] = ibuf[threadIdx.x];
}
int main() {
  const int n = 4096;
  std::vector<double> h(n);
  std::vector<int> hi(n);
  for (int i = 0; i < n; ++i) { h[i] = i * 0.5; hi[i] = 7 * i; }
  double *d, *o; int *di, *oi;
  hipMalloc(&d, n * 8); hipMalloc(&o, 1024 * 8); hipMalloc(&di, n * 4); hipMalloc(&oi, 128 * 4);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, di, o, oi, n);
  std::vector<double> r(1024); std::vector<int> ri(128);
  hipMemcpy(r.data(), o, 1024 * 8, hipMemcpyDeviceToHost); hipMemcpy(ri.data(), oi, 128 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (r[i] != h[6 + i]) { if (bad < 5) printf("buf[%d]=%g want %g\n", i, r[i], h[6 + i]); ++bad; }
  for (int i = 0; i < 64; ++i) if (ri[i] != hi[3 + i]) { if (bad < 10) printf("ibuf[%d]=%d want %d\n", i, ri[i], hi[3 + i]); ++bad; }
  for (int i = 64; i < 128; ++i) if (ri[i] != -1) ++bad;
  printf("glds test: %d mismatches\n", bad);
  return bad != 0;
}
