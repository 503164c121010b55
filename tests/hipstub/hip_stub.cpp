// Host-memory stand-in for the HIP runtime entry points that pyperiod_amd/csrc/period_hip.hip calls.  TEST
// INFRASTRUCTURE ONLY: it lets the HOST half of the library (argument checks, pass plans, geometry and CSR tables,
// staging, LDS layout decisions) run under AddressSanitizer / UBSan on a machine without a GPU.  Kernel launches are
// accepted and do nothing; device memory is host memory.  Nothing here is linked into libperiod_hip.so.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

extern "C" {
void** __hipRegisterFatBinary(const void*) {
  static void* handle = nullptr;
  return &handle;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* s, hipStream_t* st) {
  *g = dim3(1);
  *b = dim3(1);
  *s = 0;
  *st = nullptr;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof *p);
  p->multiProcessorCount = 256;
  p->sharedMemPerBlock = 64 * 1024;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->sharedMemPerBlockOptin = 160 * 1024;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = reinterpret_cast<hipStream_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  std::free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = reinterpret_cast<hipEvent_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  std::free(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.0f;
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
  *p = std::calloc(1, n ? n : 1);  // exact size: ASan sees any byte the host code touches past it
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
}
