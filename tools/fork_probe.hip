// Micro-benchmark: what does a cross-stream dependency cost the PRODUCING stream?
//   (a) hipEventRecord(main) + hipStreamWaitEvent(side)            -- what scann_train_backward's fork() does
//   (b) the producer kernel's last workgroup stores a sequence number, the side stream waits with hipStreamWaitValue64(>=)
//   (c) no dependency at all (lower bound)
//   (d) the producer kernel launched with hipExtLaunchKernelGGL(..., stopEvent): the event IS the kernel's own completion signal -- no
//       marker packet behind it on the producing stream -- + hipStreamWaitEvent(side)
// Main stream: N x [busy kernel A, busy kernel B]; side stream: one small kernel per iteration after A.
// Build: hipcc --offload-arch=gfx950 -O2 tools/fork_probe.hip -o /tmp/fork_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void busy(float* p, int iters, unsigned long long* flag, unsigned* counter, unsigned long long seq) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (flag) {  // last workgroup to finish publishes the sequence number
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = atomicAdd(counter, 1u);
      if (old == gridDim.x - 1) {
        *counter = 0;
        __threadfence();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

int main() {
  const int N = 200, WG = 270, T = 256, ITERS = 2000;
  float* buf;
  CK(hipMalloc(&buf, WG * T * 4 * 2));
  CK(hipMemset(buf, 0, WG * T * 4 * 2));
  unsigned long long* flag;
  unsigned* counter;
  CK(hipMalloc(&flag, 8));
  CK(hipMalloc(&counter, 4));
  CK(hipMemset(flag, 0, 8));
  CK(hipMemset(counter, 0, 4));
  hipStream_t s, aux;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  hipEvent_t ev[64];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  unsigned long long seq = 0;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        ++seq;
        if (mode == 3) {
          hipExtLaunchKernelGGL(busy, dim3(WG), dim3(T), 0, s, nullptr, ev[i % 64], 0, buf, ITERS, (unsigned long long*)nullptr, counter, seq);
          CK(hipStreamWaitEvent(aux, ev[i % 64], 0));
        } else
        hipLaunchKernelGGL(busy, dim3(WG), dim3(T), 0, s, buf, ITERS, mode == 1 ? flag : nullptr, counter, seq);
        if (mode == 0) {
          CK(hipEventRecord(ev[i % 64], s));
          CK(hipStreamWaitEvent(aux, ev[i % 64], 0));
        } else if (mode == 1) {
          CK(hipStreamWaitValue64(aux, flag, seq, hipStreamWaitValueGte));
        }
        hipLaunchKernelGGL(busy, dim3(64), dim3(T), 0, aux, buf + WG * T, ITERS / 4, nullptr, counter, 0ull);
        hipLaunchKernelGGL(busy, dim3(WG), dim3(T), 0, s, buf, ITERS, nullptr, counter, 0ull);
      }
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      CK(hipDeviceSynchronize());
      printf("mode %d (%s): %.2f us per iteration (two main-stream kernels + one side kernel)\n", mode,
             mode == 0 ? "event record + wait" : mode == 1 ? "in-kernel flag + hipStreamWaitValue64" : mode == 2 ? "no dependency" : "stop event of the kernel's own launch + wait", us / N);
    }
  }
  return 0;
}
