// Which XCD does workgroup b of a 1-D grid run on?  Prints HW_REG_XCC_ID per block for a few grid sizes and with a second
// stream busy: the layer launches (csrc/scann_layer.hip) rely on "all blocks with the same b % 8 share an XCD".
// hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
int main() {
  unsigned* d;
  hipMalloc((void**)&d, 1 << 20);
  hipStream_t s2;
  hipStreamCreate(&s2);
  for (int pass = 0; pass < 4; ++pass) {
    const int n = pass == 0 ? 64 : pass == 1 ? 4001 : 4001;
    if (pass == 3) hipLaunchKernelGGL(probe, dim3(3000), dim3(256), 0, s2, d + 100000, 2000);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, pass ? 200 : 0);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    printf("pass %d n %d raw[0..15]:", pass, n);
    for (int i = 0; i < 16; ++i) printf(" %x", h[i]);
    int bad = 0, hist[8][16] = {};
    for (int i = 0; i < n; ++i) hist[i & 7][h[i] & 15]++;
    printf("\n");
    for (int x = 0; x < 8; ++x) {
      printf("  b%%8=%d:", x);
      int nz = 0;
      for (int k = 0; k < 16; ++k) if (hist[x][k]) { printf(" xcc%d x%d", k, hist[x][k]); ++nz; }
      if (nz != 1) ++bad;
      printf("\n");
    }
    printf("  classes spread over more than one XCC: %d\n", bad);
  }
  return 0;
}
