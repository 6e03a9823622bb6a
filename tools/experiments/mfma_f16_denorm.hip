// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL operands, or flush them?  (Range question of the fp16-operand scheme,
// DESIGN section 9.)  build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_denorm tools/experiments/mfma_f16_denorm.hip && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float a_val, float b_val, float* out) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);     // every output = 32 * a * b
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)(_Float16)a_val; }
}

int main() {
  float* d;
  hipMalloc(&d, 8);
  const float vals[] = {1.0f, 6.2e-5f /* just normal */, 3.0e-5f /* subnormal */, 1.0e-6f, 6.0e-8f /* smallest subnormal */};
  for (float v : vals) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, v, 1.0f, d);
    float h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.3e (as fp16 %.6e)  b = 1: mfma sum of 32 products = %.6e  expected %.6e  -> %s\n", v, h[1], h[0], 32.0 * h[1],
           h[0] == 32.0f * h[1] ? "exact" : (h[0] == 0.f ? "FLUSHED" : "differs"));
  }
  // subnormal x subnormal-scale partner: a = 3e-5 (subnormal), b = 1024
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 3.0e-5f, 1024.0f, d);
  float h[2];
  hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("a = 3e-5 (subnormal), b = 1024: %.6e (expected %.6e)\n", h[0], 32.0 * h[1] * 1024.0);
  return 0;
}
