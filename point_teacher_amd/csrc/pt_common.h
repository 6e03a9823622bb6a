// Shared helpers for the gfx950 kernels of libpt_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pt_hip.h"

namespace pt {

void set_error(const char* fmt, ...);

#define PT_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      pt::set_error(__VA_ARGS__);        \
      return (code);                     \
    }                                    \
  } while (0)

// Check the launch that was just enqueued (no synchronisation).
#define PT_LAUNCH_CHECK(name)                                                  \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      pt::set_error("%s: launch failed: %s", name, hipGetErrorString(_e));     \
      return (int)_e;                                                          \
    }                                                                          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread.
// `sm` must hold >= 17 floats.
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float t = (lane < nw) ? sm[lane] : 0.f;
  t = wave_sum(t);
  return t;
}

// fp32 -> three bf16 terms x = x0 + x1 + x2 (8 + 8 + 8 significant bits, round-to-nearest at each step, exact): the operand format of
// the bf16x6 matrix kernels (csrc/gemm_split.hip)
__device__ __forceinline__ unsigned bf16_rne_pair(float a, float b) {
  // v_cvt_pk_bf16_f32: two fp32 -> packed bf16 (a in the low half), round to nearest even, NaN stays NaN
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}

// x -> (x0, x1, x2) for two values at once; planes receive the packed pairs
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = bf16_rne_pair(a, b);
  const float a1 = a - __uint_as_float(p0 << 16), b1 = b - __uint_as_float(p0 & 0xffff0000u);   // exact
  p1 = bf16_rne_pair(a1, b1);
  const float a2 = a1 - __uint_as_float(p1 << 16), b2 = b1 - __uint_as_float(p1 & 0xffff0000u); // exact
  p2 = bf16_rne_pair(a2, b2);
}


// fp32 -> TWO fp16 terms x = h0 + h1 (11 + 11 significant bits: 2^-23 relative; <= 3e-8 absolute below 0.125, where h1 is subnormal -
// the matrix cores multiply subnormals exactly): the operand format of the three-product fp16 instantiations of the matrix kernels
// (csrc/gemm_split.hip; the MIL head's 12 544 -> 1 024 layer).  Tensors are scaled by powers of two into fp16's range by their producer.
constexpr float F16_WEIGHT_SCALE = PT_F16_WEIGHT_SCALE;   // weights (|w| ~ 0.01 ... 1) are stored as fp16 planes of 16 w: h1 stays a normal number down
                                           // to |w| ~ 0.008; the consumer's alpha carries the 1 / 16 (include/pt_hip.h: PT_F16_WEIGHT_SCALE)
constexpr float F16_SAT = 60000.f;         // magnitudes beyond fp16's range saturate instead of becoming inf
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& p0, unsigned& p1) {
  a = fminf(fmaxf(a, -F16_SAT), F16_SAT);
  b = fminf(fmaxf(b, -F16_SAT), F16_SAT);
  f16x2_t h0;
  h0[0] = (_Float16)a;                                  // round to nearest even
  h0[1] = (_Float16)b;
  f16x2_t h1;
  h1[0] = (_Float16)(a - (float)h0[0]);                 // the residual is exact in fp32
  h1[1] = (_Float16)(b - (float)h0[1]);
  p0 = __builtin_bit_cast(unsigned, h0);
  p1 = __builtin_bit_cast(unsigned, h1);
}

// sigmoid exactly as torch's CPU/CUDA kernels compute it: 1 / (1 + exp(-x))
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace pt
