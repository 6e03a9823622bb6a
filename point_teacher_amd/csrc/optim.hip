// Flat-buffer EMA, gradient-norm and SGD step for gfx950.  The student's trainable
// parameters, their gradients and momentum live in ONE contiguous fp32 buffer each (weights
// first, biases after `split`), the teacher in another: the ~190 per-tensor launch pairs of
// the reference become three streaming kernels at HBM rate (16-byte accesses, grid-stride
// over 2048 workgroups).  Algorithmic bytes: EMA 12 B/param, norm 4 B/param, SGD 20 B/param.
#include "pt_common.h"

namespace pt {

constexpr int STREAM_BLOCKS = 2048;

__global__ void __launch_bounds__(256)
    ema_kernel(float* __restrict__ t, const float* __restrict__ s, long n, float a, float b) {
  const long n4 = n >> 2;
  float4* t4 = reinterpret_cast<float4*>(t);
  const float4* s4 = reinterpret_cast<const float4*>(s);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 x = t4[i];
    const float4 y = s4[i];
    x.x = x.x * a + b * y.x; x.y = x.y * a + b * y.y; x.z = x.z * a + b * y.z; x.w = x.w * a + b * y.w;
    t4[i] = x;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    t[i] = t[i] * a + b * s[i];
  }
}

__global__ void __launch_bounds__(256) sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float sm[17];
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = g4[i];
    acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float x = g[(n4 << 2) + threadIdx.x];
    acc += x * x;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// Parameter groups of mmcv's DefaultOptimizerConstructor (paramwise_cfg: bias_lr_mult / bias_decay_mult / norm_decay_mult /
// dwconv_decay_mult / dcn_offset_lr_mult / custom_keys): the live trainable buffer is a sequence of <= PT_MAX_PARAM_GROUPS
// contiguous, 16-byte aligned segments, each with its own (lr multiplier, decay multiplier).  The table travels by value.
struct SgdGroups {
  long end[PT_MAX_PARAM_GROUPS];      // exclusive end offsets, ascending; end[n-1] == n
  float lr_mult[PT_MAX_PARAM_GROUPS];
  float wd_mult[PT_MAX_PARAM_GROUPS];
  int n;
};

__device__ __forceinline__ float sgd_one(float x, float g, float& m, float clip, float lri, float wdi, float mom, int first) {
  float d = g * clip;
  if (wdi != 0.f) d = d + wdi * x;
  const float buf = first ? d : mom * m + d;
  m = buf;
  return x - lri * buf;
}

__global__ void __launch_bounds__(256)
    sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n, SgdGroups grp,
               const float* __restrict__ lr_p, float mom, float wd, const float* __restrict__ sqnorm, float max_norm, int first) {
  const float lr = lr_p[0];
  float clip = 1.f;
  if (max_norm > 0.f && sqnorm) {
    const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);  // torch.nn.utils.clip_grad_norm_
    clip = c < 1.f ? c : 1.f;
  }
  // segments are multiples of 4 elements: a float4 never straddles two groups
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long e = i << 2;
    int k = 0;
#pragma unroll
    for (int j = 0; j < PT_MAX_PARAM_GROUPS - 1; ++j) k += (j < grp.n - 1 && e >= grp.end[j]) ? 1 : 0;
    const float lri = lr * grp.lr_mult[k];
    const float wdi = wd * grp.wd_mult[k];
    float4 x = p4[i];
    const float4 gg = g4[i];
    float4 mm = m4[i];
    x.x = sgd_one(x.x, gg.x, mm.x, clip, lri, wdi, mom, first);
    x.y = sgd_one(x.y, gg.y, mm.y, clip, lri, wdi, mom, first);
    x.z = sgd_one(x.z, gg.z, mm.z, clip, lri, wdi, mom, first);
    x.w = sgd_one(x.w, gg.w, mm.w, clip, lri, wdi, mom, first);
    m4[i] = mm;
    p4[i] = x;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {        // tail of a buffer whose length is not a multiple of 4: last group
    const long i = (n4 << 2) + threadIdx.x;
    const int k = grp.n - 1;
    p[i] = sgd_one(p[i], g[i], m[i], clip, lr * grp.lr_mult[k], wd * grp.wd_mult[k], mom, first);
  }
}

}  // namespace pt

using namespace pt;

static int stream_blocks(long n) {
  int nb = cdiv(n, 1024);
  return nb < 1 ? 1 : (nb > STREAM_BLOCKS ? STREAM_BLOCKS : nb);
}

extern "C" int pt_ema_update(float* teacher, const float* student, int64_t n, float alpha, float one_minus_alpha,
                             void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(teacher && student && n > 0, PT_EINVAL, "pt_ema_update: bad argument");
  PT_REQUIRE((((uintptr_t)teacher | (uintptr_t)student) & 15) == 0, PT_EINVAL, "pt_ema_update: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(ema_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), teacher, student, (long)n,
                     alpha, one_minus_alpha);
  PT_LAUNCH_CHECK("pt_ema_update");
  return PT_OK;
}

extern "C" int pt_sqnorm_nblocks(int64_t n) { return stream_blocks(n); }

extern "C" int pt_sqnorm_partial(const float* g, int64_t n, float* partial, void* stream) {
  PT_REQUIRE(g && partial && n > 0, PT_EINVAL, "pt_sqnorm_partial: bad argument");
  PT_REQUIRE(((uintptr_t)g & 15) == 0, PT_EINVAL, "pt_sqnorm_partial: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(sqnorm_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), g, (long)n, partial);
  PT_LAUNCH_CHECK("pt_sqnorm_partial");
  return PT_OK;
}

static int sgd_launch(float* param, const float* grad, float* momentum_buf, int64_t n, const SgdGroups& grp, const float* lr,
                      float momentum, float weight_decay, const float* sqnorm, float max_norm, int first_step, void* stream,
                      const char* who) {
  PT_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0, PT_EINVAL,
             "pt_sgd_step: buffers must be 16-byte aligned");
  long prev = 0;
  for (int j = 0; j < grp.n; ++j) {
    PT_REQUIRE(grp.end[j] >= prev && ((grp.end[j] & 3) == 0 || j == grp.n - 1), PT_EINVAL,
               "pt_sgd_step: group ends must ascend; all but the last in multiples of 4");
    prev = grp.end[j];
  }
  PT_REQUIRE(prev == n, PT_EINVAL, "pt_sgd_step: the last group must end at n");
  hipLaunchKernelGGL(sgd_kernel, dim3(stream_blocks((n >> 2) + 1)), dim3(256), 0, as_stream(stream), param, grad, momentum_buf,
                     (long)n, grp, lr, momentum, weight_decay, sqnorm, max_norm, first_step);
  PT_LAUNCH_CHECK(who);
  return PT_OK;
}

extern "C" int pt_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, int64_t split,
                           const float* lr, float momentum, float weight_decay, float bias_lr_mult,
                           float bias_decay_mult, const float* sqnorm, float max_norm, int first_step, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(param && grad && momentum_buf && lr && n > 0 && split >= 0 && split <= n, PT_EINVAL,
             "pt_sgd_step: bad argument");
  SgdGroups grp{};
  grp.n = 2;
  grp.end[0] = split; grp.lr_mult[0] = 1.f; grp.wd_mult[0] = 1.f;
  grp.end[1] = n; grp.lr_mult[1] = bias_lr_mult; grp.wd_mult[1] = bias_decay_mult;
  return sgd_launch(param, grad, momentum_buf, n, grp, lr, momentum, weight_decay, sqnorm, max_norm, first_step, stream,
                    "pt_sgd_step");
}

extern "C" int pt_sgd_step_groups(float* param, const float* grad, float* momentum_buf, int64_t n, const int64_t* group_end,
                                  const float* lr_mult, const float* decay_mult, int n_groups, const float* lr,
                                  float momentum, float weight_decay, const float* sqnorm, float max_norm, int first_step,
                                  void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(param && grad && momentum_buf && lr && group_end && lr_mult && decay_mult && n > 0, PT_EINVAL,
             "pt_sgd_step_groups: bad argument");
  PT_REQUIRE(n_groups >= 1 && n_groups <= PT_MAX_PARAM_GROUPS, PT_ELIMIT, "pt_sgd_step_groups: 1 <= n_groups <= PT_MAX_PARAM_GROUPS (8)");
  SgdGroups grp{};
  grp.n = n_groups;
  for (int j = 0; j < n_groups; ++j) {
    grp.end[j] = group_end[j]; grp.lr_mult[j] = lr_mult[j]; grp.wd_mult[j] = decay_mult[j];
  }
  return sgd_launch(param, grad, momentum_buf, n, grp, lr, momentum, weight_decay, sqnorm, max_norm, first_step, stream,
                    "pt_sgd_step_groups");
}

// ----------------------------------------------------------------------------------------------
// Frozen-BatchNorm epilogue.  On this path every BatchNorm runs in eval mode with a frozen affine
// (backbones/resnet.py:647-658: norm_eval=True, requires_grad=False), i.e. it is the per-channel map
// y = x*scale[c] + shift[c].  torch runs it as 2-3 separate full passes over the activation (BN,
// residual add, ReLU) forward and 3 more backward; these kernels do each direction in ONE pass:
//   fwd: y = [relu]( x*scale[c] + shift[c] [+ residual] )           (in place on x allowed)
//   bwd: m = relu ? (y > 0) : 1 ;  grad_res = g*m ;  grad_x = g*m*scale[c]
// `inner` = H*W for NCHW tensors and 1 for channels_last ones (channel = (i / inner) % C).
namespace pt {

// Channel bookkeeping without per-element 64-bit division: a thread walks the tensor with a fixed stride, so
// the channel of its float4 advances by a fixed step modulo C (channels_last, inner == 1), or is recomputed
// with one division per float4 (NCHW - not on the benchmark path).
struct ChanWalk {
  int c, step, C;
  __device__ __forceinline__ ChanWalk(long i0, long stride, int C_) : C(C_) {
    c = (int)((i0 << 2) % C_);
    step = (int)((stride << 2) % C_);
  }
  __device__ __forceinline__ void next() {
    c += step;
    c = c >= C ? c - C : c;
  }
};

template <bool CL>
__global__ void __launch_bounds__(256)
    affine_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                           const float* __restrict__ shift, const float* __restrict__ res, long n4, int C,
                           long inner, int relu, float* __restrict__ y) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(res);
  float4* y4 = reinterpret_cast<float4*>(y);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  ChanWalk w(i, stride, C);
  for (; i < n4; i += stride) {
    float4 v = x4[i];
    float4 s, b;
    if (CL) {                      // 4 consecutive channels (C % 4 == 0): two aligned 16-byte loads
      s = *reinterpret_cast<const float4*>(scale + w.c);
      b = *reinterpret_cast<const float4*>(shift + w.c);
      w.next();
    } else {                       // NCHW: inner % 4 == 0 -> one channel for the whole float4
      const int c = (int)(((i << 2) / inner) % C);
      const float sc = scale[c], sh = shift[c];
      s = make_float4(sc, sc, sc, sc);
      b = make_float4(sh, sh, sh, sh);
    }
    v.x = v.x * s.x + b.x; v.y = v.y * s.y + b.y; v.z = v.z * s.z + b.z; v.w = v.w * s.w + b.w;
    if (res) { const float4 r = r4[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    y4[i] = v;
  }
}

template <bool CL>
__global__ void __launch_bounds__(256)
    affine_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ scale,
                           long n4, int C, long inner, int relu, float* __restrict__ gx, float* __restrict__ gres) {
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  float4* gx4 = reinterpret_cast<float4*>(gx);
  float4* gr4 = reinterpret_cast<float4*>(gres);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  ChanWalk w(i, stride, C);
  for (; i < n4; i += stride) {
    float4 v = g4[i];
    if (relu) {
      const float4 o = y4[i];
      v.x = o.x > 0.f ? v.x : 0.f; v.y = o.y > 0.f ? v.y : 0.f; v.z = o.z > 0.f ? v.z : 0.f; v.w = o.w > 0.f ? v.w : 0.f;
    }
    if (gres) gr4[i] = v;
    if (gx) {
      if (CL) {
        const float4 s = *reinterpret_cast<const float4*>(scale + w.c);
        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
      } else {
        const float s = scale[(int)(((i << 2) / inner) % C)];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      }
      gx4[i] = v;
    }
    if (CL) w.next();
  }
}

// Same epilogue for a BatchNorm in eval mode whose affine TRAINS (config 5: norm_cfg requires_grad=True,
// norm_eval=True).  One pass over the activation does the element-wise backward AND the two per-channel
// reductions the weight/bias gradients need:
//     m = relu ? (y > 0) : 1;   grad_res = g*m;   grad_x = g*m*scale[c];
//     sums[c] += g*m            (= dL/dshift)       sums[C + c] += g*m*x      (= dL/dscale)
// channels_last only.  Each thread keeps the same 4 channels for its whole walk (the launch stride is a
// multiple of C/4 float4 groups), accumulates 8 sums in registers, the block combines threads that share a
// channel group through LDS and writes one private row of partial sums; a second tiny kernel adds the rows
// (2 048 blocks x 8 float atomics per group on 2*C addresses cost more than the whole streaming pass).
constexpr int AFF_THREADS = 256;

// HAS_X = false: only the shift gradient is wanted (a convolution bias + ReLU epilogue, scale == 1): x is not read.
template <bool HAS_X>
__global__ void __launch_bounds__(AFF_THREADS)
    affine_relu_bwd_train_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ x,
                                 const float* __restrict__ scale, long n4, int C, int relu, float* __restrict__ gx,
                                 float* __restrict__ gres, float* __restrict__ partial) {
  __shared__ float red[AFF_THREADS][9];                    // padded: 8 sums per thread
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* gx4 = reinterpret_cast<float4*>(gx);
  float4* gr4 = reinterpret_cast<float4*>(gres);
  const int G = C >> 2;                                     // channel groups of 4
  const long stride = (long)gridDim.x * blockDim.x;         // multiple of G by construction (host)
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = (int)(i % G);
  const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * cg);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  for (; i < n4; i += stride) {
    float4 v = g4[i];
    if (relu) {
      const float4 o = y4[i];
      v.x = o.x > 0.f ? v.x : 0.f; v.y = o.y > 0.f ? v.y : 0.f; v.z = o.z > 0.f ? v.z : 0.f; v.w = o.w > 0.f ? v.w : 0.f;
    }
    s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    if (HAS_X) {
      const float4 xv = x4[i];
      t0 += v.x * xv.x; t1 += v.y * xv.y; t2 += v.z * xv.z; t3 += v.w * xv.w;
    }
    if (gres) gr4[i] = v;
    if (gx) gx4[i] = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
  }
  float* r = red[threadIdx.x];
  r[0] = s0; r[1] = s1; r[2] = s2; r[3] = s3; r[4] = t0; r[5] = t1; r[6] = t2; r[7] = t3;
  __syncthreads();
  // threads tid, tid+G, tid+2G, ... of this block share a channel group when G <= blockDim; otherwise every
  // thread of the block has its own group
  const int per = G < AFF_THREADS ? AFF_THREADS / G : 1;
  if (threadIdx.x < (G < AFF_THREADS ? G : AFF_THREADS)) {
    float a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.f;
    for (int u = 0; u < per; ++u) {
      const float* rr = red[threadIdx.x + u * G];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += rr[q];
    }
    // no atomics: block-private row of the workspace (blocks that cover complementary channel groups, C/4 > 256,
    // share a row), reduced by affine_train_finish -> deterministic sums
    const int grp = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % G);
    const int m = G > AFF_THREADS ? G / AFF_THREADS : 1;
    float* row = partial + (size_t)(blockIdx.x / m) * 2 * C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      row[4 * grp + q] = a[q];
      row[C + 4 * grp + q] = a[4 + q];
    }
  }
}

// sums[j] = sum over rows of partial[row][j]: 16 columns x 16 row lanes per workgroup (a column's rows are
// split over 16 threads, combined through LDS).
__global__ void __launch_bounds__(256) affine_train_finish(const float* __restrict__ partial, int rows, int C2,
                                                            float* __restrict__ sums) {
  __shared__ float red[16][17];
  const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + col;
  float a = 0.f;
  if (j < C2)
    for (int r = lane; r < rows; r += 16) a += partial[(size_t)r * C2 + j];
  red[lane][col] = a;
  __syncthreads();
  if (lane == 0 && j < C2) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += red[u][col];
    sums[j] = t;
  }
}

// ---- bf16 activations (configs[2]: bf16 autocast backbone): same epilogue on [N,H,W,C] bf16 tensors, 8 elements
// (16 bytes) per thread and iteration, arithmetic in fp32, scale/shift stay fp32, round-to-nearest-even on store.
__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
struct alignas(16) BF8 { unsigned short v[8]; };

__global__ void __launch_bounds__(256)
    affine_relu_fwd_bf16_kernel(const BF8* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                const BF8* __restrict__ res, long n8, int C, int relu, BF8* __restrict__ y) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c = (int)((i << 3) % C);
  const int step = (int)((stride << 3) % C);
  for (; i < n8; i += stride) {
    const BF8 xv = x[i];
    BF8 rv, o;
    if (res) rv = res[i];
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + c), b1 = *reinterpret_cast<const float4*>(shift + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = bf2f(xv.v[k]) * sc[k] + sh[k];
      if (res) v += bf2f(rv.v[k]);
      if (relu) v = fmaxf(v, 0.f);
      o.v[k] = f2bf(v);
    }
    y[i] = o;
    c += step;
    c = c >= C ? c - C : c;
  }
}

__global__ void __launch_bounds__(256)
    affine_relu_bwd_bf16_kernel(const BF8* __restrict__ g, const BF8* __restrict__ y, const float* __restrict__ scale,
                                long n8, int C, int relu, BF8* __restrict__ gx, BF8* __restrict__ gres) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c = (int)((i << 3) % C);
  const int step = (int)((stride << 3) % C);
  for (; i < n8; i += stride) {
    BF8 gv = g[i];
    if (relu) {
      const BF8 yv = y[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) gv.v[k] = (yv.v[k] & 0x7fffu) != 0 && !(yv.v[k] & 0x8000u) ? gv.v[k] : (unsigned short)0;
    }
    if (gres) gres[i] = gv;
    if (gx) {
      const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      BF8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.v[k] = f2bf(bf2f(gv.v[k]) * sc[k]);
      gx[i] = o;
    }
    c += step;
    c = c >= C ? c - C : c;
  }
}

// ---- top-down step of FPN / PSAGG on NHWC maps: out = a + nearest_upsample(b -> size of a) (necks/fpn.py:165-173,
// necks/ps_fpn.py:64-72; torch's 'nearest': src = min(floor(dst * (float)in / out), in - 1)).  torch runs an upsample
// kernel (96 us for a [4,256,100,100] channels_last output on MI355X: 0.4 TB/s) and an add; this is one streaming pass.
// One workgroup row per output row (blockIdx.y = n * Ha + y), 16 bytes per thread.  V = float4 (fp32) or BF8 (bf16).
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}
__device__ __forceinline__ float4 vadd(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ BF8 vadd(const BF8 a, const BF8 b) {
  BF8 o;
#pragma unroll
  for (int k = 0; k < 8; ++k) o.v[k] = f2bf(bf2f(a.v[k]) + bf2f(b.v[k]));
  return o;
}

template <typename V>
__global__ void __launch_bounds__(256)
    upsample_add_fwd_kernel(const V* __restrict__ a, const V* __restrict__ b, int Ha, int Wa, int Hb, int Wb, int CG,
                            float sh, float sw, V* __restrict__ out) {
  const int row = blockIdx.y;                       // n * Ha + y
  const int n = row / Ha, y = row - n * Ha;
  const int ys = nearest_src(y, sh, Hb);
  const V* brow = b + ((size_t)n * Hb + ys) * Wb * CG;
  const size_t base = (size_t)row * Wa * CG;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Wa * CG; t += gridDim.x * blockDim.x) {
    const int x = t / CG, cg = t - x * CG;
    out[base + t] = vadd(a[base + t], brow[(size_t)nearest_src(x, sw, Wb) * CG + cg]);
  }
}

// grad_b[n, yb, xb, :] = sum of grad_out over the output pixels whose nearest source is (yb, xb) (grad_a is grad_out itself).
// The pre-image of a source row / column is a contiguous range; found with the forward's own index function.
__device__ __forceinline__ void preimage(int src, float scale, int in, int out, int& lo, int& hi) {
  int d = (int)((float)src / scale);
  d = d < out ? d : out;
  while (d > 0 && nearest_src(d - 1, scale, in) >= src) --d;
  while (d < out && nearest_src(d, scale, in) < src) ++d;
  lo = d;
  while (d < out && nearest_src(d, scale, in) == src) ++d;
  hi = d;
}

template <bool BF>
__global__ void __launch_bounds__(256)
    upsample_add_bwd_kernel(const void* __restrict__ g_, int Ha, int Wa, int Hb, int Wb, int CG, float sh, float sw,
                            void* __restrict__ gb_) {
  const int row = blockIdx.y;                       // n * Hb + yb
  const int n = row / Hb, yb = row - n * Hb;
  int y0, y1;
  preimage(yb, sh, Hb, Ha, y0, y1);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Wb * CG; t += gridDim.x * blockDim.x) {
    const int xb = t / CG, cg = t - xb * CG;
    int x0, x1;
    preimage(xb, sw, Wb, Wa, x0, x1);
    if (BF) {
      const BF8* g = reinterpret_cast<const BF8*>(g_);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
          const BF8 v = g[(((size_t)n * Ha + y) * Wa + x) * CG + cg];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] += bf2f(v.v[k]);
        }
      BF8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o.v[k] = f2bf(acc[k]);
      reinterpret_cast<BF8*>(gb_)[(size_t)row * Wb * CG + t] = o;
    } else {
      const float4* g = reinterpret_cast<const float4*>(g_);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) acc = vadd(acc, g[(((size_t)n * Ha + y) * Wa + x) * CG + cg]);
      reinterpret_cast<float4*>(gb_)[(size_t)row * Wb * CG + t] = acc;
    }
  }
}

}  // namespace pt

static int affine_check(const char* fn, int64_t n, int C, int64_t inner) {
  PT_REQUIRE(n > 0 && C > 0 && inner > 0 && n % 4 == 0, PT_EINVAL, "%s: bad size (n must be a multiple of 4)", fn);
  PT_REQUIRE((inner == 1 && C % 4 == 0) || (inner > 1 && inner % 4 == 0), PT_EINVAL,
             "%s: need C %% 4 == 0 (channels_last) or H*W %% 4 == 0 (NCHW)", fn);
  return PT_OK;
}

extern "C" int pt_affine_relu_fwd(const float* x, const float* scale, const float* shift, const float* residual,
                                  int64_t n, int C, int64_t inner, int relu, float* y, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(x && scale && shift && y, PT_EINVAL, "pt_affine_relu_fwd: NULL pointer");
  int rc = affine_check("pt_affine_relu_fwd", n, C, inner);
  if (rc) return rc;
  if (inner == 1)
    hipLaunchKernelGGL(affine_relu_fwd_kernel<true>, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), x, scale,
                       shift, residual, (long)(n / 4), C, (long)inner, relu, y);
  else
    hipLaunchKernelGGL(affine_relu_fwd_kernel<false>, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), x,
                       scale, shift, residual, (long)(n / 4), C, (long)inner, relu, y);
  PT_LAUNCH_CHECK("pt_affine_relu_fwd");
  return PT_OK;
}

// Frozen stem: max_pool2d(relu(x * scale + shift), 3, stride 2, pad 1) on a channels_last map in ONE pass (resnet.py:633-640:
// norm1 -> relu -> maxpool of a stem no gradient reaches).  One thread per (output pixel, 4 channels): nine float4 taps (the window's
// rows are contiguous 3 * C floats; neighbouring windows share taps through L2), fp32 affine + ReLU per tap exactly as the separate
// passes compute them, maximum, one float4 store.  A padded tap never wins: every window holds its centre pixel and ReLU >= 0.
__global__ void __launch_bounds__(256)
    affine_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int B, int H,
                               int W, int C, int Ho, int Wo, float* __restrict__ y) {
  const int c4n = C >> 2;
  const long total = (long)B * Ho * Wo * c4n;
  for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(u % c4n);
    long r = u / c4n;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    const float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = 2 * oy + dy;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = 2 * ox + dx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = reinterpret_cast<const float4*>(x + (((long)b * H + iy) * W + ix) * C)[c4];
        // (torch's relu and max_pool2d propagate NaN; fmaxf would drop it - round-4 advice: a NaN maximum stays, a NaN tap wins)
        const float tx = v.x * sc.x + sh.x, ty = v.y * sc.y + sh.y, tz = v.z * sc.z + sh.z, tw = v.w * sc.w + sh.w;
        m.x = (m.x != m.x || tx <= m.x) ? m.x : tx; m.y = (m.y != m.y || ty <= m.y) ? m.y : ty;
        m.z = (m.z != m.z || tz <= m.z) ? m.z : tz; m.w = (m.w != m.w || tw <= m.w) ? m.w : tw;
      }
    }
    reinterpret_cast<float4*>(y)[u] = m;
  }
}

extern "C" int pt_affine_relu_maxpool_fwd(const float* x, const float* scale, const float* shift, int B, int H, int W, int C, float* y,
                                          void* stream) {
  PT_REQUIRE(x && scale && shift && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, PT_EINVAL,
             "pt_affine_relu_maxpool_fwd: NULL pointer or C not a multiple of 4");
  PT_REQUIRE(((((uintptr_t)x) | ((uintptr_t)scale) | ((uintptr_t)shift) | ((uintptr_t)y)) & 15) == 0, PT_EINVAL,
             "pt_affine_relu_maxpool_fwd: buffers must be 16-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;          // floor((H + 2 - 3) / 2) + 1
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(affine_relu_maxpool_kernel, dim3(stream_blocks(total)), dim3(256), 0, as_stream(stream), x, scale, shift, B, H, W, C,
                     Ho, Wo, y);
  PT_LAUNCH_CHECK("pt_affine_relu_maxpool_fwd");
  return PT_OK;
}

extern "C" int pt_affine_relu_bwd(const float* grad_y, const float* y, const float* scale, int64_t n, int C,
                                  int64_t inner, int relu, float* grad_x, float* grad_res, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(grad_y && scale && (grad_x || grad_res) && (!relu || y), PT_EINVAL, "pt_affine_relu_bwd: NULL pointer");
  int rc = affine_check("pt_affine_relu_bwd", n, C, inner);
  if (rc) return rc;
  if (inner == 1)
    hipLaunchKernelGGL(affine_relu_bwd_kernel<true>, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), grad_y,
                       y, scale, (long)(n / 4), C, (long)inner, relu, grad_x, grad_res);
  else
    hipLaunchKernelGGL(affine_relu_bwd_kernel<false>, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), grad_y,
                       y, scale, (long)(n / 4), C, (long)inner, relu, grad_x, grad_res);
  PT_LAUNCH_CHECK("pt_affine_relu_bwd");
  return PT_OK;
}

static int affine_train_blocks(int64_t n, int C) {
  const int G = C / 4;
  int nb = stream_blocks(n / 4);
  if (G > AFF_THREADS) {               // the launch stride (blocks * 256 float4) must be a multiple of G
    const int m = G / AFF_THREADS;
    nb = nb / m * m;
    if (nb < m) nb = m;
  }
  return nb;
}

extern "C" int pt_affine_train_rows(int64_t n, int C) {
  if (n <= 0 || C <= 0 || C % 4) return 0;
  const int G = C / 4;
  return affine_train_blocks(n, C) / (G > AFF_THREADS ? G / AFF_THREADS : 1);
}

extern "C" int pt_affine_relu_bwd_train(const float* grad_y, const float* y, const float* x, const float* scale,
                                        int64_t n, int C, int relu, float* grad_x, float* grad_res, float* sums,
                                        float* partial_ws, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(grad_y && scale && sums && partial_ws && (!relu || y), PT_EINVAL,
             "pt_affine_relu_bwd_train: NULL pointer");
  PT_REQUIRE(n > 0 && C > 0 && C % 4 == 0 && n % C == 0, PT_EINVAL, "pt_affine_relu_bwd_train: bad size");
  const int G = C / 4;
  PT_REQUIRE((G <= AFF_THREADS && AFF_THREADS % G == 0) || G % AFF_THREADS == 0, PT_ELIMIT,
             "pt_affine_relu_bwd_train: C/4=%d must divide or be a multiple of %d", G, AFF_THREADS);
  const int nb = affine_train_blocks(n, C);
  hipStream_t s = as_stream(stream);
  if (x)
    hipLaunchKernelGGL(affine_relu_bwd_train_kernel<true>, dim3(nb), dim3(AFF_THREADS), 0, s, grad_y, y, x, scale,
                       (long)(n / 4), C, relu, grad_x, grad_res, partial_ws);
  else                                 // sums[C..2C) come out as zeros
    hipLaunchKernelGGL(affine_relu_bwd_train_kernel<false>, dim3(nb), dim3(AFF_THREADS), 0, s, grad_y, y, x, scale,
                       (long)(n / 4), C, relu, grad_x, grad_res, partial_ws);
  PT_LAUNCH_CHECK("pt_affine_relu_bwd_train");
  hipLaunchKernelGGL(affine_train_finish, dim3(cdiv(2 * C, 16)), dim3(256), 0, s, partial_ws,
                     pt_affine_train_rows(n, C), 2 * C, sums);
  PT_LAUNCH_CHECK("pt_affine_relu_bwd_train(finish)");
  return PT_OK;
}

extern "C" int pt_affine_relu_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, const uint16_t* residual,
                                       int64_t n, int C, int relu, uint16_t* y, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(x && scale && shift && y, PT_EINVAL, "pt_affine_relu_fwd_bf16: NULL pointer");
  PT_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && n % C == 0, PT_EINVAL, "pt_affine_relu_fwd_bf16: need channels_last with C %% 8 == 0");
  hipLaunchKernelGGL(affine_relu_fwd_bf16_kernel, dim3(stream_blocks(n / 8)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const BF8*>(x), scale, shift, reinterpret_cast<const BF8*>(residual), (long)(n / 8), C,
                     relu, reinterpret_cast<BF8*>(y));
  PT_LAUNCH_CHECK("pt_affine_relu_fwd_bf16");
  return PT_OK;
}

extern "C" int pt_affine_relu_bwd_bf16(const uint16_t* grad_y, const uint16_t* y, const float* scale, int64_t n, int C,
                                       int relu, uint16_t* grad_x, uint16_t* grad_res, void* stream) {
  if (n == 0) return PT_OK;
  PT_REQUIRE(grad_y && scale && (grad_x || grad_res) && (!relu || y), PT_EINVAL, "pt_affine_relu_bwd_bf16: NULL pointer");
  PT_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && n % C == 0, PT_EINVAL, "pt_affine_relu_bwd_bf16: need channels_last with C %% 8 == 0");
  hipLaunchKernelGGL(affine_relu_bwd_bf16_kernel, dim3(stream_blocks(n / 8)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const BF8*>(grad_y), reinterpret_cast<const BF8*>(y), scale, (long)(n / 8), C, relu,
                     reinterpret_cast<BF8*>(grad_x), reinterpret_cast<BF8*>(grad_res));
  PT_LAUNCH_CHECK("pt_affine_relu_bwd_bf16");
  return PT_OK;
}

static int upsample_check(const char* fn, int N, int Ha, int Wa, int Hb, int Wb, int C, int bf16) {
  PT_REQUIRE(N > 0 && Ha > 0 && Wa > 0 && Hb > 0 && Wb > 0 && C > 0 && C % (bf16 ? 8 : 4) == 0, PT_EINVAL,
             "%s: bad shape (channels_last, C %% %d == 0)", fn, bf16 ? 8 : 4);
  PT_REQUIRE((long)N * Ha <= 65535 && (long)N * Hb <= 65535, PT_ELIMIT, "%s: N*H = %ld rows exceed the grid limit 65535", fn,
             (long)N * (Ha > Hb ? Ha : Hb));
  return PT_OK;
}

extern "C" int pt_upsample_add_fwd(const void* a, const void* b, int N, int Ha, int Wa, int Hb, int Wb, int C, int bf16,
                                   void* out, void* stream) {
  PT_REQUIRE(a && b && out, PT_EINVAL, "pt_upsample_add_fwd: NULL pointer");
  int rc = upsample_check("pt_upsample_add_fwd", N, Ha, Wa, Hb, Wb, C, bf16);
  if (rc != PT_OK) return rc;
  const int CG = C / (bf16 ? 8 : 4);
  const float sh = (float)Hb / (float)Ha, sw = (float)Wb / (float)Wa;
  const dim3 grid(cdiv(Wa * CG, 256), N * Ha);
  if (bf16)
    hipLaunchKernelGGL(upsample_add_fwd_kernel<BF8>, grid, dim3(256), 0, as_stream(stream), (const BF8*)a, (const BF8*)b, Ha, Wa,
                       Hb, Wb, CG, sh, sw, (BF8*)out);
  else
    hipLaunchKernelGGL(upsample_add_fwd_kernel<float4>, grid, dim3(256), 0, as_stream(stream), (const float4*)a,
                       (const float4*)b, Ha, Wa, Hb, Wb, CG, sh, sw, (float4*)out);
  PT_LAUNCH_CHECK("pt_upsample_add_fwd");
  return PT_OK;
}

extern "C" int pt_upsample_add_bwd(const void* grad_out, int N, int Ha, int Wa, int Hb, int Wb, int C, int bf16, void* grad_b,
                                   void* stream) {
  PT_REQUIRE(grad_out && grad_b, PT_EINVAL, "pt_upsample_add_bwd: NULL pointer");
  int rc = upsample_check("pt_upsample_add_bwd", N, Ha, Wa, Hb, Wb, C, bf16);
  if (rc != PT_OK) return rc;
  const int CG = C / (bf16 ? 8 : 4);
  const float sh = (float)Hb / (float)Ha, sw = (float)Wb / (float)Wa;
  const dim3 grid(cdiv(Wb * CG, 256), N * Hb);
  if (bf16)
    hipLaunchKernelGGL(upsample_add_bwd_kernel<true>, grid, dim3(256), 0, as_stream(stream), grad_out, Ha, Wa, Hb, Wb, CG, sh, sw,
                       grad_b);
  else
    hipLaunchKernelGGL(upsample_add_bwd_kernel<false>, grid, dim3(256), 0, as_stream(stream), grad_out, Ha, Wa, Hb, Wb, CG, sh,
                       sw, grad_b);
  PT_LAUNCH_CHECK("pt_upsample_add_bwd");
  return PT_OK;
}
