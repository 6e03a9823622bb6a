// MIL proposal bags for gfx950: bag construction, negative sampling, bag scoring
// (loss fwd/bwd) and bag selection.  One wavefront per bag: the U2 instances of a bag sit
// on the 64 lanes (strided when U2 > 64), softmax / L1-norm / weighted sums over the bag are
// wavefront shuffles, nothing is materialised in HBM (the reference runs ~15 small torch
// kernels with [N,U1,U2,C] temporaries per call).
#include "pt_common.h"

namespace pt {

// ---------------------------------------------------------- fine_proposals ---
// detectors/syn_images_generator_v2.py:270-322 (gen_proposal_mode 'fix_gen', cut_mode None)
constexpr int MAX_RATIOS = 8, MAX_SHAKE = 4;
struct FineCfg {
  float ratios[MAX_RATIOS];
  float shake[MAX_SHAKE];
  int nr, ns;
};

__global__ void fine_proposals_kernel(const float4* __restrict__ boxes, long total, FineCfg cfg, float min_scale,
                                      float img_h, float img_w, float4* __restrict__ props,
                                      uint8_t* __restrict__ valid) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= total) return;
  const int S1 = 1 + 4 * cfg.ns, RR = cfg.nr * cfg.nr;
  const int s = (int)(o % S1);
  const int rr = (int)((o / S1) % RR);
  const long g = o / ((long)S1 * RR);
  const float rw = cfg.ratios[rr / cfg.nr], rh = cfg.ratios[rr % cfg.nr];
  const float4 b = boxes[g];
  const float cx = (b.x + b.z) / 2, cy = (b.y + b.w) / 2;
  float w = fminf(fmaxf(b.z - b.x, min_scale), 1000.f), h = fminf(fmaxf(b.w - b.y, min_scale), 1000.f);
  w *= rw;
  h *= rh;
  float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h, x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
  if (s > 0) {
    const int m = (s - 1) / 4, dir = (s - 1) % 4;
    const float r = cfg.shake[m];
    // centre / size are re-derived from the xyxy box exactly as bbox_xyxy_to_cxcywh does (:290)
    float pcx = (x1 + x2) / 2, pcy = (y1 + y2) / 2;
    const float pw = x2 - x1, ph = y2 - y1;
    if (dir == 0) pcx = pcx - r * pw;
    else if (dir == 1) pcx = pcx + r * pw;
    else if (dir == 2) pcy = pcy - r * ph;
    else pcy = pcy + r * ph;
    x1 = pcx - 0.5f * pw; y1 = pcy - 0.5f * ph; x2 = pcx + 0.5f * pw; y2 = pcy + 0.5f * ph;
  }
  props[o] = make_float4(x1, y1, x2, y2);
  // IoF with the image box > 0.7 (:317-319), bbox_overlaps mode 'iof', eps 1e-6
  const float area = (x2 - x1) * (y2 - y1);
  const float ow = fmaxf(fminf(x2, img_w) - fmaxf(x1, 0.f), 0.f);
  const float oh = fmaxf(fminf(y2, img_h) - fmaxf(y1, 0.f), 0.f);
  valid[o] = (ow * oh) / fmaxf(area, 1e-6f) > 0.7f ? 1 : 0;
}

// syn_images_generator_v2.py:247-255
__global__ void negative_proposals_kernel(const float* __restrict__ u, int B, int n, const float4* __restrict__ pos,
                                          const int32_t* __restrict__ pos_off, float img_h, float img_w, float thr,
                                          float4* __restrict__ neg, uint8_t* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, j = i - b * n;
  const float* ub = u + (size_t)b * 4 * n;
  const float x1 = ub[j] * img_w * 0.8f;
  const float y1 = ub[n + j] * img_h * 0.8f;
  const float x2 = x1 + ub[2 * n + j] * 100.f;
  const float y2 = y1 + ub[3 * n + j] * 100.f;
  neg[i] = make_float4(x1, y1, x2, y2);
  const float a1 = (x2 - x1) * (y2 - y1);
  bool good = true;
  for (int p = pos_off[b]; p < pos_off[b + 1]; ++p) {
    const float4 q = pos[p];
    const float w = fmaxf(fminf(x2, q.z) - fmaxf(x1, q.x), 0.f);
    const float h = fmaxf(fminf(y2, q.w) - fmaxf(y1, q.y), 0.f);
    const float ov = w * h;
    const float iou = ov / fmaxf(a1 + (q.z - q.x) * (q.w - q.y) - ov, 1e-6f);
    good = good && (iou < thr);
  }
  ok[i] = good ? 1 : 0;
}

// ------------------------------------------------------------ bag scoring ----
// normalised instance score of one (bag, class): softmax over the bag, masked by valid,
// L1-normalised (F.normalize eps 1e-12).  Lane-strided over U2.  Returns per-lane-loop values
// through callbacks to keep everything in registers is overkill here - recompute instead.
struct BagStats {
  float mx, se, denom;
};

__device__ __forceinline__ BagStats bag_stats(const float* __restrict__ ins, const uint8_t* __restrict__ valid, int U2,
                                              int C, int c, int lane) {
  float m = -INFINITY;
  for (int u = lane; u < U2; u += 64) m = fmaxf(m, ins[(size_t)u * C + c]);
  m = wave_max(m);
  float se = 0.f;
  for (int u = lane; u < U2; u += 64) se += expf(ins[(size_t)u * C + c] - m);
  se = wave_sum(se);
  float l1 = 0.f;
  for (int u = lane; u < U2; u += 64) l1 += valid[u] ? expf(ins[(size_t)u * C + c] - m) / se : 0.f;
  l1 = wave_sum(l1);
  BagStats s;
  s.mx = m; s.se = se; s.denom = fmaxf(l1, 1e-12f);
  return s;
}

__device__ __forceinline__ float gfocal_term(float p, float q, float w, float eps, float* dp) {
  // fcos_head_p2b_ts.py:1074-1078 (one class)
  const float l1 = (p - q) * (p - q);
  const float l2 = q * logf(p + eps) + (1.f - q) * logf(1.f - p + eps);
  if (dp) *dp = -w * (2.f * (p - q) * l2 + l1 * (q / (p + eps) - (1.f - q) / (1.f - p + eps)));
  return -(l1 * l2 * w);
}

template <bool BWD>
__global__ void __launch_bounds__(64)
    mil_bag_loss_kernel(const float* __restrict__ cls, const float* __restrict__ ins,
                        const uint8_t* __restrict__ valid, const int32_t* __restrict__ labels,
                        const float* __restrict__ scale, int U2, int C, float* __restrict__ bag_loss,
                        uint8_t* __restrict__ bag_valid, float* __restrict__ gcls, float* __restrict__ gins) {
  const int nb = blockIdx.x, lane = threadIdx.x;
  const float* cb = cls + (size_t)nb * U2 * C;
  const float* ib = ins + (size_t)nb * U2 * C;
  const uint8_t* vb = valid + (size_t)nb * U2;
  int anyv = 0;
  for (int u = lane; u < U2; u += 64) anyv |= vb[u];
  anyv = __any(anyv);
  const float w = anyv ? 1.f : 0.f;
  const int lab = labels[nb];
  float loss = 0.f;
  const float sc = BWD ? scale[0] : 0.f;
  for (int c = 0; c < C; ++c) {
    const BagStats st = bag_stats(ib, vb, U2, C, c, lane);
    float bag = 0.f;
    for (int u = lane; u < U2; u += 64) {
      const float soft = expf(ib[(size_t)u * C + c] - st.mx) / st.se;
      const float n = (vb[u] ? soft : 0.f) / st.denom;
      bag += sigmoidf_(cb[(size_t)u * C + c]) * n;
    }
    bag = wave_sum(bag);
    const float q = (lab == c) ? 1.f : 0.f;
    float dp;
    loss += gfocal_term(bag, q, w, 1e-6f, BWD ? &dp : nullptr);
    if (BWD) {
      const float gc = dp * sc;  // d L / d bag[c]
      // d bag / d n_u = s_u ; n = sv / D
      float dot = 0.f;  // sum_u gn_u * n_u
      for (int u = lane; u < U2; u += 64) {
        const float soft = expf(ib[(size_t)u * C + c] - st.mx) / st.se;
        const float n = (vb[u] ? soft : 0.f) / st.denom;
        dot += gc * sigmoidf_(cb[(size_t)u * C + c]) * n;
      }
      dot = wave_sum(dot);
      const bool clamped = !(st.denom > 1e-12f);
      float dot2 = 0.f;  // sum_j g_soft_j * soft_j
      for (int u = lane; u < U2; u += 64) {
        const float soft = expf(ib[(size_t)u * C + c] - st.mx) / st.se;
        const float s = sigmoidf_(cb[(size_t)u * C + c]);
        const float gn = gc * s;
        const float gsv = clamped ? gn / st.denom : (gn - dot) / st.denom;
        const float gsoft = vb[u] ? gsv : 0.f;
        dot2 += gsoft * soft;
      }
      dot2 = wave_sum(dot2);
      for (int u = lane; u < U2; u += 64) {
        const float soft = expf(ib[(size_t)u * C + c] - st.mx) / st.se;
        const float s = sigmoidf_(cb[(size_t)u * C + c]);
        const float n = (vb[u] ? soft : 0.f) / st.denom;
        const float gn = gc * s;
        const float gsv = clamped ? gn / st.denom : (gn - dot) / st.denom;
        const float gsoft = vb[u] ? gsv : 0.f;
        gcls[((size_t)nb * U2 + u) * C + c] = gc * n * s * (1.f - s);
        gins[((size_t)nb * U2 + u) * C + c] = soft * (gsoft - dot2);
      }
    }
  }
  if (!BWD && lane == 0) {
    bag_loss[nb] = loss;
    bag_valid[nb] = anyv ? 1 : 0;
  }
}

template <bool BWD>
__global__ void mil_neg_loss_kernel(const float* __restrict__ x, const uint8_t* __restrict__ wv,
                                    const float* __restrict__ scale, int M, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float w = wv[i] ? 1.f : 0.f;
  const float sc = BWD ? scale[0] : 0.f;
  float loss = 0.f;
  for (int c = 0; c < C; ++c) {
    const float p = sigmoidf_(x[(size_t)i * C + c]);
    float dp;
    loss += gfocal_term(p, 0.f, w, 1e-6f, BWD ? &dp : nullptr);
    if (BWD) out[(size_t)i * C + c] = sc * dp * p * (1.f - p);
  }
  if (!BWD) out[i] = loss;
}

// ----------------------------------------------------------- bag selection ---
constexpr int SEL_MAXU = 4096;

// D = 4: xyxy bags, x clamped to [0,w] and y to [0,h] (fcos_head_p2b_ts.py:1092-1110).
// D = 5: (cx,cy,w,h,a) bags; the OBB head clamps columns 0 AND 1 first to [0,w] and then to [0,h]
//        (rotated_fcos_head_p2rb_ts.py:1211-1212) and leaves w, h, a alone.
template <int D>
__global__ void __launch_bounds__(64)
    mil_bag_select_kernel(const float* __restrict__ cls, const float* __restrict__ ins,
                          const uint8_t* __restrict__ valid, const int32_t* __restrict__ labels,
                          const float* __restrict__ bags, const float* __restrict__ pseudo, int U1, int U2, int C,
                          int topk, float beta, float img_h, float img_w, float* __restrict__ merged) {
  __shared__ float score[SEL_MAXU];
  const int g = blockIdx.x, lane = threadIdx.x;
  const int U = U1 * U2;
  const int lab = labels[g];
  const float* cb = cls + (size_t)g * U * C;
  const float* ib = ins + (size_t)g * U * C;
  const uint8_t* vb = valid + (size_t)g * U;
  for (int a = 0; a < U1; ++a) {
    const BagStats st = bag_stats(ib + (size_t)a * U2 * C, vb + (size_t)a * U2, U2, C, lab, lane);
    for (int u = lane; u < U2; u += 64) {
      const int uu = a * U2 + u;
      const float soft = expf(ib[(size_t)uu * C + lab] - st.mx) / st.se;
      const float n = (vb[uu] ? soft : 0.f) / st.denom;
      score[uu] = sigmoidf_(cb[(size_t)uu * C + lab]) * n;
    }
  }
  __syncthreads();
  float sw = 0.f;
  float sc_k[8];
  int id_k[8];
  unsigned long long last = 0ull;
  bool first = true;
  for (int r = 0; r < topk; ++r) {
    unsigned long long best = ~0ull;
    for (int u = lane; u < U; u += 64) {
      // descending score, ascending index: key = (~bits(score), idx); scores are >= 0
      const unsigned long long key = ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(score[u])) << 32) | (unsigned)u;
      if ((first || key > last) && key < best) best = key;
    }
    best = wave_min_u64(best);
    last = best;
    first = false;
    const int idx = (int)(best & 0xffffffffu);
    id_k[r] = idx;
    sc_k[r] = (best == ~0ull) ? 0.f : score[idx];
    sw += sc_k[r];
  }
  if (lane == 0) {
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int r = 0; r < topk; ++r) {
      const float wgt = sc_k[r] / (sw + 1e-8f);
      const float* q = bags + ((size_t)g * U + id_k[r]) * D;
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] += q[c] * wgt;
    }
    if (D == 4) {
      acc[0] = fminf(fmaxf(acc[0], 0.f), img_w); acc[2] = fminf(fmaxf(acc[2], 0.f), img_w);
      acc[1] = fminf(fmaxf(acc[1], 0.f), img_h); acc[3] = fminf(fmaxf(acc[3], 0.f), img_h);
    } else {
      acc[0] = fminf(fmaxf(fminf(fmaxf(acc[0], 0.f), img_w), 0.f), img_h);
      acc[1] = fminf(fmaxf(fminf(fmaxf(acc[1], 0.f), img_w), 0.f), img_h);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) merged[(size_t)g * D + c] = (1.f - beta) * acc[c] + beta * pseudo[(size_t)g * D + c];
  }
}

}  // namespace pt

using namespace pt;

extern "C" int pt_fine_proposals(const float* boxes, int sumG, const float* ratios, int n_ratios, const float* shake,
                                 int n_shake, float min_scale, float img_h, float img_w, float* props, uint8_t* valid,
                                 void* stream) {
  if (sumG == 0) return PT_OK;
  PT_REQUIRE(boxes && ratios && props && valid && sumG > 0, PT_EINVAL, "pt_fine_proposals: bad argument");
  PT_REQUIRE(n_ratios >= 1 && n_ratios <= MAX_RATIOS && n_shake >= 0 && n_shake <= MAX_SHAKE, PT_ELIMIT,
             "pt_fine_proposals: n_ratios=%d (max %d) n_shake=%d (max %d)", n_ratios, MAX_RATIOS, n_shake, MAX_SHAKE);
  FineCfg cfg;
  cfg.nr = n_ratios; cfg.ns = n_shake;
  for (int i = 0; i < n_ratios; ++i) cfg.ratios[i] = ratios[i];
  for (int i = 0; i < n_shake; ++i) cfg.shake[i] = shake[i];
  const long total = (long)sumG * n_ratios * n_ratios * (1 + 4 * n_shake);
  hipLaunchKernelGGL(fine_proposals_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(boxes), total, cfg, min_scale, img_h, img_w,
                     reinterpret_cast<float4*>(props), valid);
  PT_LAUNCH_CHECK("pt_fine_proposals");
  return PT_OK;
}

extern "C" int pt_negative_proposals(const float* u, int B, int n, const float* pos, const int32_t* pos_off,
                                     float img_h, float img_w, float iou_thr, float* neg, uint8_t* neg_ok,
                                     void* stream) {
  if (B * n == 0) return PT_OK;
  PT_REQUIRE(u && pos_off && neg && neg_ok && B > 0 && n > 0, PT_EINVAL, "pt_negative_proposals: bad argument");
  hipLaunchKernelGGL(negative_proposals_kernel, dim3(cdiv(B * n, 128)), dim3(128), 0, as_stream(stream), u, B, n,
                     reinterpret_cast<const float4*>(pos), pos_off, img_h, img_w, iou_thr,
                     reinterpret_cast<float4*>(neg), neg_ok);
  PT_LAUNCH_CHECK("pt_negative_proposals");
  return PT_OK;
}

extern "C" int pt_mil_bag_loss_fwd(const float* cls, const float* ins, const uint8_t* valid, const int32_t* labels,
                                   int NB, int U2, int C, float* bag_loss, uint8_t* bag_valid, void* stream) {
  if (NB == 0) return PT_OK;
  PT_REQUIRE(cls && ins && valid && labels && bag_loss && bag_valid && NB > 0 && U2 > 0 && C > 0, PT_EINVAL,
             "pt_mil_bag_loss_fwd: bad argument");
  hipLaunchKernelGGL(mil_bag_loss_kernel<false>, dim3(NB), dim3(64), 0, as_stream(stream), cls, ins, valid, labels,
                     (const float*)nullptr, U2, C, bag_loss, bag_valid, (float*)nullptr, (float*)nullptr);
  PT_LAUNCH_CHECK("pt_mil_bag_loss_fwd");
  return PT_OK;
}

extern "C" int pt_mil_bag_loss_bwd(const float* cls, const float* ins, const uint8_t* valid, const int32_t* labels,
                                   const float* scale, int NB, int U2, int C, float* grad_cls, float* grad_ins,
                                   void* stream) {
  if (NB == 0) return PT_OK;
  PT_REQUIRE(cls && ins && valid && labels && scale && grad_cls && grad_ins && NB > 0 && U2 > 0 && C > 0, PT_EINVAL,
             "pt_mil_bag_loss_bwd: bad argument");
  hipLaunchKernelGGL(mil_bag_loss_kernel<true>, dim3(NB), dim3(64), 0, as_stream(stream), cls, ins, valid, labels,
                     scale, U2, C, (float*)nullptr, (uint8_t*)nullptr, grad_cls, grad_ins);
  PT_LAUNCH_CHECK("pt_mil_bag_loss_bwd");
  return PT_OK;
}

extern "C" int pt_mil_neg_loss_fwd(const float* neg_cls, const uint8_t* neg_w, int M, int C, float* loss,
                                   void* stream) {
  if (M == 0) return PT_OK;
  PT_REQUIRE(neg_cls && neg_w && loss && M > 0 && C > 0, PT_EINVAL, "pt_mil_neg_loss_fwd: bad argument");
  hipLaunchKernelGGL(mil_neg_loss_kernel<false>, dim3(cdiv(M, 128)), dim3(128), 0, as_stream(stream), neg_cls, neg_w,
                     (const float*)nullptr, M, C, loss);
  PT_LAUNCH_CHECK("pt_mil_neg_loss_fwd");
  return PT_OK;
}

extern "C" int pt_mil_neg_loss_bwd(const float* neg_cls, const uint8_t* neg_w, const float* scale, int M, int C,
                                   float* grad, void* stream) {
  if (M == 0) return PT_OK;
  PT_REQUIRE(neg_cls && neg_w && scale && grad && M > 0 && C > 0, PT_EINVAL, "pt_mil_neg_loss_bwd: bad argument");
  hipLaunchKernelGGL(mil_neg_loss_kernel<true>, dim3(cdiv(M, 128)), dim3(128), 0, as_stream(stream), neg_cls, neg_w,
                     scale, M, C, grad);
  PT_LAUNCH_CHECK("pt_mil_neg_loss_bwd");
  return PT_OK;
}

extern "C" int pt_mil_bag_select(const float* cls, const float* ins, const uint8_t* valid, const int32_t* labels,
                                 const float* bags, const float* pseudo, int NG, int U1, int U2, int C, int topk,
                                 float beta, float img_h, float img_w, float* merged, void* stream) {
  if (NG == 0) return PT_OK;
  PT_REQUIRE(cls && ins && valid && labels && bags && pseudo && merged && NG > 0 && U1 > 0 && U2 > 0 && C > 0,
             PT_EINVAL, "pt_mil_bag_select: bad argument");
  PT_REQUIRE(topk >= 1 && topk <= 8 && topk <= U1 * U2, PT_ELIMIT, "pt_mil_bag_select: topk=%d outside [1,8]", topk);
  PT_REQUIRE(U1 * U2 <= SEL_MAXU, PT_ELIMIT, "pt_mil_bag_select: U1*U2=%d above %d", U1 * U2, SEL_MAXU);
  hipLaunchKernelGGL(mil_bag_select_kernel<4>, dim3(NG), dim3(64), 0, as_stream(stream), cls, ins, valid, labels, bags,
                     pseudo, U1, U2, C, topk, beta, img_h, img_w, merged);
  PT_LAUNCH_CHECK("pt_mil_bag_select");
  return PT_OK;
}

extern "C" int pt_mil_bag_select_obb(const float* cls, const float* ins, const uint8_t* valid, const int32_t* labels,
                                     const float* bags, const float* pseudo, int NG, int U1, int U2, int C, int topk,
                                     float beta, float img_h, float img_w, float* merged, void* stream) {
  if (NG == 0) return PT_OK;
  PT_REQUIRE(cls && ins && valid && labels && bags && pseudo && merged && NG > 0 && U1 > 0 && U2 > 0 && C > 0,
             PT_EINVAL, "pt_mil_bag_select_obb: bad argument");
  PT_REQUIRE(topk >= 1 && topk <= 8 && topk <= U1 * U2, PT_ELIMIT, "pt_mil_bag_select_obb: topk=%d outside [1,8]", topk);
  PT_REQUIRE(U1 * U2 <= SEL_MAXU, PT_ELIMIT, "pt_mil_bag_select_obb: U1*U2=%d above %d", U1 * U2, SEL_MAXU);
  hipLaunchKernelGGL(mil_bag_select_kernel<5>, dim3(NG), dim3(64), 0, as_stream(stream), cls, ins, valid, labels, bags,
                     pseudo, U1, U2, C, topk, beta, img_h, img_w, merged);
  PT_LAUNCH_CHECK("pt_mil_bag_select_obb");
  return PT_OK;
}
