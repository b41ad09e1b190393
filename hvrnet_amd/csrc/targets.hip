// Training-target generation and the RPN loss for gfx950: the steps that turn ground-truth boxes into
// the sampled RoIs / anchor targets the head and RPN losses consume.  Built with -ffp-contract=off: the IoU and delta
// arithmetic follows the reference's operation order so that thresholds and "== row maximum" tests decide identically.
//   assign   mmdet/core/bbox/assigners/max_iou_assigner.py:48-173 over mmdet/core/bbox/geometry.py:46-60
//   sample   mmdet/core/bbox/samplers/base_sampler.py:32-78 with random_sampler.py:37-53 / ohem_hnl_sampler.py:56-113;
//            the randomness is an INPUT (one key per box, the `expected` smallest keys win), not a host-side shuffle
//   targets  mmdet/core/anchor/anchor_target.py:121-155, mmdet/core/bbox/bbox_target.py:35-62, transforms.py:6-31
//   rpn loss mmdet/models/anchor_heads/anchor_head.py:141-160 (sigmoid BCE + smooth-L1, both / num_total_samples)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <cstdint>

namespace hvr {
hipError_t run_zero_fill(void*, size_t, hipStream_t);   // misc.hip: a kernel, not a memset node (captured graphs)

namespace {

__device__ __forceinline__ float iou_plus1(const float4 g, const float4 b) {
  const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y), rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
  const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
  const float overlap = w * h;
  const float area1 = (g.z - g.x + 1.f) * (g.w - g.y + 1.f), area2 = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  return overlap / (area1 + area2 - overlap);
}

__device__ __forceinline__ float4 load_box(const float* boxes, int ldb, int i) {
  const float* p = boxes + (long)i * ldb;
  return make_float4(p[0], p[1], p[2], p[3]);
}

constexpr int ASSIGN_MAX_GT = 256;

}  // namespace

// per box: max / first argmax over the gts; per gt: max over the boxes (atomicMax on the bit pattern, IoU >= 0)
__global__ __launch_bounds__(256) void assign_max_kernel(const float* __restrict__ boxes, int ldb, int n,
                                                         const float* __restrict__ gts, int k, const uint8_t* __restrict__ valid,
                                                         float* __restrict__ max_ov, int* __restrict__ argmax,
                                                         unsigned* __restrict__ gt_max_bits) {
  __shared__ float4 sg[ASSIGN_MAX_GT];
  for (int g = threadIdx.x; g < k; g += 256) sg[g] = make_float4(gts[g * 4], gts[g * 4 + 1], gts[g * 4 + 2], gts[g * 4 + 3]);
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < n && (!valid || valid[i]);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) b = load_box(boxes, ldb, i);
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < k; ++g) {
    const float v = ok ? iou_plus1(sg[g], b) : 0.f;
    if (ok && v > best) { best = v; arg = g; }
    float m = v;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(gt_max_bits + g, __float_as_uint(m));
  }
  if (i < n) {
    max_ov[i] = ok ? best : -1.f;
    argmax[i] = arg;
  }
}

__global__ __launch_bounds__(256) void assign_final_kernel(const float* __restrict__ boxes, int ldb, int n,
                                                           const float* __restrict__ gts, int k, const uint8_t* __restrict__ valid,
                                                           const float* __restrict__ max_ov, const int* __restrict__ argmax,
                                                           const unsigned* __restrict__ gt_max_bits, float pos_thr, float neg_lo,
                                                           float neg_hi, float min_pos, long long* __restrict__ gt_inds) {
  __shared__ float4 sg[ASSIGN_MAX_GT];
  __shared__ float sm[ASSIGN_MAX_GT];
  for (int g = threadIdx.x; g < k; g += 256) {
    sg[g] = make_float4(gts[g * 4], gts[g * 4 + 1], gts[g * 4 + 2], gts[g * 4 + 3]);
    sm[g] = __uint_as_float(gt_max_bits[g]);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long a = -1;
  if (!valid || valid[i]) {
    const float v = max_ov[i];
    if (v >= neg_lo && v < neg_hi) a = 0;                 // step 2
    if (v >= pos_thr) a = argmax[i] + 1;                  // step 3
    const float4 b = load_box(boxes, ldb, i);
    for (int g = 0; g < k; ++g)                           // step 4, later gts override earlier ones (gt_max_assign_all)
      if (sm[g] >= min_pos && iou_plus1(sg[g], b) == sm[g]) a = g + 1;
  }
  gt_inds[i] = a;
}

// ---------------------------------------------------------------------------------------------------------------
// sampler: one workgroup.  Group "pos" = cls > 0, "neg" = cls == 0.  From each group the `expected` members with the
// smallest keys are taken (all of them when the group is not larger), ties by lower index; output in ascending index
// order, positives first -- SamplingResult.bboxes' order after the reference's `.unique()` (base_sampler.py:62-75).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ unsigned key_order(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SampleShared {
  int hist[256];
  int scan[1024];
  unsigned prefix, mask;
  int remaining, total, base;
};

// block scan of one int per thread (1024 threads = 16 waves): returns the exclusive prefix in thread order, *sum = block total.
// Wave-level inclusive scans on shuffles, the 16 wave totals scanned by wave 0: three barriers (round 6; the Hillis-Steele form over
// the LDS took twenty, and a sampling call makes six scans -- 45 us per frame for 304 boxes, nine frames per HVR iteration)
__device__ int block_exclusive_scan(int v, int* buf, int* sum) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) buf[w] = incl;
  __syncthreads();
  if (w == 0) {
    const int t = lane < 16 ? buf[lane] : 0;
    int sc = t;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int u = __shfl_up(sc, o, 64);
      if (lane >= o) sc += u;
    }
    if (lane < 16) buf[16 + lane] = sc - t;
    if (lane == 15) buf[32] = sc;
  }
  __syncthreads();
  const int base = buf[16 + w];
  *sum = buf[32];
  __syncthreads();  // (buf is the next scan's scratch)
  return base + incl - v;
}

__device__ int sample_group(const long long* cls, const float* keys, int n, bool want_pos, int expected, long long* out,
                            SampleShared& sh) {
  const int tid = threadIdx.x;
  const int seg = (n + 1023) / 1024, lo = min(tid * seg, n), hi = min(lo + seg, n);
  auto member = [&](int i) { return want_pos ? cls[i] > 0 : cls[i] == 0; };
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += member(i) ? 1 : 0;
  int total;
  block_exclusive_scan(cnt, sh.scan, &total);
  if (expected <= 0 || total == 0) return 0;
  unsigned thr = 0xffffffffu;
  int need_eq = 0x7fffffff;
  if (total > expected) {  // radix select of the expected-th smallest key among the members
    if (tid == 0) { sh.prefix = 0; sh.mask = 0; sh.remaining = expected; }
    for (int pass = 3; pass >= 0; --pass) {
      if (tid < 256) sh.hist[tid] = 0;
      __syncthreads();
      const unsigned prefix = sh.prefix, mask = sh.mask;
      for (int i = tid; i < n; i += 1024)
        if (member(i)) {
          const unsigned u = key_order(keys[i]);
          if ((u & mask) == prefix) atomicAdd(&sh.hist[(u >> (8 * pass)) & 255], 1);
        }
      __syncthreads();
      if (tid < 64) {
        // the bin that holds the rem-th smallest matching key: lane l owns bins 4 l .. 4 l + 3, a wave scan of the lanes' sums finds
        // the lane whose range contains it (the serial walk over 256 bins by one thread this replaces: up to 256 dependent LDS reads)
        const int h0 = sh.hist[4 * tid], h1 = sh.hist[4 * tid + 1], h2 = sh.hist[4 * tid + 2], h3 = sh.hist[4 * tid + 3];
        const int own = h0 + h1 + h2 + h3;
        int incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o, 64);
          if (tid >= o) incl += t;
        }
        const int rem0 = sh.remaining, excl = incl - own;
        if (excl < rem0 && rem0 <= incl) {
          int rem = rem0 - excl, b = 4 * tid;
          if (h0 < rem) { rem -= h0; ++b; if (h1 < rem) { rem -= h1; ++b; if (h2 < rem) { rem -= h2; ++b; } } }
          sh.remaining = rem;
          sh.prefix = prefix | ((unsigned)b << (8 * pass));
          sh.mask = mask | (255u << (8 * pass));
        }
      }
      __syncthreads();
    }
    thr = sh.prefix;
    need_eq = sh.remaining;
    __syncthreads();
  }
  // ordered compaction: keys below the threshold, plus the first need_eq members that equal it
  int eq = 0;
  for (int i = lo; i < hi; ++i) eq += (member(i) && key_order(keys[i]) == thr) ? 1 : 0;
  int dummy;
  int eq_off = block_exclusive_scan(eq, sh.scan, &dummy);
  int take = 0, e = eq_off;
  for (int i = lo; i < hi; ++i)
    if (member(i)) {
      const unsigned u = key_order(keys[i]);
      if (u < thr || total <= expected) ++take;
      else if (u == thr) { take += e < need_eq ? 1 : 0; ++e; }
    }
  int taken;
  int off = block_exclusive_scan(take, sh.scan, &taken);
  e = eq_off;
  for (int i = lo; i < hi; ++i)
    if (member(i)) {
      const unsigned u = key_order(keys[i]);
      bool t = u < thr || total <= expected;
      if (!t && u == thr) { t = e < need_eq; ++e; }
      if (t) out[off++] = i;
    }
  __syncthreads();
  return taken;
}

}  // namespace

__global__ __launch_bounds__(1024) void sample_kernel(const long long* __restrict__ cls, const float* __restrict__ keys, int n, int num,
                                                      int expected_pos, float neg_pos_ub, long long* __restrict__ inds,
                                                      int* __restrict__ counts) {
  __shared__ SampleShared sh;
  const int np = sample_group(cls, keys, n, true, expected_pos, inds, sh);
  int expected_neg = num - np;
  if (neg_pos_ub >= 0.f) {
    const int ub = (int)(neg_pos_ub * (float)max(1, np));
    expected_neg = min(expected_neg, ub);
  }
  const int nn = sample_group(cls, keys, n, false, expected_neg, inds + np, sh);
  if (threadIdx.x == 0) { counts[0] = np; counts[1] = nn; }
}

// ---------------------------------------------------------------------------------------------------------------
// targets of the sampled boxes.  scatter = 1: row inds[j] of [n]-row outputs (anchor_target after `unmap`);
// scatter = 0: row j (bbox_target: positives first).  Outputs are zero-filled by the caller.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_targets_kernel(const float* __restrict__ boxes, int ldb, const float* __restrict__ gts,
                                                          const long long* __restrict__ gt_labels,
                                                          const long long* __restrict__ gt_inds, const long long* __restrict__ inds,
                                                          const int* __restrict__ counts, float4 means, float4 stds, float pos_weight,
                                                          int scatter, long long* __restrict__ labels, float* __restrict__ label_w,
                                                          float* __restrict__ bbox_t, float* __restrict__ bbox_w) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int np = counts[0], nn = counts[1];
  if (j >= np + nn) return;
  const int i = (int)inds[j];
  const long row = scatter ? i : j;
  if (j >= np) {
    label_w[row] = 1.f;
    return;
  }
  const int g = (int)gt_inds[i] - 1;
  const float4 p = load_box(boxes, ldb, i), t = make_float4(gts[g * 4], gts[g * 4 + 1], gts[g * 4 + 2], gts[g * 4 + 3]);
  const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f, pw = p.z - p.x + 1.f, ph = p.w - p.y + 1.f;
  const float gx = (t.x + t.z) * 0.5f, gy = (t.y + t.w) * 0.5f, gw = t.z - t.x + 1.f, gh = t.w - t.y + 1.f;
  const float dx = (gx - px) / pw, dy = (gy - py) / ph, dw = logf(gw / pw), dh = logf(gh / ph);
  float* o = bbox_t + row * 4;
  o[0] = (dx - means.x) / stds.x;
  o[1] = (dy - means.y) / stds.y;
  o[2] = (dw - means.z) / stds.z;
  o[3] = (dh - means.w) / stds.w;
  float* w = bbox_w + row * 4;
  w[0] = w[1] = w[2] = w[3] = 1.f;
  labels[row] = gt_labels ? gt_labels[g] : 1;
  label_w[row] = pos_weight <= 0.f ? 1.f : pos_weight;
}

// ---------------------------------------------------------------------------------------------------------------
// RPN loss on the fused head output o [rows][ldo] (columns 0..A objectness logits, A..5A deltas, anchor a's at
// A + 4a): out2 = (loss_rpn_cls, loss_rpn_bbox), d_o = d(out2[0] + out2[1]) / d o.  One workgroup, fixed order.
// avg_factor = max(counts[0],1) + max(counts[1],1)  (anchor_target.py:66-67, anchor_head.py:191-192).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rpn_loss_kernel(const float* __restrict__ o, int ldo, int A, int rows,
                                                        const long long* __restrict__ labels, const float* __restrict__ label_w,
                                                        const float* __restrict__ bbox_t, const float* __restrict__ bbox_w,
                                                        const int* __restrict__ counts, float beta, float* __restrict__ out2,
                                                        float* __restrict__ d_o) {
  __shared__ float red[2][1024];
  const int tid = threadIdx.x;
  const float avg = (float)(max(counts[0], 1) + max(counts[1], 1));
  const int M = rows * A;
  float lc = 0.f, lb = 0.f;
  for (int m = tid; m < M; m += 1024) {
    const int r = m / A, a = m - r * A;
    const float* row = o + (long)r * ldo;
    float* drow = d_o + (long)r * ldo;
    const float x = row[a], z = (float)labels[m], w = label_w[m];
    // F.binary_cross_entropy_with_logits: max(x,0) - x z + log(1 + exp(-|x|))
    const float ex = expf(-fabsf(x));
    lc += (fmaxf(x, 0.f) - x * z + log1pf(ex)) * w;
    const float sig = x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
    drow[a] = (sig - z) * w / avg;
    for (int e = 0; e < 4; ++e) {
      const float d = row[A + a * 4 + e] - bbox_t[(long)m * 4 + e], ad = fabsf(d), bw = bbox_w[(long)m * 4 + e];
      lb += (ad < beta ? 0.5f * ad * ad / beta : ad - 0.5f * beta) * bw;
      drow[A + a * 4 + e] = bw / avg * (ad < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
  }
  red[0][tid] = lc; red[1][tid] = lb;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { out2[0] = red[0][0] / avg; out2[1] = red[1][0] / avg; }
}

// per-row softmax cross entropy (reduction 'none'): the OHEM ranking key (selsa_rcnn.py:209-218)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int ldl, int cls_off, int ncls,
                                                      const long long* __restrict__ labels, int R, float* __restrict__ loss) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float* row = logits + (long)r * ldl + cls_off;
  float mx = -INFINITY;
  for (int c = 0; c < ncls; ++c) mx = fmaxf(mx, row[c]);
  float se = 0.f;
  for (int c = 0; c < ncls; ++c) se += expf(row[c] - mx);
  loss[r] = mx + logf(se) - row[(int)labels[r]];
}

// ---------------------------------------------------------------------------------------------------------------
// hard-proposal mining (hrnmp_bbox_head.py:357-414 `hardest_proposal_mining`): per query row of the scaled affinity
// matrix aff [Mq][ld], under the label masks the reference builds with masked_fill + topk:
//   out[r][0] = argmax over keys whose label differs from the row's   (`inds_for_pos_sm`,  topk(1))
//   out[r][1] = argmin over keys with the row's label                 (`inds_for_pos_nsm`, topk(1, largest=False))
//   out[r][2..3] = the two largest among keys whose label differs     (`inds_for_bg`, topk(2); used for label-0 rows)
// Masked-out keys count as -inf (+inf for the minimum), ties go to the lower index: a row without candidates answers 0
// (0, 1 for the pair).  One wavefront per row.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Best { float v; int i; };

__device__ __forceinline__ Best better_max(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ Best better_min(Best a, Best b) { return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

template <bool MAX>
__device__ __forceinline__ Best wave_best(Best x) {
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = MAX ? better_max(x, y) : better_min(x, y);
  }
  return x;
}

}  // namespace

__global__ __launch_bounds__(256) void mining_argreduce_kernel(const float* __restrict__ aff, int Mq, int Mk, long ld,
                                                               const long long* __restrict__ labels,
                                                               const long long* __restrict__ all_labels, long long* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= Mq) return;
  const float* row = aff + (long)r * ld;
  const long long lab = labels[r];
  Best dmax = {-INFINITY, 0x7fffffff}, smin = {INFINITY, 0x7fffffff};
  for (int k = lane; k < Mk; k += 64) {
    const bool diff = all_labels[k] != lab;
    const float v = row[k];
    dmax = better_max(dmax, Best{diff ? v : -INFINITY, k});
    smin = better_min(smin, Best{diff ? INFINITY : v, k});
  }
  dmax = wave_best<true>(dmax);
  smin = wave_best<false>(smin);
  Best second = {-INFINITY, 0x7fffffff};
  for (int k = lane; k < Mk; k += 64) {
    if (k == dmax.i) continue;
    const bool diff = all_labels[k] != lab;
    second = better_max(second, Best{diff ? row[k] : -INFINITY, k});
  }
  second = wave_best<true>(second);
  if (lane == 0) {
    long long* o = out + (long)r * 4;
    o[0] = dmax.i; o[1] = smin.i; o[2] = dmax.i; o[3] = Mk > 1 ? second.i : 0;
  }
}

hipError_t run_mining_argreduce(const float* aff, int Mq, int Mk, long ld, const long long* labels, const long long* all_labels,
                                long long* out, hipStream_t s) {
  hipLaunchKernelGGL(mining_argreduce_kernel, dim3((Mq + 3) / 4), dim3(256), 0, s, aff, Mq, Mk, ld, labels, all_labels, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Triplet margin loss over mined (anchor, positive, negative) index triples -- a STAND-IN: the reference calls
// `TripletNonLocalLoss(margin).compute_loss(q, k, labels, [anchors, pos, neg])` (hrnmp_bbox_head.py:555-561,672,741) from a
// fork of pytorch_metric_learning that is not part of the reference tree.  What is computed here is that library's
// published TripletMarginLoss with the anchors taken from q and the positives / negatives from k:
//   d(x, y) = || x - y + 1e-6 ||_2 (F.pairwise_distance),  l_i = max(d(q_a, k_p) - d(q_a, k_n) + margin, 0),
//   loss = sum l_i / max(#{l_i > 0}, 1)   (AvgNonZeroReducer).
// Forward and both gradients; fixed summation order (anchors are walked in order per column, no atomics).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void triplet_dist_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk, int D,
                                                           const long long* __restrict__ a_idx, const long long* __restrict__ p_idx,
                                                           const long long* __restrict__ n_idx, int n, float margin,
                                                           float* __restrict__ ws) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const T* qa = q + a_idx[i] * ldq;
  const T* kp = k + p_idx[i] * ldk;
  const T* kn = k + n_idx[i] * ldk;
  float sp = 0.f, sn = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float x = (float)qa[d];
    const float ep = x - (float)kp[d] + 1e-6f, en = x - (float)kn[d] + 1e-6f;
    sp += ep * ep;
    sn += en * en;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sp += __shfl_xor(sp, o);
    sn += __shfl_xor(sn, o);
  }
  if (lane == 0) {
    const float dp = sqrtf(sp), dn = sqrtf(sn), l = dp - dn + margin;
    ws[i * 3] = dp; ws[i * 3 + 1] = dn; ws[i * 3 + 2] = l > 0.f ? l : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void triplet_grad_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk, int D,
                                                           const long long* __restrict__ a_idx, const long long* __restrict__ p_idx,
                                                           const long long* __restrict__ n_idx, int n, const float* __restrict__ ws,
                                                           float* __restrict__ out2, float* __restrict__ dq, float* __restrict__ dk) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x;
  float ls = 0.f, cnt = 0.f;
  for (int i = tid; i < n; i += 256) {
    ls += ws[i * 3 + 2];
    cnt += ws[i * 3 + 2] > 0.f ? 1.f : 0.f;
  }
  red[0][tid] = ls; red[1][tid] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
    __syncthreads();
  }
  const float active = red[1][0], g = 1.f / fmaxf(active, 1.f);
  if (blockIdx.x == 0 && tid == 0) { out2[0] = red[0][0] * g; out2[1] = active; }
  if (!dq && !dk) return;
  for (int d = blockIdx.x * 256 + tid; d < D; d += gridDim.x * 256)
    for (int i = 0; i < n; ++i) {
      if (!(ws[i * 3 + 2] > 0.f)) continue;
      const long a = a_idx[i], p = p_idx[i], m = n_idx[i];
      const float x = (float)q[a * ldq + d];
      const float up = (x - (float)k[p * ldk + d] + 1e-6f) / ws[i * 3], un = (x - (float)k[m * ldk + d] + 1e-6f) / ws[i * 3 + 1];
      if (dq) dq[a * D + d] += g * (up - un);
      if (dk) { dk[p * D + d] -= g * up; dk[m * D + d] += g * un; }
    }
}

template <typename T>
static hipError_t launch_triplet(const void* q, long ldq, const void* k, long ldk, int D, const long long* a, const long long* p,
                                 const long long* n_, int n, float margin, float* ws, float* out2, float* dq, float* dk, hipStream_t s) {
  hipLaunchKernelGGL(triplet_dist_kernel<T>, dim3((n + 3) / 4), dim3(256), 0, s, (const T*)q, ldq, (const T*)k, ldk, D, a, p, n_, n, margin,
                     ws);
  hipLaunchKernelGGL(triplet_grad_kernel<T>, dim3((D + 255) / 256), dim3(256), 0, s, (const T*)q, ldq, (const T*)k, ldk, D, a, p, n_, n, ws,
                     out2, dq, dk);
  return hipGetLastError();
}

hipError_t run_triplet_margin(const void* q, long ldq, const void* k, long ldk, int D, int Mq, int Mk, const long long* a, const long long* p,
                              const long long* n_, int n, float margin, int bf16, float* ws, float* out2, float* dq, float* dk,
                              hipStream_t s) {
  hipError_t e = hipSuccess;
  if (dq) e = run_zero_fill(dq, (size_t)Mq * D * 4, s);
  if (e == hipSuccess && dk) e = run_zero_fill(dk, (size_t)Mk * D * 4, s);
  if (e != hipSuccess) return e;
  return bf16 ? launch_triplet<__hip_bfloat16>(q, ldq, k, ldk, D, a, p, n_, n, margin, ws, out2, dq, dk, s)
              : launch_triplet<float>(q, ldq, k, ldk, D, a, p, n_, n, margin, ws, out2, dq, dk, s);
}

// ---- launchers ----
size_t assign_workspace_bytes(int n, int k) { return (size_t)n * 4 + (size_t)((k + 63) / 64 * 64) * 4; }

hipError_t run_max_iou_assign(const float* boxes, int ldb, int n, const float* gts, int k, const uint8_t* valid, float pos_thr,
                              float neg_lo, float neg_hi, float min_pos, long long* gt_inds, float* max_ov, void* ws, hipStream_t s) {
  int* argmax = (int*)ws;
  unsigned* gt_max = (unsigned*)(argmax + n);
  hipError_t e = run_zero_fill(gt_max, (size_t)k * 4, s);
  if (e != hipSuccess) return e;
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(assign_max_kernel, dim3(blocks), dim3(256), 0, s, boxes, ldb, n, gts, k, valid, max_ov, argmax, gt_max);
  hipLaunchKernelGGL(assign_final_kernel, dim3(blocks), dim3(256), 0, s, boxes, ldb, n, gts, k, valid, max_ov, argmax, gt_max, pos_thr,
                     neg_lo, neg_hi, min_pos, gt_inds);
  return hipGetLastError();
}

hipError_t run_sample(const long long* cls, const float* keys, int n, int num, int expected_pos, float neg_pos_ub, long long* inds,
                      int* counts, hipStream_t s) {
  hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, s, cls, keys, n, num, expected_pos, neg_pos_ub, inds, counts);
  return hipGetLastError();
}

hipError_t run_box_targets(const float* boxes, int ldb, int n, const float* gts, const long long* gt_labels, const long long* gt_inds,
                           const long long* inds, const int* counts, int num, const float* means, const float* stds, float pos_weight,
                           int scatter, long long* labels, float* label_w, float* bbox_t, float* bbox_w, hipStream_t s) {
  const size_t rows = scatter ? (size_t)n : (size_t)num;
  hipError_t e = run_zero_fill(labels, rows * 8, s);
  if (e == hipSuccess) e = run_zero_fill(label_w, rows * 4, s);
  if (e == hipSuccess) e = run_zero_fill(bbox_t, rows * 16, s);
  if (e == hipSuccess) e = run_zero_fill(bbox_w, rows * 16, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(box_targets_kernel, dim3((num + 255) / 256), dim3(256), 0, s, boxes, ldb, gts, gt_labels, gt_inds, inds, counts,
                     make_float4(means[0], means[1], means[2], means[3]), make_float4(stds[0], stds[1], stds[2], stds[3]), pos_weight,
                     scatter, labels, label_w, bbox_t, bbox_w);
  return hipGetLastError();
}

hipError_t run_rpn_loss(const float* o, int ldo, int A, int rows, const long long* labels, const float* label_w, const float* bbox_t,
                        const float* bbox_w, const int* counts, float beta, float* out2, float* d_o, hipStream_t s) {
  hipError_t e = run_zero_fill(d_o, (size_t)rows * ldo * 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(1024), 0, s, o, ldo, A, rows, labels, label_w, bbox_t, bbox_w, counts, beta, out2,
                     d_o);
  return hipGetLastError();
}

hipError_t run_ce_rows(const float* logits, int ldl, int cls_off, int ncls, const long long* labels, int R, float* loss, hipStream_t s) {
  hipLaunchKernelGGL(ce_rows_kernel, dim3((R + 255) / 256), dim3(256), 0, s, logits, ldl, cls_off, ncls, labels, R, loss);
  return hipGetLastError();
}

}  // namespace hvr
