// Greedy NMS and the two post-processing pipelines built on it, all device-resident
// (the reference copies the whole IoU mask to the host and sweeps there:
// mmdet/ops/nms/src/nms_kernel.cu:104-128).
//
//   * generic batched NMS              <- mmdet/ops/nms/src/nms_cpu.cpp:5-59 (semantics,
//                                          incl. `ovr >= thr`), nms_kernel.cu:24-68 (bit mask)
//   * RPN proposal selection           <- mmdet/models/anchor_heads/rpn_head.py:55-104
//   * multi-class NMS of the RCNN head <- mmdet/core/post_processing/bbox_nms.py:6-66
//
// Integer / index work: bit-exact against the CPU reference semantics.  Tie rule (unspecified in the
// reference, which relies on torch.sort/topk): higher score first, then lower index.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

__device__ __forceinline__ uint32_t float_key(float f) {  // ascending uint == ascending float
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// In-LDS bitonic sort of n_pow2 (key, idx) pairs: descending key, ascending idx on ties.
// Pad entries must carry key 0 / idx 0xffffffff so they sink to the end.
__device__ __forceinline__ bool pair_before(uint32_t ka, uint32_t ia, uint32_t kb, uint32_t ib) {
  return ka > kb || (ka == kb && ia < ib);
}
__device__ void bitonic_sort_pairs(uint32_t* key, uint32_t* idx, int n_pow2) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t ka = key[i], ia = idx[i], kb = key[ixj], ib = idx[ixj];
          const bool up = (i & k) == 0;  // this sub-sequence sorts "first before last"
          const bool swap = up ? pair_before(kb, ib, ka, ia) : pair_before(ka, ia, kb, ib);
          if (swap) { key[i] = kb; idx[i] = ib; key[ixj] = ka; idx[ixj] = ia; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

__device__ __forceinline__ float box_iou_plus1(const float4 a, const float4 b) {
  // nms_cpu.cpp:18,46-54 (same arithmetic order; "+1" pixel convention)
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y), xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
  const float inter = w * h;
  const float aa = (a.z - a.x + 1.f) * (a.w - a.y + 1.f), ab = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  return inter / (aa + ab - inter);
}

// ---------------------------------------------------------------------------------
// generic batched NMS: problem p has n boxes at dets + p * n * 5
// ---------------------------------------------------------------------------------
// (1) order by score; writes order[p][n] and boxes4[p][n] (sorted x1,y1,x2,y2)
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ dets, int n, int presorted,
                                                        int* __restrict__ order, float4* __restrict__ boxes4) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = blockIdx.x;
  const float* d = dets + (long)p * n * 5;
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);
  const int np2 = next_pow2(n);
  uint32_t* idx = key + np2;
  if (!presorted) {
    for (int i = threadIdx.x; i < np2; i += blockDim.x) {
      key[i] = i < n ? float_key(d[i * 5 + 4]) : 0u;
      idx[i] = i < n ? (uint32_t)i : 0xffffffffu;
    }
    __syncthreads();
    bitonic_sort_pairs(key, idx, np2);
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int src = presorted ? i : (int)idx[i];
    order[(long)p * n + i] = src;
    boxes4[(long)p * n + i] = make_float4(d[src * 5 + 0], d[src * 5 + 1], d[src * 5 + 2], d[src * 5 + 3]);
  }
}

// (2) upper-triangular suppression bit mask over the score-sorted boxes
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes4, int n, float thr, int ge,
                                                      unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x, p = blockIdx.z;
  if (cb < rb) return;
  const int nb = (n + 63) >> 6;
  const float4* b = boxes4 + (long)p * n;
  __shared__ float4 colbox[64];
  const int cj = cb * 64 + threadIdx.x;
  if (cj < n) colbox[threadIdx.x] = b[cj];
  __syncthreads();
  const int i = rb * 64 + threadIdx.x;
  if (i >= n) return;
  const float4 me = b[i];
  const int csize = min(64, n - cb * 64);
  unsigned long long bits = 0ull;
  const int start = (rb == cb) ? threadIdx.x + 1 : 0;
  for (int j = start; j < csize; ++j) {
    const float ovr = box_iou_plus1(me, colbox[j]);
    if (ge ? (ovr >= thr) : (ovr > thr)) bits |= 1ull << j;
  }
  mask[((long)p * n + i) * nb + cb] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// (3) sequential sweep by wave 0 (stops after max_keep survivors when max_keep > 0), then the
//     whole block compacts the survivors in ascending ORIGINAL index (nms_cpu.cpp:58).
__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                        const int* __restrict__ order, int n, int max_keep,
                                                        long long* __restrict__ keep, int* __restrict__ n_keep,
                                                        int keep_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = blockIdx.x;
  const int nb = (n + 63) >> 6;  // <= 128
  const int nbs = nb;            // mask row stride in words
  unsigned char* flag = reinterpret_cast<unsigned char*>(smem);  // [n] by original index
  int* wsum = reinterpret_cast<int*>(smem + ((n + 15) & ~15));   // [blockDim/64 + 1]
  for (int i = threadIdx.x; i < n; i += blockDim.x) flag[i] = 0;
  __syncthreads();
  const unsigned long long* mk = mask + (long)p * n * nbs;
  const int* ord = order + (long)p * n;
  if (threadIdx.x < 64) {
    // One wave walks the score-sorted boxes chunk by chunk (64 boxes).  Only the "is this box already suppressed /
    // does it suppress later boxes of its own chunk" chain is serial, and it runs on readlanes of the chunk's
    // diagonal mask word; everything that touches memory is kept off that chain: the next chunk's diagonal words are
    // requested one chunk ahead, and the mask rows of the chunk's survivors are fetched eight rows (16 independent
    // loads) at a time before they are OR-ed into the running suppression set.
    const int lane = threadIdx.x;
    unsigned long long remv0 = 0ull, remv1 = 0ull;
    int nkept = 0;
    bool done = false;
    unsigned long long diag_next = lane < n ? mk[(long)lane * nbs] : 0ull;
    for (int cb = 0; cb < nb && !done; ++cb) {
      unsigned long long cur = cb < 64 ? readlane64(remv0, cb) : readlane64(remv1, cb - 64);
      const int bi = cb * 64 + lane;
      const unsigned long long diag = diag_next;
      if (cb + 1 < nb) {
        const int bn = bi + 64;
        diag_next = bn < n ? mk[(long)bn * nbs + cb + 1] : 0ull;
      }
      const int cnt = min(64, n - cb * 64);
      unsigned long long kept = 0ull;
      for (int j = 0; j < cnt; ++j) {
        if (!((cur >> j) & 1ull)) {
          kept |= 1ull << j;
          cur |= readlane64(diag, j);
          if (++nkept == max_keep) { done = true; break; }
        }
      }
      if (bi < n && ((kept >> lane) & 1ull)) flag[ord[bi]] = 1;
      if (!done) {
        unsigned long long kb = kept;  // wave-uniform
        const bool lo_on = lane > cb && lane < nb, hi_on = lane + 64 > cb && lane + 64 < nb;
        while (kb) {
          unsigned long long r0[8], r1[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            r0[u] = r1[u] = 0ull;
            if (kb) {
              const int j = __builtin_ctzll(kb);
              kb &= kb - 1;
              const unsigned long long* row = mk + (long)(cb * 64 + j) * nbs;
              if (lo_on) r0[u] = row[lane];
              if (hi_on) r1[u] = row[lane + 64];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) { remv0 |= r0[u]; remv1 |= r1[u]; }
        }
      }
    }
  }
  __syncthreads();
  // ordered compaction: per-thread contiguous segment
  const int per = (n + blockDim.x - 1) / blockDim.x;
  const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += flag[i];
  int incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
  int pos = base + incl - c;
  long long* kp = keep + (long)p * keep_stride;
  for (int i = lo; i < hi; ++i)
    if (flag[i]) kp[pos++] = i;
  if (threadIdx.x == blockDim.x - 1) n_keep[p] = base + incl;
}

// ---------------------------------------------------------------------------------
// RPN: sigmoid -> top-k -> delta2bbox, one workgroup per frame
// ---------------------------------------------------------------------------------
// cls  [T][HW*A] logits in (y, x, anchor) order (NHWC conv output == permute(1,2,0))
// reg  [T][HW*A][4]
// out  [T][npre][5], sorted by score (desc) when HW*A > nms_pre, else original order

template <typename T>
__global__ __launch_bounds__(1024) void rpn_select_kernel(const T* __restrict__ cls, const T* __restrict__ reg, long cls_stride,
                                                          long reg_stride, float* __restrict__ out, const RpnParams rp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = blockIdx.x, n = rp.n_anchor, k = rp.npre;
  const T* c = cls + (long)f * cls_stride;
  const T* r = reg + (long)f * reg_stride;
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);
  const int kp2 = next_pow2(k);
  uint32_t* idx = key + kp2;
  __shared__ int hist[256];
  __shared__ uint32_t sh_prefix, sh_remaining;
  __shared__ int sh_count;

  // anchor i = cell * A + a lives at pixel `cell`, channel a of a [H*W][cls_pitch] map (pitch >= A lets the
  // objectness and delta maps be channel slices of one fused conv output)
  const int A = rp.A, cpitch = rp.cls_pitch, rpitch = rp.reg_pitch;
  auto score_of = [&](int i) {
    const int cell = i / A, a = i - cell * A;
    return 1.f / (1.f + expf(-ElemTraits<T>::load(c + (long)cell * cpitch + a)));  // torch.sigmoid
  };

  if (n > k) {
    // radix select (MSB first) of the k-th largest (key, then smallest index) element
    uint32_t prefix = 0u, remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
      __syncthreads();
      const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t u = float_key(score_of(i));
        if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t rem = remaining;
        int b = 255;
        for (; b > 0; --b) {
          if ((uint32_t)hist[b] >= rem) break;
          rem -= hist[b];
        }
        sh_prefix = prefix | ((uint32_t)b << shift);
        sh_remaining = rem;
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_remaining;
      __syncthreads();
    }
    // prefix == key of the k-th element; `remaining` of the equal-key elements are taken, lowest index first
    const uint32_t thr_key = prefix;
    if (threadIdx.x == 0) sh_count = 0;
    for (int i = threadIdx.x; i < kp2; i += blockDim.x) { key[i] = 0u; idx[i] = 0xffffffffu; }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t u = float_key(score_of(i));
      if (u > thr_key) {
        const int pos = atomicAdd(&sh_count, 1);
        key[pos] = u;
        idx[pos] = (uint32_t)i;
      }
    }
    __syncthreads();
    // ties at the threshold: ordered pick by one wave (rare path: `remaining` is usually 1)
    if (threadIdx.x < 64) {
      int taken = 0, base = sh_count;
      for (int i0 = 0; i0 < n && taken < (int)remaining; i0 += 64) {
        const int i = i0 + threadIdx.x;
        const bool eq = i < n && float_key(score_of(i)) == thr_key;
        const unsigned long long m = __ballot(eq);
        const int before = __popcll(m & ((1ull << threadIdx.x) - 1ull));
        if (eq && taken + before < (int)remaining) {
          key[base + taken + before] = thr_key;
          idx[base + taken + before] = (uint32_t)i;
        }
        taken += __popcll(m);
      }
    }
    __syncthreads();
    bitonic_sort_pairs(key, idx, kp2);
  } else {
    for (int i = threadIdx.x; i < k; i += blockDim.x) idx[i] = (uint32_t)i;
    __syncthreads();
  }
  // decode (delta2bbox, mmdet/core/bbox/transforms.py:78-110) in sorted order
  const int cnt = n > k ? k : n;
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    const int i = (int)idx[j];
    const int a = i % rp.A, cell = i / rp.A, x = cell % rp.W, y = cell / rp.W;
    const float ax1 = rp.base[a][0] + x * rp.stride, ay1 = rp.base[a][1] + y * rp.stride;
    const float ax2 = rp.base[a][2] + x * rp.stride, ay2 = rp.base[a][3] + y * rp.stride;
    float d[4];
    load4(r + (long)cell * rpitch + a * 4, d);
    const float dx = d[0] * rp.s[0] + rp.m[0], dy = d[1] * rp.s[1] + rp.m[1];
    float dw = d[2] * rp.s[2] + rp.m[2], dh = d[3] * rp.s[3] + rp.m[3];
    dw = fminf(fmaxf(dw, -rp.max_ratio), rp.max_ratio);
    dh = fminf(fmaxf(dh, -rp.max_ratio), rp.max_ratio);
    const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f, pw = ax2 - ax1 + 1.0f, ph = ay2 - ay1 + 1.0f;
    const float gw = pw * expf(dw), gh = ph * expf(dh), gx = px + pw * dx, gy = py + ph * dy;
    float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f, x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f;
    x1 = fminf(fmaxf(x1, 0.f), rp.img_w - 1.f);
    y1 = fminf(fmaxf(y1, 0.f), rp.img_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), rp.img_w - 1.f);
    y2 = fminf(fmaxf(y2, 0.f), rp.img_h - 1.f);
    float* o = out + ((long)f * k + j) * 5;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = score_of(i);
  }
}

// first nms_post survivors (ascending index), then topk(min(max_num, count)) by score
// (rpn_head.py:92-103).  props [T][npre][5]; keep [T][npre]; out [T][max_num][5]; rois [T][max_num][5]
__global__ __launch_bounds__(256) void rpn_gather_kernel(const float* __restrict__ props, const long long* __restrict__ keep,
                                                         const int* __restrict__ n_keep, int npre, int nms_post, int max_num,
                                                         float* __restrict__ out, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = blockIdx.x;
  int cnt = n_keep[f];
  cnt = cnt < nms_post ? cnt : nms_post;
  const int np2 = next_pow2(cnt > 1 ? cnt : 1);
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);
  uint32_t* idx = key + np2;
  const float* pf = props + (long)f * npre * 5;
  const long long* kf = keep + (long)f * npre;
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    if (i < cnt) { key[i] = float_key(pf[kf[i] * 5 + 4]); idx[i] = (uint32_t)i; }
    else { key[i] = 0u; idx[i] = 0xffffffffu; }
  }
  __syncthreads();
  bitonic_sort_pairs(key, idx, np2);
  const int num = cnt < max_num ? cnt : max_num;
  for (int j = threadIdx.x; j < num * 5; j += blockDim.x) {
    const int row = j / 5, col = j - row * 5;
    out[((long)f * max_num + row) * 5 + col] = pf[kf[idx[row]] * 5 + col];
  }
  if (threadIdx.x == 0) counts[f] = num;
}

// ---------------------------------------------------------------------------------
// multi-class NMS (bbox_nms.py:6-66) for class-agnostic boxes
// ---------------------------------------------------------------------------------
constexpr int MC_MAX_R = 512;

// one workgroup per foreground class: keepflag[c-1][r], kcount[c-1]
__global__ __launch_bounds__(1024) void mc_nms_class_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           int R, int ncls, float score_thr, float iou_thr,
                                                           unsigned char* __restrict__ keepflag, int* __restrict__ kcount) {
  __shared__ float4 bx[MC_MAX_R];
  __shared__ unsigned long long iou[MC_MAX_R][MC_MAX_R / 64];
  __shared__ uint32_t key[MC_MAX_R], idx[MC_MAX_R];
  __shared__ unsigned char kf[MC_MAX_R];
  const int c = blockIdx.x + 1;
  const int words = (R + 63) >> 6;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    bx[r] = make_float4(boxes[r * 4 + 0], boxes[r * 4 + 1], boxes[r * 4 + 2], boxes[r * 4 + 3]);
    kf[r] = 0;
  }
  __syncthreads();
  int ncand = 0;
  for (int i = threadIdx.x; i < MC_MAX_R; i += blockDim.x) {
    const bool v = i < R && scores[(long)i * ncls + c] > score_thr;
    key[i] = v ? float_key(scores[(long)i * ncls + c]) : 0u;
    idx[i] = v ? (uint32_t)i : 0xffffffffu;
    ncand += v;
  }
  __shared__ int sh_ncand;
  if (threadIdx.x == 0) sh_ncand = 0;
  __syncthreads();
  atomicAdd(&sh_ncand, ncand);
  __syncthreads();
  ncand = sh_ncand;
  if (ncand == 0) {
    if (threadIdx.x == 0) kcount[c - 1] = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) keepflag[(long)(c - 1) * R + r] = 0;
    return;
  }
  // pairwise suppression bits (symmetric, class-agnostic boxes)
  for (int t = threadIdx.x; t < R * words; t += blockDim.x) {
    const int r = t / words, w = t - r * words;
    unsigned long long bits = 0ull;
    const float4 me = bx[r];
    const int qn = min(64, R - w * 64);
    for (int j = 0; j < qn; ++j) {
      const int q = w * 64 + j;
      if (q != r && box_iou_plus1(me, bx[q]) >= iou_thr) bits |= 1ull << j;
    }
    iou[r][w] = bits;
  }
  __syncthreads();
  bitonic_sort_pairs(key, idx, MC_MAX_R);
  if (threadIdx.x < 64) {
    // Greedy sweep in score order by one wave; lane w holds word w of the suppression set (w < 8).  The serial part
    // is register-only: per chunk of 64 candidates the lanes first fetch the candidates' indices (one LDS read) and
    // this lane's word of every candidate's IoU row (64 independent LDS reads in flight), then the chain
    // "suppressed? -> keep -> OR the row in" runs on readlanes, with no memory access in its dependency path.
    const int lane = threadIdx.x;
    unsigned long long supp = 0ull;
    int kept = 0;
    // (chunks of 32: 64 prefetched 64-bit rows are 128 VGPRs, the whole budget of a 1024-thread workgroup -- the
    // compiler spilled; nms.hip is in build.sh's -Rpass-analysis / check_regs gate now)
    constexpr int CH = 32;
    for (int base = 0; base < ncand; base += CH) {
      const int mine = (lane < CH && base + lane < ncand) ? (int)idx[base + lane] : 0;
      unsigned long long rows[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int r = __builtin_amdgcn_readlane(mine, j);
        rows[j] = lane < words ? iou[r][lane] : 0ull;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (base + j < ncand) {  // wave-uniform
          const int r = __builtin_amdgcn_readlane(mine, j);
          const unsigned long long wv = readlane64(supp, r >> 6);
          if (!((wv >> (r & 63)) & 1ull)) {
            if (lane == 0) kf[r] = 1;
            ++kept;
            supp |= rows[j];
          }
        }
      }
    }
    if (lane == 0) kcount[c - 1] = kept;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) keepflag[(long)(c - 1) * R + r] = kf[r];
}

// concatenates the classes (class-major, roi-ascending); if more than max_num survive,
// keeps the max_num highest scores in descending order (bbox_nms.py:54-61)
__global__ __launch_bounds__(1024) void mc_nms_merge_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                            int R, int ncls, const unsigned char* __restrict__ keepflag,
                                                            const int* __restrict__ kcount, int max_num, int sp2,
                                                            float* __restrict__ dets, long long* __restrict__ labels,
                                                            int* __restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int offs[128];
  __shared__ int sh_total;
  const int nfg = ncls - 1;
  if (threadIdx.x == 0) {
    int t = 0;
    for (int c = 0; c < nfg; ++c) { offs[c] = t; t += kcount[c]; }
    sh_total = t;
  }
  __syncthreads();
  const int total = sh_total;
  const int np2 = next_pow2(total > 1 ? total : 1);
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);
  uint32_t* idx = key + np2;  // (class << 16) | roi: ascending == position in the concatenated list
  for (int i = threadIdx.x; i < np2; i += blockDim.x) { key[i] = 0u; idx[i] = 0xffffffffu; }
  __syncthreads();
  // one wave per class: ordered compaction of the class's survivors
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int c = wave; c < nfg; c += nw) {
    int pos = offs[c];
    for (int r0 = 0; r0 < R; r0 += 64) {
      const int r = r0 + lane;
      const bool k = r < R && keepflag[(long)c * R + r];
      const unsigned long long m = __ballot(k);
      if (k) {
        const int q = pos + __popcll(m & ((1ull << lane) - 1ull));
        key[q] = float_key(scores[(long)r * ncls + c + 1]);
        idx[q] = ((uint32_t)c << 16) | (uint32_t)r;
      }
      pos += __popcll(m);
    }
  }
  __syncthreads();
  int nout = total;
  if (total > max_num && sp2 == 0) {
    // no LDS for the select stage (max_num close to the list length, e.g. the reference's max_num = -1 quirk): sort the
    // whole list in place -- same order (score desc, list position asc), the padding entries (key 0) go last
    bitonic_sort_pairs(key, idx, np2);
    nout = max_num;
  } else if (total > max_num) {
    // top max_num by (score desc, list position asc): radix-select the max_num-th key (4 passes over the LDS list),
    // gather the entries above it plus the first ties, and sort only those (bbox_nms.py:54-61 sorts everything and
    // cuts; the survivors and their order are the same)
    __shared__ int hist[256];
    __shared__ uint32_t sh_prefix, sh_remaining;
    __shared__ int sh_count;
    uint32_t* skey = idx + np2;
    uint32_t* sidx = skey + sp2;
    uint32_t prefix = 0u, remaining = (uint32_t)max_num;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
      __syncthreads();
      const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const uint32_t u = key[i];
        if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t rem = remaining;
        int b = 255;
        for (; b > 0; --b) {
          if ((uint32_t)hist[b] >= rem) break;
          rem -= hist[b];
        }
        sh_prefix = prefix | ((uint32_t)b << shift);
        sh_remaining = rem;
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_remaining;
      __syncthreads();
    }
    const uint32_t thr_key = prefix;  // key of the max_num-th entry; `remaining` of the entries equal to it are taken
    if (threadIdx.x == 0) sh_count = 0;
    for (int i = threadIdx.x; i < sp2; i += blockDim.x) { skey[i] = 0u; sidx[i] = 0xffffffffu; }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      if (key[i] > thr_key) {
        const int pos = atomicAdd(&sh_count, 1);
        skey[pos] = key[i];
        sidx[pos] = idx[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // ties at the threshold, lowest list position first (the list is in ascending idx order)
      int taken = 0;
      const int base = sh_count;
      for (int i0 = 0; i0 < total && taken < (int)remaining; i0 += 64) {
        const int i = i0 + threadIdx.x;
        const bool eq = i < total && key[i] == thr_key;
        const unsigned long long m = __ballot(eq);
        const int before = __popcll(m & ((1ull << threadIdx.x) - 1ull));
        if (eq && taken + before < (int)remaining) {
          skey[base + taken + before] = thr_key;
          sidx[base + taken + before] = idx[i];
        }
        taken += __popcll(m);
      }
    }
    __syncthreads();
    bitonic_sort_pairs(skey, sidx, sp2);
    idx = sidx;
    nout = max_num;
  }
  for (int j = threadIdx.x; j < nout; j += blockDim.x) {
    const uint32_t e = idx[j];
    const int c = e >> 16, r = e & 0xffff;
    dets[j * 5 + 0] = boxes[r * 4 + 0];
    dets[j * 5 + 1] = boxes[r * 4 + 1];
    dets[j * 5 + 2] = boxes[r * 4 + 2];
    dets[j * 5 + 3] = boxes[r * 4 + 3];
    dets[j * 5 + 4] = scores[(long)r * ncls + c + 1];
    labels[j] = c;
  }
  // rows behind the count are zero (the caller hands uninitialised buffers)
  for (int j = nout + threadIdx.x; j < max_num; j += blockDim.x) {
    dets[j * 5 + 0] = dets[j * 5 + 1] = dets[j * 5 + 2] = dets[j * 5 + 3] = dets[j * 5 + 4] = 0.f;
    labels[j] = 0;
  }
  if (threadIdx.x == 0) *n_out = nout;
}

// ---------------- launchers ----------------
size_t nms_workspace_bytes(int P, int n) {
  const size_t nb = (n + 63) / 64;
  size_t b = 0;
  b += ((size_t)P * n * sizeof(int) + 255) & ~(size_t)255;                    // order
  b += ((size_t)P * n * sizeof(float4) + 255) & ~(size_t)255;                 // sorted boxes
  b += ((size_t)P * n * nb * sizeof(unsigned long long) + 255) & ~(size_t)255;  // mask
  return b;
}

hipError_t run_nms_batched(const float* dets, int P, int n, float thr, int ge, int presorted, int max_keep,
                           long long* keep, int* n_keep, void* ws, hipStream_t s) {
  if (n <= 0 || P <= 0) return hipSuccess;
  if (n > 8192) return hipErrorInvalidValue;
  char* w = (char*)ws;
  int* order = (int*)w;
  w += ((size_t)P * n * sizeof(int) + 255) & ~(size_t)255;
  float4* boxes4 = (float4*)w;
  w += ((size_t)P * n * sizeof(float4) + 255) & ~(size_t)255;
  unsigned long long* mask = (unsigned long long*)w;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const int nb = (n + 63) / 64;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    attr = true;
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(P), dim3(1024), (size_t)np2 * 8, s, dets, n, presorted, order, boxes4);
  const size_t sweep_lds = ((n + 15) & ~15) + 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, P), dim3(64), 0, s, boxes4, n, thr, ge, mask);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(P), dim3(256), sweep_lds, s, mask, order, n, max_keep, keep, n_keep, n);
  return hipGetLastError();
}

hipError_t run_rpn_select(const void* cls, const void* reg, long cls_stride, long reg_stride, float* out,
                          const RpnParams& rp, int dtype, hipStream_t s) {
  int kp2 = 1;
  while (kp2 < rp.npre) kp2 <<= 1;
  if (kp2 > 8192) return hipErrorInvalidValue;
  const size_t lds = (size_t)kp2 * 8;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_select_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_select_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    attr = true;
  }
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(rpn_select_kernel<bf16_t>, dim3(rp.T), dim3(1024), lds, s, (const bf16_t*)cls, (const bf16_t*)reg, cls_stride, reg_stride, out, rp);
  else
    hipLaunchKernelGGL(rpn_select_kernel<float>, dim3(rp.T), dim3(1024), lds, s, (const float*)cls, (const float*)reg, cls_stride, reg_stride, out, rp);
  return hipGetLastError();
}

hipError_t run_rpn_gather(const float* props, const long long* keep, const int* n_keep, int T, int npre, int nms_post,
                          int max_num, float* out, int* counts, hipStream_t s) {
  int np2 = 1;
  while (np2 < nms_post) np2 <<= 1;
  if (np2 > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rpn_gather_kernel, dim3(T), dim3(256), (size_t)np2 * 8, s, props, keep, n_keep, npre, nms_post, max_num, out, counts);
  return hipGetLastError();
}

hipError_t run_multiclass_nms(const float* boxes, const float* scores, int R, int ncls, float score_thr, float iou_thr,
                              int max_num, float* dets, long long* labels, int* n_out, void* ws, hipStream_t s) {
  if (R > MC_MAX_R || ncls - 1 > 128 || R <= 0) return hipErrorInvalidValue;
  unsigned char* keepflag = (unsigned char*)ws;
  int* kcount = (int*)((char*)ws + (((size_t)(ncls - 1) * R + 255) & ~(size_t)255));
  hipLaunchKernelGGL(mc_nms_class_kernel, dim3(ncls - 1), dim3(1024), 0, s, boxes, scores, R, ncls, score_thr, iou_thr, keepflag, kcount);
  int np2 = 1;
  while (np2 < (ncls - 1) * R) np2 <<= 1;
  // the select stage (radix-select the max_num-th score, sort only the entries above it) needs a second LDS list of
  // next_pow2(max_num) entries; when that does not fit beside the survivor list -- max_num close to the list length -- or can
  // never run (max_num >= every possible survivor count) the kernel sorts the survivor list in place instead (sp2 = 0)
  int sp2 = 1;
  while (sp2 < max_num) sp2 <<= 1;
  if (max_num >= (ncls - 1) * R || (size_t)np2 * 8 + (size_t)sp2 * 8 > 160 * 1024 - 3072) sp2 = 0;
  const size_t lds = (size_t)np2 * 8 + (size_t)sp2 * 8;
  if (lds > 160 * 1024 - 3072) return hipErrorInvalidValue;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mc_nms_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 3072);
    attr = true;
  }
  hipLaunchKernelGGL(mc_nms_merge_kernel, dim3(1), dim3(1024), lds, s, boxes, scores, R, ncls, keepflag, kcount, max_num, sp2, dets, labels, n_out);
  return hipGetLastError();
}

size_t multiclass_nms_workspace_bytes(int R, int ncls) {
  return (((size_t)(ncls - 1) * R + 255) & ~(size_t)255) + (size_t)(ncls - 1) * sizeof(int) + 256;
}

}  // namespace hvr
