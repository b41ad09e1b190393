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
#include <cstdlib>
#include <atomic>
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

// Descending sort of n_pow2 64-bit composites (key << 32 | 0xffffffff - index: higher key first, lower index on ties; pads
// are 0 and sink to the end) by a 1024-thread workgroup, the same bitonic network as above with most of it off the LDS:
// thread t holds elements m * 1024 + t (m < E) in registers, so partners at distance j >= 1024 are its own registers, at
// j < 64 a lane of its own wave (shuffles, no barrier), and only 64 <= j < 1024 goes through the LDS (22 of the 91 stages at
// n = 8192; the in-LDS version pays two array reads, a conditional write and a barrier in every one of the 91).
template <int E>
__device__ __forceinline__ void block_sort_desc_u64_regs(unsigned long long* buf) {
  constexpr int B = 1024;
  const int t = threadIdx.x;
  unsigned long long v[E];
#pragma unroll
  for (int m = 0; m < E; ++m) v[m] = buf[m * B + t];
  for (int k = 2; k <= E * B; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= B) {
#pragma unroll
        for (int jm = E / 2; jm > 0; jm >>= 1) {  // register indices stay compile-time constants
          if (j != jm * B) continue;
#pragma unroll
          for (int m = 0; m < E; ++m) {
            const int pm = m ^ jm;
            if (pm > m) {
              const bool up = ((m * B) & k) == 0;
              const unsigned long long a = v[m], b = v[pm];
              const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
              v[m] = up ? hi : lo;
              v[pm] = up ? lo : hi;
            }
          }
        }
      } else if (j >= 64) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; ++m) buf[m * B + t] = v[m];
        __syncthreads();
        const bool lower = (t & j) == 0;
#pragma unroll
        for (int m = 0; m < E; ++m) {
          const unsigned long long o = buf[m * B + (t ^ j)];
          const bool up = ((m * B + t) & k) == 0;
          const bool keep_max = lower == up;
          const unsigned long long a = v[m];
          v[m] = keep_max ? (a > o ? a : o) : (a > o ? o : a);
        }
      } else {
        const bool lower = (t & j) == 0;
#pragma unroll
        for (int m = 0; m < E; ++m) {
          const unsigned long long a = v[m];
          const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)a, j), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(a >> 32), j);
          const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
          const bool up = ((m * B + t) & k) == 0;
          const bool keep_max = lower == up;
          v[m] = keep_max ? (a > o ? a : o) : (a > o ? o : a);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < E; ++m) buf[m * B + t] = v[m];
  __syncthreads();
}

// any workgroup size / length: the plain in-LDS network
__device__ void block_sort_desc_u64(unsigned long long* buf, int n_pow2) {
  if (blockDim.x == 1024 && n_pow2 >= 1024 && n_pow2 <= 8192) {
    switch (n_pow2 >> 10) {
      case 1: block_sort_desc_u64_regs<1>(buf); return;
      case 2: block_sort_desc_u64_regs<2>(buf); return;
      case 4: block_sort_desc_u64_regs<4>(buf); return;
      case 8: block_sort_desc_u64_regs<8>(buf); return;
    }
  }
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = buf[i], b = buf[ixj];
          const bool up = (i & k) == 0;
          if (up ? (b > a) : (a > b)) { buf[i] = b; buf[ixj] = a; }
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

__device__ __forceinline__ float box_area_plus1(const float4 a) { return (a.z - a.x + 1.f) * (a.w - a.y + 1.f); }

// The comparison  box_iou_plus1(a, b) >= thr  (ge) /  > thr  -- the same truth value as the rounded quotient gives, without
// the division in all but a sliver of cases.  r = fma(-thr, uni, inter) is (inter - thr uni) rounded once, so it carries the
// sign of (inter / uni - thr) whenever it is not zero.  The ROUNDED quotient can sit on the other side of thr than the real
// one only when the real one is within an ulp of thr, i.e. |inter - thr uni| < 2^-22 thr uni; inside 2^-21 thr uni (and for
// uni <= 0, NaNs, thr <= 0) the division is done as before.  aa / ab: box_area_plus1 of a / b (the same expressions
// box_iou_plus1 evaluates, so `uni` is the same float).
__device__ __forceinline__ bool box_iou_hits(const float4 a, const float aa, const float4 b, const float ab, const float thr,
                                             const float band_k, const int ge) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y), xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
  const float inter = w * h;
  const float uni = aa + ab - inter;
  const float r = __builtin_fmaf(-thr, uni, inter);
  const float band = band_k * uni;  // band_k = 2^-21 thr (<= 0 switches the shortcut off)
  if (band > 0.f && r > band) return true;
  if (band > 0.f && r < -band) return false;
  const float ovr = inter / uni;
  return ge ? (ovr >= thr) : (ovr > thr);
}
__device__ __forceinline__ float iou_band_k(float thr) { return thr > 0.f ? thr * 4.76837158203125e-07f : 0.f; }

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

// survivors (flag[i] != 0, i = original index) -> ascending index list + count; whole workgroup, wsum [blockDim / 64 + 1]
__device__ __forceinline__ void compact_flags(const unsigned char* flag, int* wsum, int n, long long* kp, int* n_out) {
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
  for (int i = lo; i < hi; ++i)
    if (flag[i]) kp[pos++] = i;
  if (threadIdx.x == blockDim.x - 1) *n_out = base + incl;
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
  compact_flags(flag, wsum, n, keep + (long)p * keep_stride, n_keep + p);
}

// (2 + 3 in one, for a SMALL survivor cap) Greedy NMS that evaluates only the IoUs it needs.  The mask kernel above prices every
// pair (n^2 / 2: 18 M IoUs per frame at the RPN's n = 6000) although only the rows of boxes that SURVIVE are ever read, the RPN
// keeps at most nms_post = 300 of them (rpn_head.py:92-103), and nothing behind the box at which the 300th survivor is found
// matters at all.  One workgroup per problem, the score-sorted boxes in the LDS, a suppression set that is kept up to date
// only for a FRONT of `L` 64-box words which grows as the sweep reaches it; per chunk of 64 boxes:
//   0. (chunk at the front) the next EXT words are brought up to date against every survivor so far;
//   1. the chunk's 64 x 64 diagonal block: wave w takes rows 4 w .. 4 w + 3 -- row j's word is ONE ballot over the lanes'
//      IoU(box j, box lane) (the block is symmetric, so the ballot over rows is the row over columns);
//   2. the serial "survives / suppresses the rest of its chunk" chain, scalar code (ctz over the candidates, a readlane of
//      the row word) run by ONE wave -- the CU has a single scalar unit, sixteen waves running it redundantly to save the
//      broadcast took sixteen times as long;
//   3. the chunk's new survivors against the later words inside the front.
// Steps 0 and 3 are one routine over (word, block of 16 survivors) tasks dealt to the waves, four survivors per step (four
// independent IoU chains per lane), results OR-ed into the word with an LDS atomic.  IoU work: survivors x boxes walked
// (300 x ~1000 on typical RPN output) instead of n^2 / 2, and no more than survivors x n; fully suppressed words are skipped.
// Same arithmetic and comparison as the mask kernel (box_iou_hits == the rounded quotient's verdict, row = the earlier box), so
// the survivor set is the same bit for bit; 15 workgroups instead of a chip-filling mask launch: the chip stays with res5.
__global__ __launch_bounds__(1024) void nms_greedy_kernel(const float* __restrict__ dets, const float4* __restrict__ boxes4,
                                                          const int* __restrict__ order, int n, float thr, int ge, int max_keep,
                                                          long long* __restrict__ keep, int* __restrict__ n_keep, int keep_stride,
                                                          const int* __restrict__ skip) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EXT = 4;  // words the front grows by
  if (skip && skip[blockIdx.x]) return;  // (the band sweep below has already finished this problem)
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (int)(blockDim.x >> 6);
  const int nb = (n + 63) >> 6;  // <= 128
  float4* box = reinterpret_cast<float4*>(smem);                                        // [nb * 64], score order
  unsigned long long* removed = reinterpret_cast<unsigned long long*>(box + nb * 64);  // [nb]
  unsigned long long* diag = removed + nb;                                               // [64]
  unsigned char* flag = reinterpret_cast<unsigned char*>(diag + 64);                     // [n] by original index
  int* wsum = reinterpret_cast<int*>(flag + ((n + 15) & ~15));                           // [blockDim / 64 + 1]
  unsigned short* kall = reinterpret_cast<unsigned short*>(wsum + 32);                   // [max_keep] survivors, score order
  __shared__ unsigned long long sh_kept;
  __shared__ int sh_nkept, sh_done;
  const float band_k = iou_band_k(thr);
  const float* d = dets ? dets + (long)p * n * 5 : nullptr;  // presorted: read in place
  const float4* b4 = boxes4 ? boxes4 + (long)p * n : nullptr;
  const int* ord = order ? order + (long)p * n : nullptr;
  for (int i = tid; i < nb * 64; i += blockDim.x) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) v = b4 ? b4[i] : make_float4(d[i * 5 + 0], d[i * 5 + 1], d[i * 5 + 2], d[i * 5 + 3]);
    box[i] = v;
  }
  for (int i = tid; i < nb; i += blockDim.x) removed[i] = 0ull;
  for (int i = tid; i < n; i += blockDim.x) flag[i] = 0;
  __syncthreads();

  // survivors kall[k0 .. k1) against the words [w0, w1)
  auto apply = [&](int w0, int w1, int k0, int k1) {
    const int nkb = (k1 - k0 + 15) >> 4, ntask = (w1 - w0) * nkb;
    for (int t = wave; t < ntask; t += nwaves) {
      const int cw = w0 + t / nkb, kb = k0 + (t % nkb) * 16, cnt = min(16, k1 - kb);
      const unsigned long long gone = removed[cw];
      if (__builtin_amdgcn_readfirstlane((int)(gone == ~0ull))) continue;  // nothing left to suppress in this word
      const float4 col = box[cw * 64 + lane];
      const float ca = box_area_plus1(col);
      const int myk = (int)kall[kb + min(lane & 15, cnt - 1)];  // lanes 0..15: the block's survivors (the tail repeats the last)
      bool hit = false;
      for (int q = 0; q < cnt; q += 4) {
        float4 bj[4];
        float ba[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bj[e] = box[__builtin_amdgcn_readlane(myk, q + e)];
          ba[e] = box_area_plus1(bj[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) hit |= box_iou_hits(bj[e], ba[e], col, ca, thr, band_k, ge);
      }
      const unsigned long long m = __ballot(hit) & ~gone;
      if (lane == 0 && m) atomicOr(&removed[cw], m);
    }
  };

  int nkept = 0, L = min(nb, EXT);
  for (int cb = 0; cb < nb; ++cb) {
    if (cb >= L) {  // the sweep has reached the front: bring the next words up to date
      const int L2 = min(nb, L + EXT);
      if (nkept > 0) apply(L, L2, 0, nkept);
      L = L2;
      __syncthreads();
    }
    const int bi = cb * 64 + lane;
    const float4 mine = box[bi];
    const float mine_area = box_area_plus1(mine);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 4 * wave + u;
      const float4 bj = box[cb * 64 + j];
      const bool hit = box_iou_hits(bj, box_area_plus1(bj), mine, mine_area, thr, band_k, ge) && lane > j && bi < n;
      const unsigned long long m = __ballot(hit);
      if (lane == 0) diag[j] = m;
    }
    __syncthreads();
    if (wave == 0) {
      const unsigned long long drow = diag[lane];
      const unsigned long long rm = removed[cb];
      unsigned long long cur = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rm >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rm);
      const int cnt = min(64, n - cb * 64);
      const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
      unsigned long long kept_w = 0ull, cand = ~cur & valid;
      int nk_w = nkept;
      bool done_w = false;
      while (cand) {
        const int j = __builtin_ctzll(cand);
        kept_w |= 1ull << j;
        if (++nk_w == max_keep) { done_w = true; break; }
        cur |= readlane64(drow, j);
        cand = ~cur & valid & ~((2ull << j) - 1ull);
      }
      if ((kept_w >> lane) & 1ull) {
        kall[nkept + (int)__popcll(kept_w & ((1ull << lane) - 1ull))] = (unsigned short)bi;
        flag[ord ? ord[bi] : bi] = 1;
      }
      if (lane == 0) { sh_kept = kept_w; sh_nkept = nk_w; sh_done = done_w ? 1 : 0; }
    }
    __syncthreads();
    const int nk0 = nkept;
    nkept = sh_nkept;
    if (sh_done) break;
    if (nkept > nk0 && cb + 1 < L) apply(cb + 1, L, nk0, nkept);
    __syncthreads();
  }
  __syncthreads();
  compact_flags(flag, wsum, n, keep + (long)p * keep_stride, n_keep + p);
}

// ---------------------------------------------------------------------------------
// RPN: sigmoid -> top-k -> delta2bbox, one workgroup per frame
// ---------------------------------------------------------------------------------
// cls  [T][HW*A] logits in (y, x, anchor) order (NHWC conv output == permute(1,2,0))
// reg  [T][HW*A][4]
// out  [T][npre][5], sorted by score (desc) when HW*A > nms_pre, else original order

// delta2bbox (mmdet/core/bbox/transforms.py:78-110) of anchor i (= cell * A + a, generated on the fly from the base anchors) with
// its four deltas d and objectness logit lgt -> o[0..5) = clipped box + sigmoid score.  One body for the one-workgroup and the
// chip-wide selection kernels: the same expressions, the same contractions, the same bits.
__device__ __forceinline__ void rpn_decode_one(const RpnParams& rp, int i, const float d[4], float lgt, float* __restrict__ o) {
  const int a = i % rp.A, cell = i / rp.A, x = cell % rp.W, y = cell / rp.W;
  const float ax1 = rp.base[a][0] + x * rp.stride, ay1 = rp.base[a][1] + y * rp.stride;
  const float ax2 = rp.base[a][2] + x * rp.stride, ay2 = rp.base[a][3] + y * rp.stride;
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
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = 1.f / (1.f + expf(-lgt));   // torch.sigmoid, as score_of
}

template <typename T>
__global__ __launch_bounds__(1024) void rpn_select_kernel(const T* __restrict__ cls, const T* __restrict__ reg, long cls_stride,
                                                          long reg_stride, float* __restrict__ out, const RpnParams rp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = blockIdx.x, n = rp.n_anchor, k = rp.npre;
  const T* c = cls + (long)f * cls_stride;
  const T* r = reg + (long)f * reg_stride;
  unsigned long long* comp = reinterpret_cast<unsigned long long*>(smem);  // [kp2] key << 32 | ~index
#ifdef HVR_DBG_SEL_CLK
  long long dbgc[6] = {0, 0, 0, 0, 0, 0};
#endif
  const int kp2 = next_pow2(k);
  __shared__ int whist[16][256];  // one histogram per wave: no cross-wave contention
  __shared__ int tot[256];
  __shared__ int wsum[17];
  __shared__ int tcnt[32 * 16];
  __shared__ uint32_t sh_prefix, sh_remaining;
  __shared__ int sh_count;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  // anchor i = cell * A + a lives at pixel `cell`, channel a of a [H*W][cls_pitch] map (pitch >= A lets the
  // objectness and delta maps be channel slices of one fused conv output)
  const int A = rp.A, cpitch = rp.cls_pitch, rpitch = rp.reg_pitch;
  auto score_of = [&](int i) {
    const int cell = i / A, a = i - cell * A;
    return 1.f / (1.f + expf(-ElemTraits<T>::load(c + (long)cell * cpitch + a)));  // torch.sigmoid
  };

#ifdef HVR_DBG_SEL_CLK
  dbgc[0] = clock64();
#endif
  if (n > k) {
    // radix select (MSB first) of the k-th largest (key, then smallest index) element.  The scores are sigmoids: the top
    // byte of almost every key is one of a handful of values, and an atomicAdd per element serialises on those few LDS
    // words (4 passes x 28 728 adds took 0.4 ms of this kernel's 0.53).  Here the lanes of a wave that hit the same bin are
    // counted with a ballot and added once (up to 4 distinct bins per 64 elements; what is left -- the evenly spread lower
    // bytes of the later passes -- goes to the wave's own histogram with plain LDS atomics, which then rarely collide).
    // Up to CE * 1024 anchors (32 768; the C4 map of a 1000 x 600 frame has 28 728) the keys are computed ONCE and stay in
    // registers for the seven scans (four histogram passes, the collection, two for the ties); longer inputs recompute them.
    constexpr int CE = 32;
    const bool cached = blockDim.x == 1024 && n <= CE * 1024;
    const int mcount = (n + 1023) >> 10;
    uint32_t uc[CE];
    if (cached) {
      // all of a thread's logits are requested before the first is used (indices past n are clamped, not predicated): under
      // `i < n ? load : 0` every one of the 29 strided 2-byte loads was its own round trip (load, s_waitcnt vmcnt(0), next)
      float lg[CE];
#pragma unroll
      for (int m = 0; m < CE; ++m) {
        int i = m * 1024 + (int)threadIdx.x;
        i = i < n ? i : n - 1;
        const int cell = i / A, a = i - cell * A;
        lg[m] = ElemTraits<T>::load(c + (long)cell * cpitch + a);
      }
#pragma unroll
      for (int m = 0; m < CE; ++m) {
        const int i = m * 1024 + (int)threadIdx.x;
        uc[m] = i < n ? float_key(1.f / (1.f + expf(-lg[m]))) : 0u;   // torch.sigmoid, as score_of
      }
    }
    int* wh = whist[wave & 15];
    auto hist_add = [&](bool valid, int bin) {
      unsigned long long todo = __ballot(valid);
      for (int it = 0; it < 4 && todo; ++it) {
        const int leader = __builtin_ctzll(todo);
        const int lb = __builtin_amdgcn_readlane(bin, leader);
        const unsigned long long same = __ballot(valid && bin == lb);
        if (lane == leader) atomicAdd(&wh[lb], (int)__popcll(same));
        todo &= ~same;
      }
      if ((todo >> lane) & 1ull) atomicAdd(&wh[bin], 1);
    };
#ifdef HVR_DBG_SEL_CLK
  dbgc[1] = clock64();
#endif
    uint32_t prefix = 0u, remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) (&whist[0][0])[i] = 0;
      __syncthreads();
      const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      if (cached) {
#pragma unroll
        for (int m = 0; m < CE; ++m) {
          if (m < mcount) {
            const int i = m * 1024 + (int)threadIdx.x;
            const uint32_t u = uc[m];
            hist_add(i < n && (u & himask) == prefix, (int)((u >> shift) & 255u));
          }
        }
      } else {
        for (int i0 = wave * 64; i0 < n; i0 += blockDim.x) {
          const int i = i0 + lane;
          uint32_t u = 0u;
          if (i < n) u = float_key(score_of(i));
          hist_add(i < n && (u & himask) == prefix, (int)((u >> shift) & 255u));
        }
      }
      __syncthreads();
      if (threadIdx.x < 256) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += whist[w][threadIdx.x];
        tot[threadIdx.x] = t;
      }
      __syncthreads();
      if (wave == 0) {
        // bins from 255 down until `remaining` elements are covered: lane l owns bins 4 l .. 4 l + 3, suffix sums by shuffles
        const int t0 = tot[4 * lane], t1 = tot[4 * lane + 1], t2 = tot[4 * lane + 2], t3 = tot[4 * lane + 3];
        int suf = t0 + t1 + t2 + t3;
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_down(suf, o);
          if (lane + o < 64) suf += v;
        }
        const unsigned long long ge = __ballot((uint32_t)suf >= remaining);
        const int ls = ge ? 63 - __builtin_clzll(ge) : 0;         // the highest lane whose suffix still covers `remaining`
        const int above = __shfl(suf - (t0 + t1 + t2 + t3), ls);  // elements in the bins above lane ls's four
        if (lane == ls) {
          uint32_t rem = remaining - (uint32_t)above;
          int b = 3;
          if ((uint32_t)t3 < rem) {
            rem -= (uint32_t)t3; b = 2;
            if ((uint32_t)t2 < rem) {
              rem -= (uint32_t)t2; b = 1;
              if ((uint32_t)t1 < rem) { rem -= (uint32_t)t1; b = 0; }
            }
          }
          sh_prefix = prefix | ((uint32_t)(4 * ls + b) << shift);
          sh_remaining = rem;
        }
      }
      __syncthreads();
      prefix = sh_prefix;
      remaining = sh_remaining;
      __syncthreads();
    }
#ifdef HVR_DBG_SEL_CLK
  dbgc[2] = clock64();
#endif
    // prefix == key of the k-th element; `remaining` of the equal-key elements are taken, lowest index first
    const uint32_t thr_key = prefix;
    if (threadIdx.x == 0) sh_count = 0;
    for (int i = threadIdx.x; i < kp2; i += blockDim.x) comp[i] = 0ull;
    __syncthreads();
    auto collect = [&](int i, uint32_t u) {
      const bool sel = i < n && u > thr_key;
      const unsigned long long m = __ballot(sel);
      if (m) {  // one slot reservation per wave and step (the order inside comp is irrelevant: it is sorted below)
        int base = 0;
        if (lane == 0) base = atomicAdd(&sh_count, (int)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (sel) comp[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)u << 32) | (0xffffffffu - (uint32_t)i);
      }
    };
    if (cached) {
#pragma unroll
      for (int m = 0; m < CE; ++m)
        if (m < mcount) collect(m * 1024 + (int)threadIdx.x, uc[m]);
    } else {
      for (int i0 = wave * 64; i0 < n; i0 += blockDim.x) {
        const int i = i0 + lane;
        collect(i, i < n ? float_key(score_of(i)) : 0u);
      }
    }
    __syncthreads();
#ifdef HVR_DBG_SEL_CLK
  dbgc[3] = clock64();
#endif
    // ties at the threshold, lowest index first (bf16 logits tie often)
    if (cached) {
      // index order = (m, wave, lane): per-(m, wave) tie counts, one exclusive scan over the 512 of them, ranks by ballot
#pragma unroll
      for (int m = 0; m < CE; ++m) {
        const int i = m * 1024 + (int)threadIdx.x;
        const unsigned long long mm = __ballot(i < n && uc[m] == thr_key);
        if (lane == 0) tcnt[m * 16 + wave] = (int)__popcll(mm);
      }
      __syncthreads();
      if (wave == 0) {
        int c8[8], sum = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { c8[e] = tcnt[8 * lane + e]; sum += c8[e]; }
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o);
          if (lane >= o) incl += v;
        }
        int run = incl - sum;
#pragma unroll
        for (int e = 0; e < 8; ++e) { tcnt[8 * lane + e] = run; run += c8[e]; }
      }
      __syncthreads();
      const int base = sh_count;
#pragma unroll
      for (int m = 0; m < CE; ++m) {
        if (m < mcount) {
          const int i = m * 1024 + (int)threadIdx.x;
          const bool eq = i < n && uc[m] == thr_key;
          const unsigned long long mm = __ballot(eq);
          const int rank = tcnt[m * 16 + wave] + (int)__popcll(mm & ((1ull << lane) - 1ull));
          if (eq && rank < (int)remaining) comp[base + rank] = ((unsigned long long)thr_key << 32) | (0xffffffffu - (uint32_t)i);
        }
      }
    } else {  // every thread ranks a contiguous index segment
      const int per = (n + blockDim.x - 1) / blockDim.x;
      const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
      int cnt = 0;
      for (int i = lo; i < hi; ++i) cnt += float_key(score_of(i)) == thr_key;
      int incl = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int rank = incl - cnt;
      for (int w = 0; w < wave; ++w) rank += wsum[w];
      const int base = sh_count;
      if (cnt > 0 && rank < (int)remaining) {
        for (int i = lo; i < hi && rank < (int)remaining; ++i)
          if (float_key(score_of(i)) == thr_key) {
            comp[base + rank] = ((unsigned long long)thr_key << 32) | (0xffffffffu - (uint32_t)i);
            ++rank;
          }
      }
    }
#ifdef HVR_DBG_SEL_CLK
  dbgc[4] = clock64();
#endif
    __syncthreads();
    block_sort_desc_u64(comp, kp2);
  } else {
    for (int i = threadIdx.x; i < k; i += blockDim.x) comp[i] = 0xffffffffu - (uint32_t)i;
    __syncthreads();
  }
#ifdef HVR_DBG_SEL_CLK
  dbgc[5] = clock64();
#endif
  // decode (delta2bbox, mmdet/core/bbox/transforms.py:78-110) in sorted order
  const int cnt = n > k ? k : n;
  // four proposals per thread and step: their deltas and logits are requested together (clamped slots, nothing predicated) -- one
  // proposal per iteration made every one of a thread's six iterations a round trip of its own
  constexpr int DU = 4;
  for (int j0 = threadIdx.x; j0 < cnt; j0 += DU * blockDim.x) {
    int ii[DU];
    float d[DU][4], lgt[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      int j = j0 + u * (int)blockDim.x;
      j = j < cnt ? j : cnt - 1;
      ii[u] = (int)(0xffffffffu - (uint32_t)comp[j]);
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int a = ii[u] % rp.A, cell = ii[u] / rp.A;
      load4(r + (long)cell * rpitch + a * 4, d[u]);
      lgt[u] = ElemTraits<T>::load(c + (long)cell * cpitch + a);
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int j = j0 + u * (int)blockDim.x;
      if (j >= cnt) continue;
      rpn_decode_one(rp, ii[u], d[u], lgt[u], out + ((long)f * k + j) * 5);
    }
  }
#ifdef HVR_DBG_SEL_CLK
  if (threadIdx.x == 0 && f == 0 && n > k)
    printf("rpn_select clocks: keys %lld  hist %lld  collect %lld  ties %lld  sort %lld  decode %lld\n", dbgc[1] - dbgc[0], dbgc[2] - dbgc[1],
           dbgc[3] - dbgc[2], dbgc[4] - dbgc[3], dbgc[5] - dbgc[4], (long long)clock64() - dbgc[5]);
#endif
}

// first nms_post survivors (ascending index), then topk(min(max_num, count)) by score
// (rpn_head.py:92-103).  props [T][npre][5]; keep [T][npre]; out [T][max_num][5]; rois [T][max_num][5]
__global__ __launch_bounds__(256) void rpn_gather_kernel(const float* __restrict__ props, const long long* __restrict__ keep,
                                                         const int* __restrict__ n_keep, int npre, int nms_post, int max_num,
                                                         int presorted, float* __restrict__ out, int* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int f = blockIdx.x;
  int cnt = n_keep[f];
  cnt = cnt < nms_post ? cnt : nms_post;
  const int np2 = next_pow2(cnt > 1 ? cnt : 1);
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);
  uint32_t* idx = key + np2;
  const float* pf = props + (long)f * npre * 5;
  const long long* kf = keep + (long)f * npre;
  // props sorted by (score desc, anchor asc) and survivors listed by ascending position: the list IS in (score desc, position
  // asc) order already, the network below would return the identity
  if (!presorted) {
    for (int i = threadIdx.x; i < np2; i += blockDim.x) {
      if (i < cnt) { key[i] = float_key(pf[kf[i] * 5 + 4]); idx[i] = (uint32_t)i; }
      else { key[i] = 0u; idx[i] = 0xffffffffu; }
    }
    __syncthreads();
    bitonic_sort_pairs(key, idx, np2);
  }
  const int num = cnt < max_num ? cnt : max_num;
  for (int j = threadIdx.x; j < num * 5; j += blockDim.x) {
    const int row = j / 5, col = j - row * 5;
    out[((long)f * max_num + row) * 5 + col] = pf[kf[presorted ? row : (int)idx[row]] * 5 + col];
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

// ---------------------------------------------------------------------------------
// RPN proposals for FEW frames per call (stream mode: one new frame per output frame): the chip-wide form
// ---------------------------------------------------------------------------------
// The one-workgroup-per-frame kernels above leave 255 of the 256 CUs idle when a call brings one frame, and their time
// (select 0.19 ms + greedy NMS 0.19 ms) sits on the critical path of every output frame (tools/test.py:214-250).  Same results,
// bit for bit, from launches that cover the chip:
//   rpn_wide_hist     keys (sigmoid -> sortable u32) of 1024 anchors per workgroup, histogram of their top 12 bits
//   rpn_wide_scatter  every workgroup scans the 4096-bin histogram (suffix sums = where a bin's elements start in score order,
//                     b0 = the bin that holds the nms_pre-th element) and scatters the elements of the bins >= b0 into their
//                     bin's slot range (order inside a bin: arrival) -- a counting sort on the top 12 key bits
//   rpn_wide_rank     rank of a candidate = start of its bin + the candidates of ITS BIN that come before it (higher key, or
//                     equal key and lower anchor index: the tie rule of the bitonic network above); rank < nms_pre -> decode
//                     (delta2bbox) and store at out[rank].  Work: sum over bins of size^2 instead of a sort.
//   nms_band_mask     suppression bit mask of the first R0 <= 4096 score-sorted boxes, transposed (who suppresses box i)
//   nms_band_sweep    one workgroup per frame, a pipeline of eight waves over the 64-box chunks (see the kernel); stops at
//                     max_keep survivors.  A frame whose max_keep-th survivor lies behind box R0 is handed to
//                     nms_greedy_kernel (skip flag clear).
constexpr int WBINS = 4096, WSEL_THREADS = 256, WSEL_PER_WG = 1024;

// Histogram bin of a score key (float_key of a sigmoid: a float in [0, 1]), monotone non-decreasing in the key -- all the counting
// sort needs.  The key's top bits alone would give [0.5, 1) eight bins (one exponent), and that is where the nms_pre best scores
// of a frame live: the upper 2048 bins split [0.5, 1] linearly (2^-12 each), the lower 2048 take the top 16 bits of the float
// (128 steps per octave) from 0.5 down to 2^-17, everything below shares bin 0.
__device__ __forceinline__ int wide_bin(uint32_t u) {
  if (!(u & 0x80000000u)) return 0;   // (negative floats: no sigmoid is)
  const uint32_t bits = u & 0x7fffffffu;
  if (bits >= 0x3f000000u) return 2048 + (int)min((bits - 0x3f000000u) >> 12, 2047u);
  const int b = (int)(bits >> 16) - (0x3f00 - 2048);
  return b > 0 ? b : 0;
}

__device__ __forceinline__ void wave_hist_add(int* __restrict__ h, bool valid, int bin, int lane) {
  unsigned long long todo = __ballot(valid);
  for (int it = 0; it < 4 && todo; ++it) {   // lanes that hit the same bin are counted with a ballot and added once
    const int leader = __builtin_ctzll(todo);
    const int lb = __builtin_amdgcn_readlane(bin, leader);
    const unsigned long long same = __ballot(valid && bin == lb);
    if (lane == leader) atomicAdd(&h[lb], (int)__popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1ull) atomicAdd(&h[bin], 1);
}

template <typename T>
__global__ __launch_bounds__(WSEL_THREADS) void rpn_wide_hist_kernel(const T* __restrict__ cls, long cls_stride, int n, int A,
                                                                    int cpitch, int* __restrict__ hist0) {
  __shared__ int lh[WBINS];
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  for (int b = tid; b < WBINS; b += WSEL_THREADS) lh[b] = 0;
  const T* c = cls + (long)f * cls_stride;
  const int base = blockIdx.x * WSEL_PER_WG;
  constexpr int E = WSEL_PER_WG / WSEL_THREADS;
  float lg[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    int i = base + e * WSEL_THREADS + tid;
    i = i < n ? i : n - 1;
    const int cell = i / A, a = i - cell * A;
    lg[e] = ElemTraits<T>::load(c + (long)cell * cpitch + a);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = base + e * WSEL_THREADS + tid;
    const uint32_t u = float_key(1.f / (1.f + expf(-lg[e])));   // torch.sigmoid, as rpn_select_kernel
    wave_hist_add(lh, i < n, wide_bin(u), lane);
  }
  __syncthreads();
  for (int b = tid; b < WBINS; b += WSEL_THREADS) {
    const int v = lh[b];
    if (v) atomicAdd(&hist0[(long)f * WBINS + b], v);
  }
}

// meta[f] = {b0, candidates, 0, 0}
template <typename T>
__global__ __launch_bounds__(WSEL_THREADS) void rpn_wide_scatter_kernel(const T* __restrict__ cls, long cls_stride, int n, int A,
                                                                       int cpitch, int k, const int* __restrict__ hist0,
                                                                       int* __restrict__ gcount, int* __restrict__ binstart,
                                                                       int* __restrict__ meta, unsigned long long* __restrict__ cand) {
  __shared__ int start[WBINS];
  __shared__ int lcnt[WBINS];
  __shared__ int wtot[WSEL_THREADS / 64];
  __shared__ int sh_b0;
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int BPT = WBINS / WSEL_THREADS;   // bins per thread
  const int* hf = hist0 + (long)f * WBINS;
  const T* c = cls + (long)f * cls_stride;
  const int base = blockIdx.x * WSEL_PER_WG;
  constexpr int E = WSEL_PER_WG / WSEL_THREADS;
  float lg[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    int i = base + e * WSEL_THREADS + tid;
    i = i < n ? i : n - 1;
    const int cell = i / A, a = i - cell * A;
    lg[e] = ElemTraits<T>::load(c + (long)cell * cpitch + a);
  }
  int h[BPT], sum = 0;
#pragma unroll
  for (int e = 0; e < BPT; ++e) { h[e] = hf[BPT * tid + e]; sum += h[e]; }
  int suf = sum;   // elements in this thread's bins and in all higher ones
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_down(suf, o);
    if (lane + o < 64) suf += v;
  }
  if (lane == 0) wtot[wave] = suf;
#pragma unroll
  for (int e = 0; e < BPT; ++e) lcnt[BPT * tid + e] = 0;
  __syncthreads();
  for (int w = wave + 1; w < WSEL_THREADS / 64; ++w) suf += wtot[w];
  int run = suf - sum;   // elements in the bins above this thread's
#pragma unroll
  for (int e = BPT - 1; e >= 0; --e) {
    start[BPT * tid + e] = run;
    if (run < k && k <= run + h[e]) sh_b0 = BPT * tid + e;   // exactly one bin (n > k)
    run += h[e];
  }
  __syncthreads();
  const int b0 = sh_b0;
  uint32_t u[E];
  int loc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = base + e * WSEL_THREADS + tid;
    u[e] = float_key(1.f / (1.f + expf(-lg[e])));
    const int b = wide_bin(u[e]);
    loc[e] = (i < n && b >= b0) ? atomicAdd(&lcnt[b], 1) : -1;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < BPT; ++e) {
    const int b = BPT * tid + e, cnt = lcnt[b];
    if (cnt > 0) lcnt[b] = atomicAdd(&gcount[(long)f * WBINS + b], cnt);   // this workgroup's range inside the bin
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (loc[e] < 0) continue;
    const int i = base + e * WSEL_THREADS + tid;
    const int b = wide_bin(u[e]);
    cand[(long)f * n + start[b] + lcnt[b] + loc[e]] = ((unsigned long long)u[e] << 32) | (0xffffffffu - (uint32_t)i);
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int e = 0; e < BPT; ++e) binstart[(long)f * WBINS + BPT * tid + e] = start[BPT * tid + e];
    if (tid == 0) { meta[f * 4 + 0] = b0; meta[f * 4 + 1] = start[b0] + hf[b0]; }
  }
}

constexpr int WRANK_TILE = 2048;
template <typename T>
__global__ __launch_bounds__(256) void rpn_wide_rank_kernel(const T* __restrict__ cls, const T* __restrict__ reg, long cls_stride,
                                                            long reg_stride, const int* __restrict__ hist0,
                                                            const int* __restrict__ binstart, const int* __restrict__ meta,
                                                            const unsigned long long* __restrict__ cand, float* __restrict__ out,
                                                            const RpnParams rp) {
  __shared__ unsigned long long tile[WRANK_TILE];
  __shared__ int sh_we[4];
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = rp.n_anchor, k = rp.npre;
  const int C = meta[f * 4 + 1];
  const int p0 = blockIdx.x * 256;
  if (p0 >= C) return;
  const int p = p0 + tid;
  const bool valid = p < C;
  const unsigned long long* cf = cand + (long)f * n;
  const unsigned long long comp = valid ? cf[p] : 0ull;
  int s = 0, e = 0;
  if (valid) {
    const int b = wide_bin((uint32_t)(comp >> 32));
    s = binstart[(long)f * WBINS + b];
    e = s + hist0[(long)f * WBINS + b];
    e = e < C ? e : C;   // (the threshold bin is the last one: its range ends at C by construction)
  }
  // Slots ascend with descending bins.  A wave's 64 candidates: everything in front of its first lane's bin is ahead of all
  // of them, everything behind its last lane's bin behind all of them, so ONE range [ws, we) serves the wave -- rank = ws +
  // the elements of that range that compare greater (higher key, or the same key and a lower anchor index).  Uniform loop
  // bounds, every LDS read a broadcast.
  if (lane == 0) sh_we[wave] = 0;
  __syncthreads();
  if (valid && (p == C - 1 || lane == 63)) sh_we[wave] = e;
  __syncthreads();
  const int ws = __builtin_amdgcn_readfirstlane(s), we = sh_we[wave];
  int cnt = 0;
  __shared__ int sh_lo;
  if (tid == 0) sh_lo = s;
  __syncthreads();
  const int blo = sh_lo;
  int bhi = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) bhi = max(bhi, sh_we[w]);
  for (int t0 = blo; t0 < bhi; t0 += WRANK_TILE) {
    const int tn = min(WRANK_TILE, bhi - t0);
    for (int j = tid; j < tn; j += 256) tile[j] = cf[t0 + j];
    __syncthreads();
    const int js = max(ws, t0) - t0, je = min(we, t0 + tn) - t0;
    int j = js;
    for (; j + 8 <= je; j += 8) {
      unsigned long long v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tile[j + q];
#pragma unroll
      for (int q = 0; q < 8; ++q) cnt += v[q] > comp ? 1 : 0;
    }
    for (; j < je; ++j) cnt += tile[j] > comp ? 1 : 0;
    __syncthreads();
  }
  const int rank = ws + cnt;
  if (!valid || rank >= k) return;
  const int i = (int)(0xffffffffu - (uint32_t)comp);
  const int a = i % rp.A, cell = i / rp.A;
  const T* c = cls + (long)f * cls_stride;
  const T* r = reg + (long)f * reg_stride;
  float d[4];
  load4(r + (long)cell * rp.reg_pitch + a * 4, d);
  const float lgt = ElemTraits<T>::load(c + (long)cell * rp.cls_pitch + a);
  rpn_decode_one(rp, i, d, lgt, out + ((long)f * k + rank) * 5);
}

// maskT [P][nb0 chunks][nb0 words][64] over the first R0 score-sorted boxes of dets [P][n][5], TRANSPOSED: word w of box i = the
// boxes j of chunk w (j < i) that suppress box i.  Only words w <= i / 64 are written (and read).  IoU is symmetric in its arguments (box_iou_hits:
// min / max and one commutative sum), so this is the bit matrix of nms_mask_kernel read by columns.
__global__ __launch_bounds__(64) void nms_band_mask_kernel(const float* __restrict__ dets, int n, int R0, float thr, int ge,
                                                           unsigned long long* __restrict__ maskT) {
  // blockIdx.x = linear index of the (rb <= cb) block pairs: t = cb (cb + 1) / 2 + rb
  const int p = blockIdx.y, t = blockIdx.x;
  int cb = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
  while (cb * (cb + 1) / 2 > t) --cb;
  const int rb = t - cb * (cb + 1) / 2;
  const int nb0 = (R0 + 63) >> 6;
  const float* d = dets + (long)p * n * 5;
  __shared__ float4 rowbox[64];
  __shared__ float rowarea[64];
  const int rj = rb * 64 + (int)threadIdx.x;   // (< R0: rb <= cb and chunk cb starts inside the band)
  const int rjc = min(rj, R0 - 1);
  const float4 rbx = make_float4(d[rjc * 5 + 0], d[rjc * 5 + 1], d[rjc * 5 + 2], d[rjc * 5 + 3]);
  rowbox[threadIdx.x] = rbx;
  rowarea[threadIdx.x] = box_area_plus1(rbx);
  const int i = cb * 64 + threadIdx.x;
  const int ic = min(i, R0 - 1);
  const float4 me = make_float4(d[ic * 5 + 0], d[ic * 5 + 1], d[ic * 5 + 2], d[ic * 5 + 3]);
  const float my_area = box_area_plus1(me);
  const float band_k = iou_band_k(thr);
  __syncthreads();
  if (i >= R0) return;
  const int jend = (rb == cb) ? (int)threadIdx.x : 64;   // earlier boxes only (every box of an earlier chunk is)
  unsigned long long bits = 0ull;
#pragma unroll 4
  for (int j = 0; j < 64; ++j) {
    const bool hit = box_iou_hits(rowbox[j], rowarea[j], me, my_area, thr, band_k, ge);   // (row = the earlier box, as the sweeps above)
    if (hit && j < jend) bits |= 1ull << j;
  }
  maskT[(((long)p * nb0 + cb) * nb0 + rb) * 64 + threadIdx.x] = bits;   // [chunk][word][box of the chunk]: coalesced for the sweep
}

// One workgroup of sixteen waves per problem, a pipeline over the 64-box chunks: wave w takes chunks w, w + 16, ...  A chunk's
// verdict needs (a) for every box the survivors of the EARLIER chunks that suppress it -- (row of maskT) AND (survivor words), all
// of it known except the last fifteen chunks' words when the wave starts (it published chunk cb - 16 itself), so those mask words
// stream in while the chunks in between are being resolved by the other waves; (b) the chunk's own 64 x 64 block: the greedy rule
// K_i = alive_i and no j < i in K suppresses i, iterated from K = alive until it stops changing (bit i is final once the bits
// below it are; the fixed point is unique = the serial sweep's survivors) -- each round two ANDs and a ballot instead of a
// scalar chain over the survivors.  Hand-over between waves through LDS (survivor word, running count, chunks-done counter).
constexpr int SWEEP_WAVES = 16, SWEEP_FIN = 1 << 20;
__global__ __launch_bounds__(64 * SWEEP_WAVES) void nms_band_sweep_kernel(const unsigned long long* __restrict__ maskT, int n, int R0,
                                                                          int max_keep, long long* __restrict__ keep,
                                                                          int* __restrict__ n_keep, int keep_stride,
                                                                          int* __restrict__ done) {
  __shared__ unsigned long long kept[64];
  __shared__ int sh_done, sh_nkept;
#ifdef HVR_DBG_SWEEP_CLK   // (tools/build_dbg.sh, NMS_DBG=1: per-chunk clock stamps of the hand-over chain, printed for problem 0)
  __shared__ long long dbg_clk[64][5];
#endif
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb0 = (R0 + 63) >> 6;   // <= 64
  const unsigned long long* mk = maskT + (long)p * nb0 * nb0 * 64;
  long long* kp = keep + (long)p * keep_stride;
  if (tid == 0) { sh_done = 0; sh_nkept = 0; }
  __syncthreads();
  auto chunks_done = [&]() { return __hip_atomic_load(&sh_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  for (int cb = wave; cb < nb0; cb += SWEEP_WAVES) {
    const int i = cb * 64 + lane;
    const bool inb = i < R0;
    const unsigned long long* row = mk + (long)cb * nb0 * 64 + lane;   // word w of this lane's box: row[w * 64]
    constexpr int TW = SWEEP_WAVES;   // tail words: the chunks that may still be open when the wave arrives
    const int t0 = cb > TW - 1 ? cb - (TW - 1) : 0;
    unsigned long long tailw[TW];   // words t0 .. cb (the last one: the chunk's own block)
#pragma unroll
    for (int q = 0; q < TW; ++q) tailw[q] = (t0 + q <= cb) ? row[(t0 + q) * 64] : 0ull;
    unsigned long long sup = 0ull;
    // words [0, t0): chunks resolved before this wave arrived here
    for (int w0 = 0; w0 < t0; w0 += 16) {
      unsigned long long r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = (w0 + q < t0) ? row[(w0 + q) * 64] : 0ull;
#pragma unroll
      for (int q = 0; q < 16; ++q) sup |= r[q] & kept[(w0 + q) & 63];
    }
    int d = 0;
    unsigned long long cdiag = 0ull;
#pragma unroll
    for (int q = 0; q < TW; ++q) {
      const int w = t0 + q;
      if (w < cb) {
        while ((d = chunks_done()) <= w) __builtin_amdgcn_s_sleep(1);
        sup |= tailw[q] & kept[w];
      } else if (w == cb) {
        cdiag = tailw[q];
      }
    }
    if (d >= SWEEP_FIN) break;
#ifdef HVR_DBG_SWEEP_CLK
    const long long c0 = clock64();
#endif
    const unsigned long long alive = __ballot(inb && sup == 0ull);
    unsigned long long K = alive;
#ifdef HVR_DBG_SWEEP_CLK
    const long long c1 = clock64();
    int rounds = 0;
#endif
    while (true) {
      const unsigned long long Kn = alive & ~__ballot((cdiag & K) != 0ull);
#ifdef HVR_DBG_SWEEP_CLK
      ++rounds;
#endif
      if (Kn == K) break;
      K = Kn;
    }
#ifdef HVR_DBG_SWEEP_CLK
    const long long c2 = clock64();
#endif
    const int nk0 = cb > 0 ? *(volatile int*)&sh_nkept : 0;
    bool fin = false;
    int below = (int)__popcll(K & ((1ull << lane) - 1ull));
    if (nk0 + (int)__popcll(K) >= max_keep) {   // the cap falls inside this chunk: its first max_keep - nk0 survivors count
      K = __ballot(((K >> lane) & 1ull) && below < max_keep - nk0);
      fin = true;
    }
    const int nk1 = nk0 + (int)__popcll(K);
    const bool last = fin || cb == nb0 - 1;
    // hand over FIRST, survivor list's global stores behind it.  Observed: with the kp store in front, the compiler's release
    // sequence is `s_waitcnt vmcnt(0) lgkmcnt(0)` + ds_write (build/nms-*.s) and the kernel took 34.6 us on the bench model's
    // frame (41 chunks walked); with the store behind the release 23.6 us (one A/B, same box).  Likely reason: the next chunk's
    // wave spins on sh_done while this one waits for the store's acknowledgement -- not isolated (the edit also moved the cap
    // logic; a timing build without the store would tell), so no latency figure is claimed here.
    if (lane == 0) {
      kept[cb] = K;
      *(volatile int*)&sh_nkept = nk1;
      __hip_atomic_store(&sh_done, last ? SWEEP_FIN : cb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef HVR_DBG_SWEEP_CLK
    if (lane == 0) { dbg_clk[cb][0] = c0; dbg_clk[cb][1] = c1; dbg_clk[cb][2] = c2; dbg_clk[cb][3] = clock64(); dbg_clk[cb][4] = rounds; }
    if (last && lane == 0 && p == 0) {
      __builtin_amdgcn_s_sleep(100);
      for (int c = 0; c <= cb; ++c)
        printf("sweep chunk %2d: wait-end->alive %4lld  fixpoint %4lld (%lld rounds)  publish %4lld  | since previous publish %5lld\n", c,
               dbg_clk[c][1] - dbg_clk[c][0], dbg_clk[c][2] - dbg_clk[c][1], dbg_clk[c][4], dbg_clk[c][3] - dbg_clk[c][2],
               c ? dbg_clk[c][0] - dbg_clk[c - 1][3] : 0ll);
    }
#endif
    if ((K >> lane) & 1ull) kp[nk0 + below] = i;
    if (lane == 0) {
      if (last) {
        const bool complete = fin || R0 >= n;
        done[p] = complete ? 1 : 0;
        if (complete) n_keep[p] = nk1;
      }
    }
    if (last) break;
  }
}

// ---------------- launchers ----------------
size_t nms_workspace_bytes(int P, int n) {
  const size_t nb = (n + 63) / 64;
  size_t b = 0;
  b += ((size_t)P * n * sizeof(int) + 255) & ~(size_t)255;                    // order
  b += ((size_t)P * n * sizeof(float4) + 255) & ~(size_t)255;                 // sorted boxes
  b += ((size_t)P * nb * 64 * nb * sizeof(unsigned long long) + 255) & ~(size_t)255;  // mask (whole 64-box chunks: the band form)
  return b;
}

// band > 0 (score-sorted input, a survivor cap): mask of the first `band` boxes chip-wide + a one-wave sweep per problem; the greedy
// kernel then only runs the problems the sweep could not finish inside the band (done flags, `band_done` [P])
hipError_t run_nms_batched(const float* dets, int P, int n, float thr, int ge, int presorted, int max_keep,
                           long long* keep, int* n_keep, void* ws, hipStream_t s, int band, int* band_done) {
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
  static std::atomic<unsigned> attr_dev{0};   // (the attribute is per device)
  per_device_once(attr_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
  });
  // a small survivor cap (the RPN's nms_post): the greedy kernel prices max_keep x n IoUs instead of n^2 / 2
  static const bool no_greedy = std::getenv("HVR_NMS_MASK") != nullptr;
  if (max_keep > 0 && max_keep <= 1024 && !no_greedy) {
    static std::atomic<unsigned> gattr_dev{0};   // (the attribute is per device)
    per_device_once(gattr_dev, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_greedy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    });
    if (!presorted) hipLaunchKernelGGL(nms_sort_kernel, dim3(P), dim3(1024), (size_t)np2 * 8, s, dets, n, 0, order, boxes4);
    const int* skip = nullptr;
    if (band > 0 && presorted && band_done) {
      const int R0 = n < band ? n : (band < 4096 ? band : 4096), nb0 = (R0 + 63) / 64;
      hipLaunchKernelGGL(nms_band_mask_kernel, dim3(nb0 * (nb0 + 1) / 2, P), dim3(64), 0, s, dets, n, R0, thr, ge, mask);
      hipLaunchKernelGGL(nms_band_sweep_kernel, dim3(P), dim3(64 * SWEEP_WAVES), 0, s, mask, n, R0, max_keep, keep, n_keep, n, band_done);
      skip = band_done;
    }
    const size_t lds = (size_t)nb * 64 * 16 + (size_t)nb * 8 + 64 * 8 + ((n + 15) & ~15) + 128 + 2 * 1024;
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(P), dim3(1024), lds, s, presorted ? dets : nullptr, presorted ? nullptr : boxes4,
                       presorted ? nullptr : order, n, thr, ge, max_keep, keep, n_keep, n, skip);
    return hipGetLastError();
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
  static std::atomic<unsigned> attr_dev{0};   // (the attribute is per device)
  per_device_once(attr_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_select_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rpn_select_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
  });
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(rpn_select_kernel<bf16_t>, dim3(rp.T), dim3(1024), lds, s, (const bf16_t*)cls, (const bf16_t*)reg, cls_stride, reg_stride, out, rp);
  else
    hipLaunchKernelGGL(rpn_select_kernel<float>, dim3(rp.T), dim3(1024), lds, s, (const float*)cls, (const float*)reg, cls_stride, reg_stride, out, rp);
  return hipGetLastError();
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
size_t rpn_wide_workspace_bytes(int T, long n_anchor) {
  return al256((size_t)T * WBINS * 4 * 2) + al256((size_t)T * WBINS * 4) + al256((size_t)T * 16) + al256((size_t)T * 4) +
         al256((size_t)T * n_anchor * 8);
}
// calls with at most this many frames take the chip-wide kernels (HVR_RPN_WIDE=<frames>, 0 = never)
int rpn_wide_max_frames(int set) {
  // process-wide (include/hvr_hip.h says so): an atomic, so that a thread changing it never tears a concurrent call's read; a call
  // reads it once, and a captured graph keeps the form that was selected at capture time
  static std::atomic<int> v([] {
    const char* e = std::getenv("HVR_RPN_WIDE");
    return e ? std::atoi(e) : 4;
  }());
  return set >= 0 ? v.exchange(set, std::memory_order_relaxed) : v.load(std::memory_order_relaxed);
}
int* rpn_wide_done_flags(void* wws, int T) {
  return (int*)((char*)wws + al256((size_t)T * WBINS * 4 * 2) + al256((size_t)T * WBINS * 4) + al256((size_t)T * 16));
}

__global__ void rpn_wide_zero_kernel(int* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

hipError_t run_rpn_select_wide(const void* cls, const void* reg, long cls_stride, long reg_stride, float* out, const RpnParams& rp,
                               void* wws, hipStream_t s) {
  const int T = rp.T, n = rp.n_anchor;
  char* w = (char*)wws;
  int* hist0 = (int*)w;                    // [T][WBINS]
  int* gcount = hist0 + (size_t)T * WBINS; // [T][WBINS]  (zeroed with hist0)
  w += al256((size_t)T * WBINS * 4 * 2);
  int* binstart = (int*)w;  w += al256((size_t)T * WBINS * 4);
  int* meta = (int*)w;      w += al256((size_t)T * 16);
  w += al256((size_t)T * 4);               // done flags (rpn_wide_done_flags)
  unsigned long long* cand = (unsigned long long*)w;
  // (a kernel, not hipMemsetAsync: a captured graph whose chain holds a memset node, replayed again behind an ordinary launch on the
  // same stream while its previous replay is still in flight, took a GPU memory fault on ROCm 7.2 -- profiles/r04_graph_memset_fault.txt)
  hipLaunchKernelGGL(rpn_wide_zero_kernel, dim3((T * WBINS * 2 + 255) / 256), dim3(256), 0, s, hist0, T * WBINS * 2);
  const int G = (n + WSEL_PER_WG - 1) / WSEL_PER_WG;
  const float* c = (const float*)cls;
  const float* r = (const float*)reg;
  hipLaunchKernelGGL(rpn_wide_hist_kernel<float>, dim3(G, T), dim3(WSEL_THREADS), 0, s, c, cls_stride, n, rp.A, rp.cls_pitch, hist0);
  hipLaunchKernelGGL(rpn_wide_scatter_kernel<float>, dim3(G, T), dim3(WSEL_THREADS), 0, s, c, cls_stride, n, rp.A, rp.cls_pitch, rp.npre,
                     hist0, gcount, binstart, meta, cand);
  hipLaunchKernelGGL(rpn_wide_rank_kernel<float>, dim3((n + 255) / 256, T), dim3(256), 0, s, c, r, cls_stride, reg_stride, hist0, binstart,
                     meta, cand, out, rp);
  return hipGetLastError();
}

hipError_t run_rpn_gather(const float* props, const long long* keep, const int* n_keep, int T, int npre, int nms_post,
                          int max_num, int presorted, float* out, int* counts, hipStream_t s) {
  int np2 = 1;
  while (np2 < nms_post) np2 <<= 1;
  if (np2 > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rpn_gather_kernel, dim3(T), dim3(256), (size_t)np2 * 8, s, props, keep, n_keep, npre, nms_post, max_num, presorted, out, counts);
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
  static std::atomic<unsigned> attr_dev{0};   // (the attribute is per device)
  per_device_once(attr_dev, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mc_nms_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 3072);
  });
  hipLaunchKernelGGL(mc_nms_merge_kernel, dim3(1), dim3(1024), lds, s, boxes, scores, R, ncls, keepflag, kcount, max_num, sp2, dets, labels, n_out);
  return hipGetLastError();
}

size_t multiclass_nms_workspace_bytes(int R, int ncls) {
  return (((size_t)(ncls - 1) * R + 255) & ~(size_t)255) + (size_t)(ncls - 1) * sizeof(int) + 256;
}

}  // namespace hvr
