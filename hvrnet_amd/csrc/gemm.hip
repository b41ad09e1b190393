// MFMA tile engine for every dense contraction on the HVR forward path:
//   * fc layers / 1x1 convs / Q,K and output projections  -> EPI_LINEAR
//   * 3x3 / strided / dilated convs (implicit GEMM, NHWC)   -> EPI_LINEAR + conv gather
//   * relation scores  P~ = exp(scale*QK^T - tilemax)        -> EPI_SCORES
//   * relation apply   O  = sum_t g[:,t] * (P~_t V_t)        -> EPI_APPLY
//
// Reference op sequence being replaced (no reference kernel exists; the reference
// delegates to ATen/cuDNN): mmdet/models/bbox_heads/selsa_bbox_head.py:156-190,
// mmdet/models/backbones/resnet.py:220-266, mmdet/models/anchor_heads/rpn_head.py:30-35.
//
// Layout: A is [M][K] (or an NHWC activation gathered per filter tap), B is [N][K]
// (weights, K contiguous).  A K-step is 128 bytes of K per row (64 bf16 / 32 f32), so
// the LDS image, the loader and the XOR swizzle are byte-identical for both dtypes;
// only the fragment->MFMA step differs (16x16x32 bf16 vs exact-f32 16x16x4).
// Operands are fed to the MFMA swapped (weights as the "A" operand) so that each lane
// ends up with 4 consecutive n of one output row m -> 8/16-byte row-major stores.
#include "common.h"
#include "gemm_params.h"

namespace hvr {

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
  // one 16-byte chunk per lane = 8 consecutive k
  static __device__ __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                  __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  }
  static constexpr int kChunkSteps = 2;  // chunk reads per 128-byte K-step (2 x 4 lane groups)
};

template <> struct Mma<float> {
  // one 16-byte chunk per lane = 4 consecutive k; the 4 lane groups cover 16 k per
  // read, element i of every lane forms MFMA i (any k permutation is fine as long as
  // both operands use the same one).
  static __device__ __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  }
  static constexpr int kChunkSteps = 2;
};

template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS>
__global__ __launch_bounds__(WM* WN * 64) void tile_kernel(const GemmParams p) {
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16, NT = WM * WN * 64;
  constexpr int EPC = ElemTraits<T>::kPerChunk;  // elements per 16-byte chunk
  constexpr int BKE = 8 * EPC;                   // elements per K-step (128 bytes)
  constexpr int A_SLOTS = (BM * 8 + NT - 1) / NT, B_SLOTS = (BN * 8 + NT - 1) / NT;
  constexpr int STAGE_BYTES = (BM + BN) * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  // ---------------- loader setup: one 16-byte chunk per (thread, slot) ----------------
  const char* a_ptr[A_SLOTS];
  int a_iy[A_SLOTS], a_ix[A_SLOTS];
  const char* b_ptr[B_SLOTS];
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    const int s = i * NT + tid, row = s >> 3, c = (s & 7) ^ (row & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    if (p.conv) {
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
      a_iy[i] = oy * p.stride - p.pad;
      a_ix[i] = ox * p.stride - p.pad;
      a_ptr[i] = (const char*)p.A +
                 ((((long)b * p.H + a_iy[i]) * p.W + a_ix[i]) * (long)p.Cin + c * EPC) * (long)sizeof(T);
    } else {
      a_iy[i] = a_ix[i] = 0;
      a_ptr[i] = (const char*)p.A + ((long)m * p.lda + c * EPC) * (long)sizeof(T);
    }
  }
#pragma unroll
  for (int i = 0; i < B_SLOTS; ++i) {
    const int s = i * NT + tid, row = s >> 3, c = (s & 7) ^ (row & 7);
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;
    b_ptr[i] = (const char*)p.B + ((long)n * p.ldb + c * EPC) * (long)sizeof(T);
  }

  uint4 a_reg[A_SLOTS], b_reg[B_SLOTS];  // register staging (unused when GLDS)

  auto issue_loads = [&](int kt, char* stage) {
    long a_koff;
    int dy = 0, dx = 0;
    if (p.conv) {
      const int k = kt * BKE, tap = k / p.Cin, cin0 = k - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      dy = ky * p.dil;
      dx = kx * p.dil;
      a_koff = (((long)dy * p.W + dx) * p.Cin + cin0) * (long)sizeof(T);
    } else {
      a_koff = (long)kt * 128;
    }
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      if (A_SLOTS * NT == BM * 8 || i * NT + (tid & ~63) < BM * 8) {
        const char* src = a_ptr[i] + a_koff;
        if (p.conv) {
          const bool ok = (unsigned)(a_iy[i] + dy) < (unsigned)p.H && (unsigned)(a_ix[i] + dx) < (unsigned)p.W;
          src = ok ? src : (const char*)p.zero;
        }
        if constexpr (GLDS) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(stage + (i * NT + wave * 64) * 16),
                                           16, 0, 0);
        } else {
          a_reg[i] = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
      if (B_SLOTS * NT == BN * 8 || i * NT + (tid & ~63) < BN * 8) {
        const char* src = b_ptr[i] + (long)kt * 128;
        if constexpr (GLDS) {
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(stage + BM * 128 + (i * NT + wave * 64) * 16), 16, 0, 0);
        } else {
          b_reg[i] = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
  };

  auto commit_stage = [&](char* stage) {
    if constexpr (!GLDS) {
#pragma unroll
      for (int i = 0; i < A_SLOTS; ++i)
        if (A_SLOTS * NT == BM * 8 || i * NT + (tid & ~63) < BM * 8)
          *reinterpret_cast<uint4*>(stage + (i * NT + tid) * 16) = a_reg[i];
#pragma unroll
      for (int i = 0; i < B_SLOTS; ++i)
        if (B_SLOTS * NT == BN * 8 || i * NT + (tid & ~63) < BN * 8)
          *reinterpret_cast<uint4*>(stage + BM * 128 + (i * NT + tid) * 16) = b_reg[i];
    }
  };

  // ---------------- accumulators ----------------
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // EPI_APPLY keeps a second accumulator set: `acc` is the running total, `pacc` the
  // current 128-key block's un-scaled partial product.
  f32x4 pacc[EPI == EPI_APPLY ? FM : 1][EPI == EPI_APPLY ? FN : 1];
  float gcur[EPI == EPI_APPLY ? FM : 1], gnext[EPI == EPI_APPLY ? FM : 1];
  constexpr int STEPS_PER_BLOCK = 128 / BKE;  // K-steps per 128-key statistics block
  int grow[EPI == EPI_APPLY ? FM : 1];
  if constexpr (EPI == EPI_APPLY) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) pacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int m = m0 + (wm * FM + i) * 16 + (lane & 15);
      grow[i] = (m < p.M ? m : p.M - 1) * p.ntile;
      gcur[i] = p.g[grow[i]];
      gnext[i] = 0.f;
    }
  }

  const int nk = p.K / BKE;
  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;

  // ---------------- main loop: double-buffered LDS, one barrier per K-step ----------------
  issue_loads(0, smem);
  commit_stage(smem);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nk;
    if (more) issue_loads(kt + 1, nxt);
    if constexpr (EPI == EPI_APPLY) {
      if ((kt % STEPS_PER_BLOCK) == 0 && kt + STEPS_PER_BLOCK < nk) {
#pragma unroll
        for (int i = 0; i < FM; ++i) gnext[i] = p.g[grow[i] + kt / STEPS_PER_BLOCK + 1];
      }
    }

    const char* a_base = cur + (wm * FM * 16 + frag_row) * 128;
    const char* b_base = cur + BM * 128 + (wn * FN * 16 + frag_row) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = ((kk * 4 + frag_grp) ^ swz) * 16;
      uint4 xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) xa[i] = *reinterpret_cast<const uint4*>(a_base + i * 16 * 128 + chunk);
#pragma unroll
      for (int j = 0; j < FN; ++j) wb[j] = *reinterpret_cast<const uint4*>(b_base + j * 16 * 128 + chunk);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if constexpr (EPI == EPI_APPLY) Mma<T>::run(wb[j], xa[i], pacc[i][j]);
          else Mma<T>::run(wb[j], xa[i], acc[i][j]);
        }
    }

    if constexpr (EPI == EPI_APPLY) {
      if ((kt % STEPS_PER_BLOCK) == STEPS_PER_BLOCK - 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(gcur[i], pacc[i][j][r], acc[i][j][r]);
            pacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          gcur[i] = gnext[i];
        }
      }
    }

    if (more) commit_stage(nxt);
    __syncthreads();
  }

  // ---------------- epilogues ----------------
  // lane holds, for fragment (i, j): m = .. + (lane & 15), n = .. + (lane >> 4) * 4 + r
  if constexpr (EPI == EPI_LINEAR || EPI == EPI_APPLY) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + (wm * FM + i) * 16 + frag_row;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + (wn * FN + j) * 16 + frag_grp * 4;
        if (n >= p.N) continue;  // N is a multiple of 4 (checked on the host)
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (p.resid) {
          float rv[4];
          load4(reinterpret_cast<const T*>(p.resid) + (long)m * p.ldr + n, rv);
          v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
        }
        if (p.relu) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        if (p.out_f32) store4(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n, v);
        else store4(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, v);
      }
    }
  } else {  // EPI_SCORES: per (row, 128-key tile) max / sum and P~ = exp(s - tilemax)
    static_assert(EPI != EPI_SCORES || BN == 128, "score tiles are 128 keys wide");
    float* red = reinterpret_cast<float*>(smem);  // [WN][BM] scratch, main loop is done
    const float sl2 = p.scale * 1.4426950408889634f;  // logits in log2 units
    float tmax[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + (wn * FN + j) * 16 + frag_grp * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = (n + r < p.N) ? acc[i][j][r] * sl2 : -INFINITY;
          acc[i][j][r] = s;
          mx = fmaxf(mx, s);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      tmax[i] = mx;
      if (frag_grp == 0) red[wn * BM + (wm * FM + i) * 16 + frag_row] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int w = 0; w < WN; ++w) tmax[i] = fmaxf(tmax[i], red[w * BM + (wm * FM + i) * 16 + frag_row]);
    }
    __syncthreads();
    float tsum[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + (wm * FM + i) * 16 + frag_row;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + (wn * FN + j) * 16 + frag_grp * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = exp2f(acc[i][j][r] - tmax[i]);  // masked keys: exp2(-inf) = 0
          // the row sum is taken over the values PV will actually multiply
          if constexpr (ElemTraits<T>::kCode == DT_BF16) e = bf2f(f2bf(e));
          v[r] = e;
          sum += e;
        }
        if (m < p.M && n < p.ldc) store4(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, v);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      tsum[i] = sum;
      if (frag_grp == 0) red[wn * BM + (wm * FM + i) * 16 + frag_row] = sum;
    }
    __syncthreads();
    if (wn == 0 && frag_grp == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = (wm * FM + i) * 16 + frag_row, m = m0 + row;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) sum += red[w * BM + row];
        if (m < p.M) {
          const int t = n0 / 128;
          p.mstat[(long)m * p.ntile + t] = tmax[i];  // log2 units
          p.lstat[(long)m * p.ntile + t] = sum;
        }
      }
    }
  }
}

// Per-row combine of the tile statistics: g[m][t] = 2^(m_t - m*) / L with
// L = sum_t l_t 2^(m_t - m*).  One wave per row.
__global__ void relation_stats_kernel(const float* __restrict__ mstat, const float* __restrict__ lstat,
                                      float* __restrict__ g, int M, int ntile) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float mx = -INFINITY;
  for (int t = lane; t < ntile; t += 64) mx = fmaxf(mx, mstat[(long)row * ntile + t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float L = 0.f;
  for (int t = lane; t < ntile; t += 64) L += lstat[(long)row * ntile + t] * exp2f(mstat[(long)row * ntile + t] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o);
  const float inv = 1.f / L;
  for (int t = lane; t < ntile; t += 64) g[(long)row * ntile + t] = exp2f(mstat[(long)row * ntile + t] - mx) * inv;
}

// ---------------- host-side dispatch ----------------
template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS>
static hipError_t launch_tile(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
  static bool attr_set = false;
  auto kern = tile_kernel<T, WM, WN, FM, FN, EPI, GLDS>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError();
}

template <typename T, int EPI>
static hipError_t dispatch_shape(const GemmParams& p, hipStream_t stream) {
  const bool glds = p.staging == 1;
  if (EPI == EPI_LINEAR && p.N <= 64) {
    return glds ? launch_tile<T, 4, 1, 2, 4, EPI_LINEAR, true>(p, stream)
                : launch_tile<T, 4, 1, 2, 4, EPI_LINEAR, false>(p, stream);
  }
  return glds ? launch_tile<T, 2, 2, 4, 4, EPI, true>(p, stream)
              : launch_tile<T, 2, 2, 4, 4, EPI, false>(p, stream);
}

hipError_t run_tile_op(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.dtype == DT_BF16) {
    if (epi == EPI_LINEAR) return dispatch_shape<bf16_t, EPI_LINEAR>(p, stream);
    if (epi == EPI_SCORES) return dispatch_shape<bf16_t, EPI_SCORES>(p, stream);
    return dispatch_shape<bf16_t, EPI_APPLY>(p, stream);
  }
  if (epi == EPI_LINEAR) return dispatch_shape<float, EPI_LINEAR>(p, stream);
  if (epi == EPI_SCORES) return dispatch_shape<float, EPI_SCORES>(p, stream);
  return dispatch_shape<float, EPI_APPLY>(p, stream);
}

hipError_t run_relation_stats(const float* mstat, const float* lstat, float* g, int M, int ntile, hipStream_t stream) {
  const int rows_per_block = 4;
  hipLaunchKernelGGL(relation_stats_kernel, dim3((M + rows_per_block - 1) / rows_per_block), dim3(rows_per_block * 64), 0,
                     stream, mstat, lstat, g, M, ntile);
  return hipGetLastError();
}

}  // namespace hvr
