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

constexpr bool kAsmLdsReads = true;  // see tile_kernel::do_step

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ uint4 lds_read128(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
  // one 16-byte chunk per lane = 8 consecutive k
  template <bool ZERO>
  static __device__ __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
    // ZERO: start a fresh accumulation (C operand is the inline constant 0, no register clear)
    const f32x4 c = ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
  }
  static constexpr int kChunkSteps = 2;  // chunk reads per 128-byte K-step (2 x 4 lane groups)
};

template <> struct Mma<float> {
  // one 16-byte chunk per lane = 4 consecutive k; the 4 lane groups cover 16 k per
  // read, element i of every lane forms MFMA i (any k permutation is fine as long as
  // both operands use the same one).
  template <bool ZERO>
  static __device__ __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
    const f32x4 c = ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), c, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  }
  static constexpr int kChunkSteps = 2;
};

// RESPRE: the residual tile is fetched into registers in one burst at the top of the epilogue (in the store-phase
// mapping) instead of one dependent load per store-phase iteration.
template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS, bool RESPRE = false>
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
  int pid_m = tile / tiles_n, pid_n = tile % tiles_n;  // n fastest: the A panel is reused across the N tiles
  if (p.group_m > 1) {
    // grouped order for problems whose B operand is as large as A (relation scores / apply): `group_m` row tiles
    // share each B panel while their A panels stay L2-resident, instead of streaming all of B once per row tile
    const int per_group = p.group_m * tiles_n, gid = tile / per_group, first = gid * p.group_m;
    const int gsz = tiles_m - first < p.group_m ? tiles_m - first : p.group_m;
    const int in_group = tile - gid * per_group;
    pid_m = first + in_group % gsz;
    pid_n = in_group / gsz;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;

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

  // `first` / `last`: position of this K-step inside a 128-key statistics block (EPI_APPLY only;
  // compile-time constants after unrolling, so the zero-C MFMA and the block combine fold away elsewhere)
  auto do_step = [&](int kt, bool first, bool last) {
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nk;
    if (more) issue_loads(kt + 1, nxt);
    if constexpr (EPI == EPI_APPLY) {
      if (first && kt + STEPS_PER_BLOCK < nk) {
#pragma unroll
        for (int i = 0; i < FM; ++i) gnext[i] = p.g[grow[i] + kt / STEPS_PER_BLOCK + 1];
      }
    }
    if constexpr (GLDS && kAsmLdsReads && (!(WM == 3 && FN == 4) || EPI == EPI_APPLY)) {
      // Fragment reads through inline asm: the compiler does not see them as LDS accesses, so it does not park an
      // s_waitcnt vmcnt(0) in front of them for the LDS-DMA just issued into the OTHER stage -- the next K-step's
      // loads stay in flight under this step's MFMAs (it tracks pending LDS-DMA per LDS object and there is one).
      // LDS returns data in order, so `lgkmcnt(FM + FN)` after all 2 x (FM + FN) reads means "kk = 0 has landed".
      // (The 6-wave 144x128 shape is register-bound and has only 24 MFMAs per wave-step to cover the rigid
      // read / wait structure: measured slower on the short-K convs, so it keeps compiler-scheduled reads -- except
      // in the relation apply pass, whose K = 4608 loop runs one workgroup per CU and needs the in-wave overlap.)
      const uint32_t a_addr = lds_addr(cur) + (wm * FM * 16 + frag_row) * 128;
      const uint32_t b_addr = lds_addr(cur) + BM * 128 + (wn * FN * 16 + frag_row) * 128;
      uint4 xa[2][FM], wb[2][FN];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int chunk = ((kk * 4 + frag_grp) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < FM; ++i) xa[kk][i] = lds_read128(a_addr + i * 16 * 128 + chunk);
#pragma unroll
        for (int j = 0; j < FN; ++j) wb[kk][j] = lds_read128(b_addr + j * 16 * 128 + chunk);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        __builtin_amdgcn_sched_barrier(0);  // pin: MFMAs of kk = 0 stay above the second wait
        if (kk == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // and no MFMA is hoisted above the wait it depends on
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            if constexpr (EPI == EPI_APPLY) {
              if (first && kk == 0) Mma<T>::template run<true>(wb[kk][j], xa[kk][i], pacc[i][j]);
              else Mma<T>::template run<false>(wb[kk][j], xa[kk][i], pacc[i][j]);
            } else {
              Mma<T>::template run<false>(wb[kk][j], xa[kk][i], acc[i][j]);
            }
          }
      }
    } else {
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
          if constexpr (EPI == EPI_APPLY) {
            if (first && kk == 0) Mma<T>::template run<true>(wb[j], xa[i], pacc[i][j]);
            else Mma<T>::template run<false>(wb[j], xa[i], pacc[i][j]);
          } else {
            Mma<T>::template run<false>(wb[j], xa[i], acc[i][j]);
          }
        }
    }
    }
    if constexpr (EPI == EPI_APPLY) {
      if (last) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(gcur[i], pacc[i][j][r], acc[i][j][r]);
          gcur[i] = gnext[i];
        }
      }
    }
    if (more) commit_stage(nxt);
    __syncthreads();
  };
  if constexpr (EPI == EPI_APPLY) {
    for (int kb = 0; kb < nk; kb += STEPS_PER_BLOCK) {
#pragma unroll
      for (int st = 0; st < STEPS_PER_BLOCK; ++st) do_step(kb + st, st == 0, st == STEPS_PER_BLOCK - 1);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) do_step(kt, false, false);
  }

  // ---------------- residual fetch (RESPRE): all of the tile's residual loads are issued here, at the top of the
  // epilogue (no LDS DMA is in flight any more, so the barriers below do not drain them): one overlapped round
  // trip instead of one per store-phase iteration ----------------
  constexpr int E_ROWS = FM * 16, E_CH = BN / 8, E_ITERS = (E_ROWS * E_CH) / NT;
  static_assert(!RESPRE || ((E_ROWS * E_CH) % NT == 0 && sizeof(T) == 2), "RESPRE needs an even store-phase split");
  uint4 rres[RESPRE ? WM : 1][RESPRE ? E_ITERS : 1];
  if constexpr (RESPRE) {
#pragma unroll
    for (int pass = 0; pass < WM; ++pass)
#pragma unroll
      for (int it = 0; it < E_ITERS; ++it) {
        const int c = it * NT + tid, r = c / E_CH, cc = c - r * E_CH;
        const int m = m0 + pass * E_ROWS + r, n = n0 + cc * 8;
        rres[pass][it] = make_uint4(0u, 0u, 0u, 0u);
        if (m < p.M && n < p.N) rres[pass][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.resid) + (long)m * p.ldr + n);
      }
  }

  // ---------------- epilogues ----------------
  // lane holds, for fragment (i, j): m = .. + (lane & 15), n = .. + (lane >> 4) * 4 + r
  if constexpr (EPI == EPI_LINEAR || EPI == EPI_APPLY) {
    // Stage the f32 tile through LDS (one wave-row block of FM*16 rows per pass) so that every global
    // access of the epilogue is a full 16-byte-per-lane row segment: residual loads and output stores
    // are whole lines instead of the 8-byte pieces of the MFMA fragment layout.
    constexpr int ROWS = FM * 16, LDW = BN + 4, CH = BN / 8;
    float* ebuf = reinterpret_cast<float*>(smem);
    const int out_es = p.out_f32 ? 4 : (int)sizeof(T);
    const bool wide_c = ((p.ldc * out_es) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (p.N & 7) == 0;
    const bool wide_r = p.resid && ((p.ldr * (long)sizeof(T)) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
#pragma unroll
    for (int pass = 0; pass < WM; ++pass) {
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int col = (wn * FN + j) * 16 + frag_grp * 4;
            f32x4 v = acc[i][j];
            if (p.bias && n0 + col < p.N) {
              const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + col);
              v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            *reinterpret_cast<f32x4*>(ebuf + (i * 16 + frag_row) * LDW + col) = v;
          }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < (ROWS * CH + NT - 1) / NT; ++it) {
        const int c = it * NT + tid;
        if (c >= ROWS * CH) continue;
        const int r = c / CH, cc = c - r * CH;
        const int m = m0 + pass * ROWS + r, n = n0 + cc * 8;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        const float4 lo = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ebuf + r * LDW + cc * 8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        const bool full = n + 8 <= p.N;  // N % 4 == 0: a chunk is either 8 or 4 valid columns
        if constexpr (RESPRE) {
          const uint4 t = rres[pass][it];
          v[0] += __uint_as_float(t.x << 16); v[1] += __uint_as_float(t.x & 0xffff0000u);
          v[2] += __uint_as_float(t.y << 16); v[3] += __uint_as_float(t.y & 0xffff0000u);
          v[4] += __uint_as_float(t.z << 16); v[5] += __uint_as_float(t.z & 0xffff0000u);
          v[6] += __uint_as_float(t.w << 16); v[7] += __uint_as_float(t.w & 0xffff0000u);
        } else if (p.resid) {
          const T* rp = reinterpret_cast<const T*>(p.resid) + (long)m * p.ldr + n;
          float rv[8];
          if (full && wide_r) {
            if constexpr (sizeof(T) == 2) {
              const uint4 t = *reinterpret_cast<const uint4*>(rp);
              rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
              rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
              rv[4] = __uint_as_float(t.z << 16); rv[5] = __uint_as_float(t.z & 0xffff0000u);
              rv[6] = __uint_as_float(t.w << 16); rv[7] = __uint_as_float(t.w & 0xffff0000u);
            } else {
              load4(rp, rv);
              load4(rp + 4, rv + 4);
            }
          } else {
            load4(rp, rv);
            if (full) load4(rp + 4, rv + 4);
            else rv[4] = rv[5] = rv[6] = rv[7] = 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.out_f32 || sizeof(T) == 4) {
          float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
          store4(cp, v);
          if (full) store4(cp + 4, v + 4);
        } else {
          bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
          if (full && wide_c) {
            *reinterpret_cast<uint4*>(cp) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
          } else {
            store4(cp, v);
            if (full) store4(cp + 4, v + 4);
          }
        }
      }
      __syncthreads();
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
template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS, bool RESPRE = false>
static hipError_t launch_tile_impl(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
  constexpr size_t stage = 2 * (size_t)(BM + BN) * 128;
  constexpr size_t epi = (size_t)FM * 16 * (BN + 4) * 4;  // LDS-staged epilogue buffer
  constexpr size_t lds = stage > epi ? stage : epi;
  static bool attr_set = false;
  auto kern = tile_kernel<T, WM, WN, FM, FN, EPI, GLDS, RESPRE>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError();
}

template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS>
static hipError_t launch_tile(const GemmParams& p, hipStream_t stream) {
  if constexpr (EPI == EPI_LINEAR && GLDS && sizeof(T) == 2) {
    const bool pre = p.resid && (p.N & 7) == 0 && ((p.ldr * 2) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
    if (pre) return launch_tile_impl<T, WM, WN, FM, FN, EPI, GLDS, true>(p, stream);
  }
  return launch_tile_impl<T, WM, WN, FM, FN, EPI, GLDS, false>(p, stream);
}

// Tile menu (BM x BN, waves).  Only the bf16 + global_load_lds path carries the whole menu; the
// register-staged and f32 paths (tests, parity mode) keep the two base shapes.
//   0: 128x128 (2x2 waves)   2 workgroups / CU      1: 128x64  (4x1)   3 / CU
//   2: 144x256 (3x2)         1 / CU                 3: 144x128 (3x2)   2 / CU
//   4: 256x128 (4x2)         1 / CU
const TileShape kTileShapes[kNumTileShapes] = {
    {128, 128, 2, 1.25f}, {128, 64, 3, 0.80f}, {144, 256, 1, 0.90f}, {144, 128, 2, 1.00f}, {256, 128, 1, 0.80f}};

int choose_tile(const GemmParams& p, int epi) {
  if (p.tile_hint > 0 && p.tile_hint <= kNumTileShapes) {
    const int t = p.tile_hint - 1;
    if (epi == EPI_SCORES && kTileShapes[t].bn != 128) return 0;
    if (epi == EPI_APPLY && t == 2) return 3;
    return t;
  }
  const bool full_menu = p.dtype == DT_BF16 && p.staging == 1;
  if (!full_menu) return (epi == EPI_LINEAR && p.N <= 64) ? 1 : 0;
  // Cost model (fitted to tools/kernel_bench.py sweeps on MI355X, profiles/r01_tile_sweep.txt):
  // the chip runs `slots = 256 CUs x wg_per_cu` workgroups at a time; one full round of a shape costs
  // area x wg_per_cu / eff; a trailing partial round that leaves CUs with fewer co-resident workgroups
  // is cheaper, but not proportionally (a lone workgroup cannot saturate a CU).
  static const double kPartial[4][4] = {{0, 0, 0, 0}, {0, 1.0, 0, 0}, {0, 0.8, 1.0, 0}, {0, 0.5, 0.8, 1.0}};
  const int ksteps = p.K / (p.dtype == DT_BF16 ? 64 : 32);
  int best = 0;
  double best_cost = 1e300;
  for (int t = 0; t < kNumTileShapes; ++t) {
    const TileShape& s = kTileShapes[t];
    if (epi == EPI_SCORES && s.bn != 128) continue;
    if (epi == EPI_APPLY && t == 2) continue;  // two accumulator sets do not fit 144x256
    if (p.N <= 64 && s.bn > 64 && t != 0) continue;
    const long tiles = (long)((p.M + s.bm - 1) / s.bm) * ((p.N + s.bn - 1) / s.bn);
    const long slots = 256L * s.wg_per_cu;
    const long full = tiles / slots, rem = tiles % slots;
    const double rounds = (double)full + (rem ? kPartial[s.wg_per_cu][(rem + 255) / 256] : 0.0);
    // short K loops expose the prologue/epilogue of shapes that run one workgroup per CU
    // 144x256 (inline-asm fragment reads, one workgroup per CU) only pays off on long K loops
    const double eff = (t == 2 && ksteps >= 64) ? 1.33 : s.eff * ((s.wg_per_cu == 1 && ksteps <= 8) ? 0.7 : 1.0);
    const double cost = rounds * s.bm * s.bn * s.wg_per_cu / eff;
    if (cost < best_cost * 0.999) { best_cost = cost; best = t; }
  }
  return best;
}

template <typename T, int EPI, bool GLDS>
static hipError_t dispatch_tile(const GemmParams& p, int tile, hipStream_t stream) {
  if constexpr (GLDS && sizeof(T) == 2) {
    switch (tile) {
      case 1: if constexpr (EPI != EPI_SCORES) return launch_tile<T, 4, 1, 2, 4, EPI, GLDS>(p, stream); break;
      case 2: if constexpr (EPI == EPI_LINEAR) return launch_tile<T, 3, 2, 3, 8, EPI, GLDS>(p, stream); break;
      case 3: return launch_tile<T, 3, 2, 3, 4, EPI, GLDS>(p, stream);
      case 4: return launch_tile<T, 4, 2, 4, 4, EPI, GLDS>(p, stream);
      default: break;
    }
    return launch_tile<T, 2, 2, 4, 4, EPI, GLDS>(p, stream);
  } else {
    if (tile == 1) {
      if constexpr (EPI != EPI_SCORES) return launch_tile<T, 4, 1, 2, 4, EPI, GLDS>(p, stream);
    }
    return launch_tile<T, 2, 2, 4, 4, EPI, GLDS>(p, stream);
  }
}

template <typename T, int EPI>
static hipError_t dispatch_shape(const GemmParams& p, hipStream_t stream) {
  const int tile = choose_tile(p, EPI);
  return p.staging == 1 ? dispatch_tile<T, EPI, true>(p, tile, stream) : dispatch_tile<T, EPI, false>(p, tile, stream);
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
