// The MFMA tile engine's kernel template and its dispatch (see gemm.hip for the overview).  Included by the translation units
// that instantiate it: gemm.hip (bf16, f32) and gemm_f16.hip (half, split half) -- two files so that they compile side by side.
#pragma once
#include <type_traits>
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
template <int OFF> __device__ __forceinline__ uint4 lds_read128_off(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// A global load the compiler does not track: the caller guarantees (by counting vmcnt) that it has landed before the
// value is consumed, and marks that point with `landed()`.
__device__ __forceinline__ float load_f32_untracked(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void landed(float& v) { asm volatile("" : "+v"(v)); }
// 16-byte flavour.  The destination must stay where it is until `landed()`: a kernel that uses untracked loads must
// not spill (a spill would store the register before the load has written it) -- hvrnet_amd/csrc/check_regs.py
// verifies that on every build from the compiler's resource remarks.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_u128_untracked(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void landed(u32x4& v) { asm volatile("" : "+v"(v)); }
// 16 bytes per lane, global -> LDS, through buffer addressing: base (uniform) + voff (per lane) + soff (uniform); offsets from
// 2^31 up are outside the resource and come back as zeros.  (The resource type exists in the device pass only: the host
// pass, which just needs the kernel's stub, sees an empty body.)
__device__ __forceinline__ void buffer_load_lds16(const void* base, char* lds, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0x80000000u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, soff, 0, 0);
#else
  (void)base; (void)lds; (void)voff; (void)soff;
#endif
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// loads one K-step costs the wave that issues the fewest (the last one: slots are dealt to waves in order)
constexpr int min_wave_loads(int rows, int nt) {
  int n = 0;
  for (int i = 0; i * nt < rows * 8; ++i) n += (i * nt + (nt - 64) < rows * 8) ? 1 : 0;
  return n;
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

template <> struct Mma<f16_t> {
  template <bool ZERO>
  static __device__ __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
    const f32x4 c = ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  }
  static constexpr int kChunkSteps = 2;
};
// split-half operands: the fragments are half fragments of one plane; which planes meet in an MFMA is the K loop's business
template <> struct Mma<f16s_t> : Mma<f16_t> {};

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
//
// NS: LDS stages.  NS = 2 is the classic double buffer (the next K-step's loads are issued at the top of a step and
// drained at its bottom).  NS = 3 / 4 keep NS - 1 K-steps of LDS-DMA in flight: the waits are counted by hand
// (`s_waitcnt vmcnt(n)` with n = the loads of the steps that may stay outstanding), which needs the inline-asm
// fragment reads -- the compiler would otherwise drain every DMA in front of the first LDS read it can see.
// out[C][ldt] = in[R][ldx]^T in 64 x 64 tiles of 16-bit elements, zero for columns R .. ldt - 1 (misc.hip's transpose_pad_bf16x8_kernel as a
// device function: 16-byte global accesses on both sides); workgroup `b` of `nb` takes tiles b, b + nb, ...  C, ldx, ldt multiples of 8.
template <int NT>
__device__ __forceinline__ void transpose_pad_tiles16(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int R, int C,
                                                      long ldx, long ldt, int b, int nb, char* tile) {
  constexpr int PITCH = 64 * 2 + 16;
  const int tiles_c = (C + 63) / 64, tiles_r = (int)((ldt + 63) / 64), tid = threadIdx.x;
  for (int t = b; t < tiles_c * tiles_r; t += nb) {
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    for (int s = tid; s < 512; s += NT) {
      const int i = s >> 3, q = s & 7, r = r0 + i, c = c0 + q * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r < R && c < C) v = *reinterpret_cast<const uint4*>(in + (long)r * ldx + c);
      *reinterpret_cast<uint4*>(tile + i * PITCH + q * 16) = v;
    }
    __syncthreads();
    for (int s = tid; s < 512; s += NT) {
      const int q = s >> 6, i = s & 63, c = c0 + i, r = r0 + q * 8;
      if (c < C && r < ldt) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = *reinterpret_cast<const unsigned short*>(tile + (q * 8 + 2 * e) * PITCH + i * 2);
          const uint32_t hi = *reinterpret_cast<const unsigned short*>(tile + (q * 8 + 2 * e + 1) * PITCH + i * 2);
          w[e] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4*>(out + (long)c * ldt + r) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    __syncthreads();
  }
}

template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS, bool RESPRE = false, int NS = 2>
__global__ __launch_bounds__(WM* WN * 64) void tile_kernel(const GemmParams p) {
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16, NT = WM * WN * 64;
  // SPLIT (f16s_t, common.h): 4 bytes per logical element in memory, [32 hi | 32 lo] half groups, so the 128-byte line the loader
  // stages per row and K-step holds BOTH planes of 32 logical elements: chunks 0-3 = hi, chunks 4-7 = lo.  What is the second
  // 32-wide half of a K-step for the other formats is here the lo plane of the same 32 elements, and a K-step issues three MFMAs
  // per fragment pair -- B_hi x A_hi when the hi fragments have landed, B_lo x A_hi and B_hi x A_lo when the lo ones have -- from
  // one LDS image: two thirds of the LDS-DMA and fragment-read traffic per MFMA of the half formats.
  constexpr bool SPLIT = std::is_same<T, f16s_t>::value;
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr bool TWO = EPI == EPI_LINEAR2;                 // linear epilogue, two-level accumulation (gemm_params.h)
  constexpr bool LIN = EPI == EPI_LINEAR || TWO;           // ... everything the linear epilogue and its loaders do
  constexpr bool PACC = EPI == EPI_APPLY || TWO;           // a block accumulator beside the running total
  static_assert(!TWO || NS == 2, "the two-level form lives on the double-buffered loop");
  constexpr int EB = (int)sizeof(T);             // bytes per logical element in global memory
  constexpr int BKE = (F32 || SPLIT) ? 32 : 64;  // logical elements per K-step (one 128-byte line per row)
  constexpr int KSG = 128;                       // global bytes per K-step of a row
  constexpr int A_SLOTS = (BM * 8 + NT - 1) / NT, B_SLOTS = (BN * 8 + NT - 1) / NT;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr bool ASM_READS = GLDS && kAsmLdsReads && (NS > 2 || !(WM == 3 && FN == 4) || EPI == EPI_APPLY);
  static_assert(NS == 2 || (ASM_READS && NS <= 4), "deep pipelines need the hand-counted waits");
  constexpr int MINL = min_wave_loads(BM, NT) + min_wave_loads(BN, NT);
  constexpr bool kConvOk = LIN;  // the relation passes never gather: drop the conv state there

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const long grp = (long)blockIdx.z;   // GemmParams::batch: the problem of this workgroup (0 in a launch of one problem: every gs_* term vanishes)
  if constexpr (EPI == EPI_SCORES && sizeof(T) == 2) {
    // the workgroups behind the score tiles transpose V for the apply pass (GemmParams::tr_*: a few-row score grid leaves most CUs idle)
    if (p.tr_blocks > 0 && (int)blockIdx.x >= tiles_m * tiles_n) {
      transpose_pad_tiles16<NT>((const unsigned short*)((const char*)p.tr_in + grp * p.gs_tr_in), (unsigned short*)((char*)p.tr_out + grp * p.gs_tr_out),
                                p.tr_R, p.tr_C, p.tr_ldx, p.tr_ldt, (int)blockIdx.x - tiles_m * tiles_n, p.tr_blocks, smem);
      return;
    }
  }
  float* const mstat_g = p.mstat + grp * p.gs_stat;
  float* const lstat_g = p.lstat + grp * p.gs_stat;
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

  // split-K (EPI_LINEAR plain GEMMs; EPI_APPLY with few query rows, whose block weights g are global per row so that the
  // slices' partials simply add): gridDim.y slices of `ksplit_steps` K-steps each write their own f32 partial
  // [M][N] at C + blockIdx.y * csplit_bytes; a weight-gradient GEMM has a few output tiles and a K of 10^4..10^5, which
  // one workgroup per tile would walk alone while most of the chip idles.  The slice is folded into the base pointers.
  const char* Ab = (const char*)p.A + grp * p.gs_a;
  const char* Bb = (const char*)p.B + grp * p.gs_b;
  char* Cb = (char*)p.C + grp * p.gs_c;
  int nk_slice = p.K / BKE;
  int blk0 = 0;  // EPI_APPLY: first 128-key block of this workgroup's K slice (slices are whole blocks)
  int kt_base = 0;  // conv + split-K: first K-step of this workgroup's slice (the tap / channel offset is derived from it)
  if constexpr (LIN || EPI == EPI_APPLY) {
    if (p.ksplit_steps > 0) {
      const int kt0 = blockIdx.y * p.ksplit_steps;
      if (kConvOk && p.conv) kt_base = kt0;  // an implicit-GEMM slice starts at filter tap (kt0 * BKE) / Cin: the gather takes the offset
      else Ab += (long)kt0 * KSG;
      Bb += (long)kt0 * KSG;
      Cb += (long)blockIdx.y * p.csplit_bytes;
      nk_slice = nk_slice - kt0 < p.ksplit_steps ? nk_slice - kt0 : p.ksplit_steps;
      blk0 = kt0 / (128 / BKE);
    }
  }

  // ---------------- loader setup: one 16-byte chunk per (thread, slot) ----------------
  // 32-bit byte offsets from p.A / p.B (the C ABI rejects operands of 2 GiB and more) and the conv origin of the
  // row packed as (iy << 16) | (ix & 0xffff): half the registers of pointers + two ints, which the pipelined
  // shapes need for their fragments
  int a_off[A_SLOTS], a_yx[A_SLOTS];
  // (B rows are weights / keys: register-staged loads recompute their offsets at issue time, two VALU ops per piece)
  auto b_off = [&](int i) {
    const int s = i * NT + tid, row = s >> 3, c = (s & 7) ^ (row & 7);
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;
    return (int)((long)n * p.ldb * EB + c * 16);
  };
  // Direct-to-LDS loads go through BUFFER addressing (`buffer_load_dwordx4 ... offen lds`): address = resource base (SGPRs) +
  // the slot's byte offset (one VGPR, fixed for the whole kernel) + the K-step's offset (one SGPR: K-step x 128 bytes, or the
  // filter tap's pixel offset for a conv -- it is the same for every row).  A piece then costs no vector address arithmetic at
  // all (a flat `global_load_lds` needs the 64-bit sum per lane per piece: 2.4 VALU + 2 SALU per MFMA in the round-2 counters,
  // r02_window_pmc_sq.txt), and a conv's out-of-image tap is one select -- offset 2^31, past the resource's range, which the
  // hardware answers with zeros -- instead of a compare pair, a 64-bit select and a load from a zero page.  The conv origin can
  // lie `pad` rows / pixels in front of the tensor: the resource base is moved back by that much and every offset forward.
  constexpr unsigned kOob = 0x80000000u;
  int a_bias = 0;
  if (kConvOk && p.conv) a_bias = (int)(((long)p.pad * p.W + p.pad) * p.Cin * (long)EB);
  const char* const rs_a = Ab - a_bias;
  const char* const rs_b = Bb;
  int b_offr[GLDS ? B_SLOTS : 1];
  if constexpr (GLDS) {
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) b_offr[i] = b_off(i);
  }
#pragma unroll
  for (int i = 0; i < A_SLOTS; ++i) {
    const int s = i * NT + tid, row = s >> 3, c = (s & 7) ^ (row & 7);
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;
    if (kConvOk && p.conv) {
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
      const int iy = oy * p.stride - p.pad, ix = ox * p.stride - p.pad;
      a_yx[i] = (iy << 16) | (ix & 0xffff);
      a_off[i] = (int)((((long)b * p.H + iy) * p.W + ix) * (long)p.Cin * EB + c * 16) + (GLDS ? a_bias : 0);
    } else {
      a_yx[i] = 0;
      a_off[i] = (int)((long)m * p.lda * EB + c * 16);
    }
  }
  // Second K segment (p.s2 > 0, plain products only: a Bottleneck's projection shortcut folded into its closing 1x1 --
  // K-steps from K1 on read the block INPUT, an NHWC map [.][H2][W2][K - K1] sampled at stride s2, instead of A).
  const bool seg2 = kConvOk && GLDS && p.s2 > 0;
  const int k1_steps = seg2 ? p.K1 / BKE : 0x7fffffff;
  int a_off2[GLDS ? A_SLOTS : 1];
  if (seg2) {
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      const int s = i * NT + tid, row = s >> 3, c = (s & 7) ^ (row & 7);
      int m = m0 + row;
      m = m < p.M ? m : p.M - 1;
      const int ox = m % p.OW, t = m / p.OW, oy = t % p.OH, b = t / p.OH;
      a_off2[i] = (int)((((long)b * p.H2 + oy * p.s2) * p.W2 + ox * p.s2) * (long)(p.K - p.K1) * EB + c * 16);
    }
  }
  const char* const rs_a2 = (const char*)p.A2;
  uint4 a_reg[A_SLOTS], b_reg[B_SLOTS];  // register staging (unused when GLDS)

  auto issue_loads = [&](int kt, char* stage) {
    long a_koff;
    int dy = 0, dx = 0;
    if (kConvOk && p.conv) {
      const int k = (kt + kt_base) * BKE, tap = k / p.Cin, cin0 = k - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      dy = ky * p.dil;
      dx = kx * p.dil;
      a_koff = (((long)dy * p.W + dx) * p.Cin + cin0) * (long)EB;
    } else {
      a_koff = (long)kt * KSG;
    }
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
      if (A_SLOTS * NT == BM * 8 || i * NT + (tid & ~63) < BM * 8) {
        if constexpr (GLDS) {
          unsigned voff = (unsigned)a_off[i];
          if (kConvOk && p.conv) {
            const bool ok = (unsigned)((a_yx[i] >> 16) + dy) < (unsigned)p.H && (unsigned)((short)a_yx[i] + dx) < (unsigned)p.W;
            voff = ok ? voff : kOob;
          }
          if (kt >= k1_steps) buffer_load_lds16(rs_a2, stage + (i * NT + wave * 64) * 16, (unsigned)a_off2[i], __builtin_amdgcn_readfirstlane((kt - k1_steps) * KSG));
          else buffer_load_lds16(rs_a, stage + (i * NT + wave * 64) * 16, voff, __builtin_amdgcn_readfirstlane((int)a_koff));
        } else {
          const char* src = Ab + ((long)a_off[i] + a_koff);
          if (kConvOk && p.conv) {
            const bool ok = (unsigned)((a_yx[i] >> 16) + dy) < (unsigned)p.H && (unsigned)((short)a_yx[i] + dx) < (unsigned)p.W;
            src = ok ? src : (const char*)p.zero;
          }
          a_reg[i] = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
      if (B_SLOTS * NT == BN * 8 || i * NT + (tid & ~63) < BN * 8) {
        if constexpr (GLDS) {
          buffer_load_lds16(rs_b, stage + BM * 128 + (i * NT + wave * 64) * 16, (unsigned)b_offr[i], __builtin_amdgcn_readfirstlane(kt * KSG));
        } else {
          const char* src = Bb + ((long)b_off(i) + (long)kt * KSG);
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
  // current 128-key block's un-scaled partial product.  The next block's weights g are fetched at the very top of
  // a block, ahead of that block's LDS-DMA, by loads the compiler does not track (a tracked load makes it drain the
  // whole DMA queue in front of the first use): the hand-counted vmcnt wait of the block's last K-step covers them.
  constexpr int GN = PACC ? FM : 1;
  f32x4 pacc[GN][PACC ? FN : 1];
  float gcur[GN], gnext[GN];
  float gref[GN];  // per row: m* + log2(L), so that the block weight is g = 2^(m_t - gref)
  constexpr int STEPS_PER_BLOCK = TWO ? kTwoLevelSteps : 128 / BKE;  // K-steps per 128-key statistics block (EPI_LINEAR2: per block of the two-level sum)
  const int nk = nk_slice;
  const int nblk = nk / STEPS_PER_BLOCK;
  // statistics row of fragment row i of this lane (rows past M read row M - 1: their outputs are never stored)
  auto stat_row = [&](int i) {
    const int m = m0 + (wm * FM + i) * 16 + (lane & 15);
    return (m < p.M ? m : p.M - 1) * p.ntile;
  };
  // Per-row combine of the score pass's tile statistics, g[m][t] = 2^(m_t - m*) / L with L = sum_t l_t 2^(m_t - m*)
  // (v_exp_f32 / v_log_f32 directly: one instruction each, 1 ulp; exp2f / log2f wrap them in denormal handling these
  // weights never need): the four lanes that share a row split the key tiles, then merge their (max, sum) pairs.  It is
  // serial work in front of the tile's first MFMA, so (a) it runs AFTER the pipeline's first K-steps have been requested
  // and (b) its loads go out in batches of 8 tiles x FM rows -- one memory round trip per batch instead of one per tile.
  auto apply_prologue = [&]() {
    if constexpr (EPI == EPI_APPLY) {
      constexpr int TB = 8;
      float mx[FM], l[FM];
      int rowo[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) { mx[i] = -INFINITY; l[i] = 0.f; rowo[i] = stat_row(i); }
      // the first block's maximum of every row, requested with the first batch below (fetched at its use -- after the
      // combine -- the FM loads were FM serial round trips: each sat under its own s_waitcnt vmcnt(0))
      float mfirst[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) mfirst[i] = mstat_g[rowo[i] + blk0];
      for (int t0 = lane >> 4; t0 < p.ntile; t0 += 4 * TB) {
        float mt[FM][TB], lt[FM][TB];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int u = 0; u < TB; ++u) {
            const int t = t0 + 4 * u;
            const bool ok = t < p.ntile;
            const int tt = ok ? t : p.ntile - 1;
            const float a = mstat_g[rowo[i] + tt], b = lstat_g[rowo[i] + tt];
            mt[i][u] = ok ? a : -INFINITY;   // a tile past the end: weight 0 (mx stays finite: t0 itself is a real tile)
            lt[i][u] = ok ? b : 0.f;
          }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int u = 0; u < TB; ++u) {
            const float mn = fmaxf(mx[i], mt[i][u]);
            l[i] = l[i] * __builtin_amdgcn_exp2f(mx[i] - mn) + lt[i][u] * __builtin_amdgcn_exp2f(mt[i][u] - mn);  // 0 * exp2(-inf) = 0
            mx[i] = mn;
          }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        float m_ = mx[i], l_ = l[i];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float mo = __shfl_xor(m_, o), lo = __shfl_xor(l_, o);
          const float mn = fmaxf(m_, mo);  // a lane without tiles carries (-inf, 0); some lane has a finite max
          l_ = (m_ == mn ? l_ : l_ * __builtin_amdgcn_exp2f(m_ - mn)) + (mo == mn ? lo : lo * __builtin_amdgcn_exp2f(mo - mn));
          m_ = mn;
        }
        gref[i] = m_ + __builtin_amdgcn_logf(l_);
        gcur[i] = __builtin_amdgcn_exp2f(mfirst[i] - gref[i]);
        landed(gcur[i]);  // waited for here, ahead of the pipeline, not inside the K loop
        gnext[i] = 0.f;
      }
    }
  };

  const int frag_row = lane & 15, frag_grp = lane >> 4, swz = lane & 7;
  // per-lane part of the fragment read addresses (kk = 1 is the same address with bit 6 flipped)
  const uint32_t a_lane = lds_addr(smem) + (wm * FM * 16 + frag_row) * 128 + ((frag_grp ^ swz) * 16);
  const uint32_t b_lane = lds_addr(smem) + BM * 128 + (wn * FN * 16 + frag_row) * 128 + ((frag_grp ^ swz) * 16);

  // Residual tile (RESPRE): one 16-byte piece per (thread, store-phase slot).  The pipelined loop fetches it right
  // after its prologue DMA with loads the compiler does not track, so that it travels under the whole K loop; the
  // hand-counted waits of the first NS - 2 K-steps let those loads stay in flight (they are younger than the prologue's
  // DMA groups), every later wait covers them.  Addresses are clamped, not predicated: every wave must issue the
  // same number of loads for the counts to hold.
  constexpr int E_ROWS = FM * 16, E_CH = BN / 8, E_ITERS = (E_ROWS * E_CH + NT - 1) / NT;
  static_assert(!RESPRE || sizeof(T) == 2, "RESPRE is the bf16 residual path");
  constexpr bool RES_EARLY = RESPRE && NS > 2 && BN != 256;  // (the 144x256 shapes have no registers left)
  constexpr int RES_LOADS = RES_EARLY ? WM * E_ITERS : 0;
  u32x4 rres[RESPRE ? WM : 1][RESPRE ? E_ITERS : 1];

  // ---------------- main loop ----------------
  if constexpr (NS == 2) {
    // Double buffer, one barrier per K-step: the next step's loads are issued at the top of a step and drained at
    // its bottom (the compiler waits for them in front of the barrier).
    issue_loads(0, smem);
    apply_prologue();
    commit_stage(smem);
    __syncthreads();

    // `first` / `last`: position of this K-step inside a 128-key statistics block (EPI_APPLY only;
    // compile-time constants after unrolling, so the zero-C MFMA and the block combine fold away elsewhere)
    auto do_step = [&](int kt, bool first, bool last) {
      char* cur = smem + (kt & 1) * STAGE_BYTES;
      char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
      const bool more = kt + 1 < nk;
      if constexpr (EPI == EPI_APPLY) {
        if (first) {
          // (unconditional -- the last block re-reads its own weight: a load under `if (a next block exists)` turns gnext into a
          // value merged with its previous self, and the copies that merge costs are the compiler's to place, see pipe_step)
          const int nb = kt / STEPS_PER_BLOCK + 1 < nblk ? kt / STEPS_PER_BLOCK + 1 : nblk - 1;
#pragma unroll
          for (int i = 0; i < FM; ++i) gnext[i] = load_f32_untracked(mstat_g + stat_row(i) + blk0 + nb);
          __builtin_amdgcn_sched_barrier(0);  // the g loads stay ahead of this step's DMA in the vmcnt queue
        }
      }
      if (more) issue_loads(kt + 1, nxt);
      if constexpr (ASM_READS) {
        // Fragment reads through inline asm: the compiler does not see them as LDS accesses, so it does not park an
        // s_waitcnt vmcnt(0) in front of them for the LDS-DMA just issued into the OTHER stage -- the next K-step's
        // loads stay in flight under this step's MFMAs (it tracks pending LDS-DMA per LDS object and there is one).
        // LDS returns data in order, so `lgkmcnt(FM + FN)` after all 2 x (FM + FN) reads means "kk = 0 has landed".
        // (The 6-wave 144x128 shape is register-bound and has only 24 MFMAs per wave-step to cover the rigid
        // read / wait structure: measured slower on the short-K convs, so it keeps compiler-scheduled reads there --
        // except in the relation apply pass.)
        const uint32_t soff = (uint32_t)(cur - smem);
        const uint32_t a0 = a_lane + soff, b0 = b_lane + soff, a1 = a0 ^ 64u, b1 = b0 ^ 64u;
        uint4 xa[2][FM], wb[2][FN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint32_t ab = kk ? a1 : a0, bb = kk ? b1 : b0;
          static_for<FM>([&](auto I) { xa[kk][decltype(I)::value] = lds_read128_off<decltype(I)::value * 2048>(ab); });
          static_for<FN>([&](auto J) { wb[kk][decltype(J)::value] = lds_read128_off<decltype(J)::value * 2048>(bb); });
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
              if constexpr (PACC && SPLIT) {
                // the block's un-scaled partial, three terms per K-step (four K-steps per 128-key block)
                if (kk == 0) {
                  if (first) Mma<T>::template run<true>(wb[0][j], xa[0][i], pacc[i][j]);
                  else Mma<T>::template run<false>(wb[0][j], xa[0][i], pacc[i][j]);
                } else {
                  Mma<T>::template run<false>(wb[1][j], xa[0][i], pacc[i][j]);
                  Mma<T>::template run<false>(wb[0][j], xa[1][i], pacc[i][j]);
                }
              } else if constexpr (PACC) {
                if (first && kk == 0) Mma<T>::template run<true>(wb[kk][j], xa[kk][i], pacc[i][j]);
                else Mma<T>::template run<false>(wb[kk][j], xa[kk][i], pacc[i][j]);
              } else if constexpr (SPLIT) {
                // kk = 0: the hi fragments have landed -> B_hi x A_hi; kk = 1: the lo fragments -> the two cross terms
                if (kk == 0) {
                  Mma<T>::template run<false>(wb[0][j], xa[0][i], acc[i][j]);
                } else {
                  Mma<T>::template run<false>(wb[1][j], xa[0][i], acc[i][j]);
                  Mma<T>::template run<false>(wb[0][j], xa[1][i], acc[i][j]);
                }
              } else {
                Mma<T>::template run<false>(wb[kk][j], xa[kk][i], acc[i][j]);
              }
            }
        }
      } else {
        const char* a_base = cur + (wm * FM * 16 + frag_row) * 128;
        const char* b_base = cur + BM * 128 + (wn * FN * 16 + frag_row) * 128;
        if constexpr (SPLIT) {
          uint4 xa[2][FM], wb[2][FN];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int chunk = ((kk * 4 + frag_grp) ^ swz) * 16;
#pragma unroll
            for (int i = 0; i < FM; ++i) xa[kk][i] = *reinterpret_cast<const uint4*>(a_base + i * 16 * 128 + chunk);
#pragma unroll
            for (int j = 0; j < FN; ++j) wb[kk][j] = *reinterpret_cast<const uint4*>(b_base + j * 16 * 128 + chunk);
          }
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              if constexpr (PACC) {
                if (first) Mma<T>::template run<true>(wb[0][j], xa[0][i], pacc[i][j]);
                else Mma<T>::template run<false>(wb[0][j], xa[0][i], pacc[i][j]);
                Mma<T>::template run<false>(wb[1][j], xa[0][i], pacc[i][j]);
                Mma<T>::template run<false>(wb[0][j], xa[1][i], pacc[i][j]);
              } else {
                Mma<T>::template run<false>(wb[0][j], xa[0][i], acc[i][j]);
                Mma<T>::template run<false>(wb[1][j], xa[0][i], acc[i][j]);
                Mma<T>::template run<false>(wb[0][j], xa[1][i], acc[i][j]);
              }
            }
        } else {
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
              if constexpr (PACC) {
                if (first && kk == 0) Mma<T>::template run<true>(wb[j], xa[i], pacc[i][j]);
                else Mma<T>::template run<false>(wb[j], xa[i], pacc[i][j]);
              } else {
                Mma<T>::template run<false>(wb[j], xa[i], acc[i][j]);
              }
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
            landed(gnext[i]);
            gcur[i] = __builtin_amdgcn_exp2f(gnext[i] - gref[i]);
          }
        }
      } else if constexpr (TWO) {
        if (last) {   // the block's sum joins the running total: one f32 addition per element and block
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][j][r] += pacc[i][j][r];
        }
      }
      if (more) commit_stage(nxt);
      __syncthreads();
    };
    if constexpr (PACC) {
      for (int kb = 0; kb < nk; kb += STEPS_PER_BLOCK) {
#pragma unroll
        for (int st = 0; st < STEPS_PER_BLOCK; ++st) do_step(kb + st, st == 0, st == STEPS_PER_BLOCK - 1);
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) do_step(kt, false, false);
    }
  } else {
    // NS-slot LDS ring with a hand-placed instruction stream.  A K-step is two halves (kk = 0 / 1, 32 bf16 of K each);
    // each half issues its FM x FN MFMAs with one "filler" after every MFMA: first the fragment reads of the NEXT
    // half (into the other fragment register set), then this step's share of the LDS-DMA for K-step kt + NS - 1
    // (A slots in half 0, B slots in half 1).  The per-CU load path (64 B/clk) and the LDS therefore run UNDER the
    // MFMAs instead of in a burst in front of them, and a half never waits for data requested less than a half ago.
    //   half 0:  lgkmcnt(0)                      -> frags(kt, 0) landed
    //   half 1:  lgkmcnt(0)                      -> frags(kt, 1) landed: this wave is done reading slot kt % NS
    //            vmcnt(n), s_barrier             -> K-step kt + 1 is in LDS for everyone; slot kt % NS is free
    // The waits are counted by hand (vmcnt counts this wave's loads in issue order; n = the loads that may stay in
    // flight, i.e. those of the K-steps after kt + 1, taken for the wave that issues the fewest).
    static_assert(EPI != EPI_APPLY || STEPS_PER_BLOCK == 2, "the pipelined apply loop assumes 2 K-steps per block");
    constexpr int MIN_A = min_wave_loads(BM, NT);
    constexpr int NR = FM + FN, NM = FM * FN;
    // fragment sets: [0] = kk 0, [1] = kk 1.  SPLIT: [0] = the hi plane, [1] = the lo plane; the B side keeps a third set, [2]: the
    // next step's B_hi fragments are requested while this step's B_hi x A_lo terms still read the current ones, so B_hi lives in
    // [0] on even and in [2] on odd K-steps (an A_hi fragment is re-requested as soon as its row's B_lo x A_hi terms are issued)
    uint4 fa[2][FM], fb[SPLIT ? 3 : 2][FN];
    int kt_load = 0;     // the K-step being loaded (second-segment products switch operand at k1_steps)
    int b_koff = 0;      // its byte offset in a B row

    // BRANCH-FREE state of the K-step being loaded.  A taken branch costs a wave ~100 cycles of instruction fetch (tools/
    // kloop_probe.hip, profiles/r04_kloop_probe.txt: an MFMA loop of 36 MFMAs per iteration runs at 1.58 PF/s, unrolled twice at
    // 1.89), and this loop used to take one or two around every DMA piece -- `if (conv)`, `if (second K segment)`, `if (this wave
    // has a piece in the last slot)` as an exec-mask branch -- while every VALU operation squeezed between two 16-cycle MFMAs costs
    // issue slots the matrix pipe then waits for.  So: ONE form for plain products and convs (the plain product is the conv formula
    // with an unreachable channel count), the filter tap advances by scalar selects (no division, no branch), a piece's out-of-image
    // test is a precomputed per-tap bit (two VALU operations: shift the lane's mask by the tap, OR the bit into bit 31 of the offset --
    // offsets from 2^31 up are outside the buffer resource and read as zeros), the second K segment is scalar selects plus one
    // v_cndmask, and the last-slot test is a scalar compare on the wave index.
    const bool is_conv = kConvOk && p.conv;
    const int cCin = is_conv ? p.Cin : 0x40000000, cKW = is_conv ? p.KW : 1, cDil = is_conv ? p.dil : 0;
    const int cRowB = is_conv ? p.W * p.Cin * EB : 0, cPixB = is_conv ? p.Cin * EB : 0;   // bytes per input row / pixel (tensors < 2 GiB)
    // bit t of oob[i]: filter tap t of slot i's pixel lies outside the image (<= 32 taps: run_tile_op keeps larger filters off this path)
    unsigned oob[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) oob[i] = 0u;
    if (is_conv) {
      // (rows and columns tested separately: KH + KW compares per slot instead of KH x KW -- a tile of a 9-step conv notices)
      unsigned colbad[A_SLOTS];
#pragma unroll
      for (int i = 0; i < A_SLOTS; ++i) colbad[i] = 0u;
      for (int kx = 0, dx = 0; kx < p.KW; ++kx, dx += p.dil) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) colbad[i] |= (unsigned)((short)a_yx[i] + dx) < (unsigned)p.W ? 0u : (1u << kx);
      }
      const unsigned all_kw = p.KW >= 32 ? ~0u : (1u << p.KW) - 1u;   // (a 1 x 32 filter passes the <= 32 taps gate: no shift by 32)
      for (int ky = 0, dy = 0, sh = 0; ky < p.KH; ++ky, dy += p.dil, sh += p.KW) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) oob[i] |= ((unsigned)((a_yx[i] >> 16) + dy) < (unsigned)p.H ? colbad[i] : all_kw) << sh;
      }
    }
    int t_cin0 = 0, t_kx = 0, t_ky = 0, t_tap = 0, a_koff32 = 0;
    auto tap_set = [&](int kt) {   // (division form: once, behind the prologue)
      kt_load = kt;
      b_koff = kt * KSG;
      const int k = (kt + kt_base) * BKE;
      t_tap = k / cCin;
      t_cin0 = k - t_tap * cCin;
      t_ky = t_tap / cKW;
      t_kx = t_tap - t_ky * cKW;
      a_koff32 = t_ky * cDil * cRowB + t_kx * cDil * cPixB + t_cin0 * EB;
    };
    auto tap_next = [&]() {        // the next K-step, by scalar selects
      ++kt_load;
      b_koff += KSG;
      t_cin0 += BKE;
      const bool w1 = t_cin0 >= cCin;
      t_cin0 = w1 ? 0 : t_cin0;
      t_tap += w1 ? 1 : 0;
      t_kx += w1 ? 1 : 0;
      const bool w2 = t_kx >= cKW;
      t_kx = w2 ? 0 : t_kx;
      t_ky += w2 ? 1 : 0;
      const int wm = -(int)w1;   // both arms computed and masked: a select here comes back as two scalar branches per K-step
      a_koff32 = ((t_ky * cDil * cRowB + t_kx * cDil * cPixB) & wm) | ((a_koff32 + KSG) & ~wm);
    };
    auto dma_a = [&](auto I, char* stage) {
      constexpr int i = decltype(I)::value;
      if (A_SLOTS * NT == BM * 8 || (i + 1) * NT <= BM * 8 || i * NT + wave * 64 < BM * 8) {   // (scalar: `wave` lives in an SGPR)
        const unsigned voff = (unsigned)a_off[i] | ((oob[i] >> (t_tap & 31)) << 31);
#ifndef HVR_DBG_NODMA
        const bool s2 = kt_load >= k1_steps;
        const char* const base = s2 ? rs_a2 : rs_a;
        const unsigned vo = s2 ? (unsigned)a_off2[i] : voff;
        const int so = s2 ? (kt_load - k1_steps) * KSG : a_koff32;
        buffer_load_lds16(base, stage + (i * NT + wave * 64) * 16, vo, __builtin_amdgcn_readfirstlane(so));
#else
        asm volatile("" ::"v"(voff), "v"(stage));
#endif
      }
    };
    auto dma_b = [&](auto I, char* stage) {   // (of the K-step the tap state stands at)
      constexpr int i = decltype(I)::value;
      if (B_SLOTS * NT == BN * 8 || (i + 1) * NT <= BN * 8 || i * NT + wave * 64 < BN * 8) {
#ifndef HVR_DBG_NODMA
        buffer_load_lds16(rs_b, stage + BM * 128 + (i * NT + wave * 64) * 16, (unsigned)b_offr[i], __builtin_amdgcn_readfirstlane(b_koff));
#else
        { const unsigned bo = (unsigned)b_offr[i]; asm volatile("" ::"v"(bo), "v"(stage)); }
#endif
      }
    };
    // fragment read r of half kk from the slot at byte offset soff
    auto read_frag = [&](auto R, auto KK, uint32_t soff) {
      constexpr int r = decltype(R)::value, kk = decltype(KK)::value;
#ifdef HVR_DBG_NOLDSREAD
      (void)soff;
#else
      if constexpr (r < FM) fa[kk == 1 ? 1 : 0][r] = lds_read128_off<r * 2048>((a_lane + soff) ^ (kk == 1 ? 64u : 0u));
      else fb[kk][r - FM] = lds_read128_off<(r - FM) * 2048>((b_lane + soff) ^ (kk == 1 ? 64u : 0u));
#endif
    };

    // prologue: NS - 1 K-steps in flight, K-step 0 landed, its first half's fragments requested
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) issue_loads(s, smem + s * STAGE_BYTES);
    tap_set(NS - 2);   // the last K-step requested: every loading pipe_step advances the state by one
    apply_prologue();
    if constexpr (RES_EARLY) {
#pragma unroll
      for (int pass = 0; pass < WM; ++pass)
#pragma unroll
        for (int it = 0; it < E_ITERS; ++it) {
          int c = it * NT + tid;
          c = c < E_ROWS * E_CH ? c : E_ROWS * E_CH - 1;
          const int r = c / E_CH, cc = c - r * E_CH;
          int m = m0 + pass * E_ROWS + r, n = n0 + cc * 8;
          m = m < p.M ? m : p.M - 1;
          n = n < p.N ? n : p.N - 8;
          rres[pass][it] = load_u128_untracked(reinterpret_cast<const T*>(p.resid) + (long)m * p.ldr + n);
        }
    }
    if (nk >= NS - 1) wait_vmcnt<(NS - 2) * MINL + RES_LOADS>();
    else wait_vmcnt<RES_LOADS>();
    __builtin_amdgcn_s_barrier();
    static_for<NR>([&](auto R) { read_frag(R, std::integral_constant<int, 0>{}, 0u); });

    int cs = 0;  // ring slot of the K-step being computed
    // LOAD: this step issues the DMA of K-step kt + NS - 1: 1 / 0 at compile time (the steady-state loop must not
    // branch between its MFMAs), 2 = decided at run time from kt (loop tails).  NEXT: a K-step kt + 1 exists (its
    // first fragments are requested in half 1).  FIRST / LAST: position inside a 128-key block (EPI_APPLY).
    // PAR (SPLIT): parity of the K-step.  The B_hi fragments alternate between sets [0] and [2] -- even steps multiply out of [0]
    // and request the next step's into [2], odd steps the other way round.  (Not "request into [2], move to [0] at the top of the
    // next step": a value renamed across iterations is resolved by REGISTER COPIES the compiler places where it likes -- it put
    // them at the loop header, in front of the lgkmcnt wait of half 0, reading fragments whose inline-asm ds_read it knows no
    // latency for.  Alone the data had always landed by then; with another launch's workgroups keeping the LDS busy it had not.
    // check_asm_waits.py now scans every build for reads of that kind.)
    auto pipe_step = [&](int kt, auto LOAD, auto NEXT, auto FIRST, auto LAST, auto PAR) {
      constexpr int HS = decltype(PAR)::value ? 2 : 0, HN = decltype(PAR)::value ? 0 : 2;  // SPLIT: this / the next step's B_hi set
      constexpr int lmode = decltype(LOAD)::value;
      constexpr bool next = decltype(NEXT)::value, first = decltype(FIRST)::value, last = decltype(LAST)::value;
      const bool load = lmode == 2 ? kt + NS - 1 < nk : lmode == 1;
      const uint32_t soff = (uint32_t)cs * STAGE_BYTES;
      const int ns = cs + 1 == NS ? 0 : cs + 1;
      char* lstage = smem + (cs == 0 ? NS - 1 : cs - 1) * STAGE_BYTES;  // freed by the previous step's barrier
      if constexpr (EPI == EPI_APPLY && first) {
        // every block requests a weight, the last one its own again (its waits are vmcnt(0)): loading only `if a next block
        // exists` would make gnext a value merged with its previous self across iterations, and the register copies such a merge
        // costs are placed by the compiler, which knows nothing of this load's latency (see PAR above, check_asm_waits.py)
        const int nb = kt / STEPS_PER_BLOCK + 1 < nblk ? kt / STEPS_PER_BLOCK + 1 : nblk - 1;
#pragma unroll
        for (int i = 0; i < FM; ++i) gnext[i] = load_f32_untracked(mstat_g + stat_row(i) + blk0 + nb);
      }
      if (load) tap_next();
      static_for<2>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        constexpr int ND = kk == 0 ? A_SLOTS : B_SLOTS;
        constexpr bool reads = kk == 0 || next;  // half 1 requests the next step's first fragments
        constexpr int NF = (reads ? NR : 0) + ND;
        constexpr int NMK = (SPLIT && kk == 1) ? 2 * NM : NM;   // SPLIT: half 1 issues the two cross terms of every fragment pair
        constexpr int PER = (NF + NMK - 1) / NMK;  // fillers after each MFMA
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (kk == 1) {
          // loads that may stay in flight: the K-steps after kt + 1 (full groups), plus the A half of the group
          // this step is issuing; an apply block's g loads sit exactly that far back, hence one fewer there
          constexpr int G = (EPI == EPI_APPLY && NS == 4) ? 1 : 0;
          // (the residual prefetch is younger than the DMA groups of K-steps 1 .. NS - 2: it may stay in flight
          // while those are awaited)
          if (RES_EARLY && kt <= NS - 3) {
            if (load) wait_vmcnt<(NS - 3) * MINL + MIN_A - G + RES_LOADS>();
            else if (NS == 4 && kt + 2 < nk) wait_vmcnt<MINL - G + RES_LOADS>();
            else wait_vmcnt<RES_LOADS>();
          } else {
            if (load) wait_vmcnt<(NS - 3) * MINL + MIN_A - G>();
            else if (NS == 4 && kt + 2 < nk) wait_vmcnt<MINL - G>();
            else wait_vmcnt<0>();
          }
          __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NMK>([&](auto Q) {
          constexpr int qq = decltype(Q)::value, q = qq % NM, i = q / FN, j = q % FN;
          if constexpr (EPI == EPI_APPLY) {
            if constexpr (first && kk == 0) Mma<T>::template run<true>(fb[kk][j], fa[kk][i], pacc[i][j]);
            else Mma<T>::template run<false>(fb[kk][j], fa[kk][i], pacc[i][j]);
          } else if constexpr (SPLIT) {
            if constexpr (kk == 0) Mma<T>::template run<false>(fb[HS][j], fa[0][i], acc[i][j]);     // B_hi x A_hi
            else if constexpr (qq < NM) Mma<T>::template run<false>(fb[1][j], fa[0][i], acc[i][j]);  // B_lo x A_hi, row by row
            else Mma<T>::template run<false>(fb[HS][j], fa[1][i], acc[i][j]);                       // B_hi x A_lo
          } else {
            Mma<T>::template run<false>(fb[kk][j], fa[kk][i], acc[i][j]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SPLIT && kk == 1) {
            // one filler per slot: the slot behind a row's last B_lo x A_hi term re-requests that row's A_hi fragment (the next
            // K-step's); every other slot takes the next of (the next step's B_hi fragments into the spare set, then the B DMA)
            constexpr bool row_end = qq < NM && (qq + 1) % FN == 0;
            constexpr int ends_before = (qq / FN) < FM ? (qq / FN) : FM;
            constexpr int oi = qq - ends_before;            // index in the "other" list (only meaningful when !row_end)
            constexpr int n_bhi = next ? FN : 0;
            static_assert(2 * NM - FM >= FN + B_SLOTS, "half 1 has a slot for every filler");
            if constexpr (row_end) {
              if constexpr (next) read_frag(std::integral_constant<int, qq / FN>{}, std::integral_constant<int, 0>{}, (uint32_t)ns * STAGE_BYTES);
            } else if constexpr (oi < n_bhi) {
              read_frag(std::integral_constant<int, FM + oi>{}, std::integral_constant<int, HN>{}, (uint32_t)ns * STAGE_BYTES);
            } else if constexpr (oi - n_bhi < B_SLOTS) {
              if (load) dma_b(std::integral_constant<int, oi - n_bhi>{}, lstage);
            }
          } else
          static_for<PER>([&](auto U) {
            constexpr int f = qq * PER + decltype(U)::value;
            constexpr int fr = reads ? f : f + NR;  // filler index in the (reads, DMA) order
            if constexpr (f < NF) {
              if constexpr (fr < NR) {
                if constexpr (kk == 0) read_frag(std::integral_constant<int, fr>{}, std::integral_constant<int, 1>{}, soff);
                else read_frag(std::integral_constant<int, fr>{}, std::integral_constant<int, 0>{}, (uint32_t)ns * STAGE_BYTES);
              } else if (load) {
                if constexpr (kk == 0) dma_a(std::integral_constant<int, fr - NR>{}, lstage);
                else dma_b(std::integral_constant<int, fr - NR>{}, lstage);
              }
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      if constexpr (EPI == EPI_APPLY && last) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(gcur[i], pacc[i][j][r], acc[i][j][r]);
          landed(gnext[i]);
          gcur[i] = __builtin_amdgcn_exp2f(gnext[i] - gref[i]);
        }
      }
      cs = ns;
    };
    constexpr std::true_type Y{};
    constexpr std::false_type N{};
    constexpr std::integral_constant<int, 1> L1{};
    constexpr std::integral_constant<int, 0> L0{};
    constexpr std::integral_constant<int, 2> LR{};
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    if constexpr (EPI == EPI_APPLY) {  // nk is even: 128-key blocks of two K-steps
      int kt = 0;
      for (; kt + NS < nk; kt += 2) {  // both steps of the block load
        pipe_step(kt, L1, Y, Y, N, P0);
        pipe_step(kt + 1, L1, Y, N, Y, P0);
      }
      for (; kt < nk; kt += 2) {  // the last NS / 2 blocks: the second step never loads
        pipe_step(kt, LR, Y, Y, N, P0);
        if (kt + 2 < nk) pipe_step(kt + 1, L0, Y, N, Y, P0);
        else pipe_step(kt + 1, L0, N, N, Y, P0);
      }
    } else if constexpr (SPLIT) {  // steps in (even, odd) pairs: the B_hi sets swap roles from one step to the next
      int kt = 0;
      for (; kt + NS < nk; kt += 2) {
        pipe_step(kt, L1, Y, N, N, P0);
        pipe_step(kt + 1, L1, Y, N, N, P1);
      }
      for (; kt + 2 < nk; kt += 2) {
        pipe_step(kt, LR, Y, N, N, P0);
        pipe_step(kt + 1, LR, Y, N, N, P1);
      }
      if (kt + 2 == nk) {
        pipe_step(kt, L0, Y, N, N, P0);
        pipe_step(kt + 1, L0, N, N, N, P1);
      } else {  // kt + 1 == nk (nk >= 1)
        pipe_step(kt, L0, N, N, N, P0);
      }
    } else {
      int kt = 0;
      for (; kt + NS < nk; kt += 2) {   // two K-steps per iteration: one loop branch per 4 x FM x FN MFMAs
        pipe_step(kt, L1, Y, N, N, P0);
        pipe_step(kt + 1, L1, Y, N, N, P0);
      }
      for (; kt + NS - 1 < nk; ++kt) pipe_step(kt, L1, Y, N, N, P0);
      for (; kt + 1 < nk; ++kt) pipe_step(kt, L0, Y, N, N, P0);
      pipe_step(kt, L0, N, N, N, P0);  // kt == nk - 1: a product has at least one K-step
    }
  }

  // ---------------- residual fetch (RESPRE): all of the tile's residual loads are issued here, at the top of the
  // epilogue (no LDS DMA is in flight any more, so the barriers below do not drain them): one overlapped round
  // trip instead of one per store-phase iteration ----------------
  if constexpr (RES_EARLY) {
    wait_vmcnt<0>();  // K loops shorter than the ring never reach a wait that covers the residual prefetch
#pragma unroll
    for (int pass = 0; pass < WM; ++pass)
#pragma unroll
      for (int it = 0; it < E_ITERS; ++it) landed(rres[pass][it]);
  }
  if constexpr (RESPRE && !RES_EARLY) {
#pragma unroll
    for (int pass = 0; pass < WM; ++pass)
#pragma unroll
      for (int it = 0; it < E_ITERS; ++it) {
        const int c = it * NT + tid, r = c / E_CH, cc = c - r * E_CH;
        const int m = m0 + pass * E_ROWS + r, n = n0 + cc * 8;
        rres[pass][it] = u32x4{0u, 0u, 0u, 0u};
        if (c < E_ROWS * E_CH && m < p.M && n < p.N) rres[pass][it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.resid) + (long)m * p.ldr + n);
      }
  }

  // ---------------- epilogues ----------------
  // lane holds, for fragment (i, j): m = .. + (lane & 15), n = .. + (lane >> 4) * 4 + r
  if constexpr (LIN || EPI == EPI_APPLY) {
    // Stage the f32 tile through LDS (one wave-row block of FM*16 rows per pass) so that every global
    // access of the epilogue is a full 16-byte-per-lane row segment: residual loads and output stores
    // are whole lines instead of the 8-byte pieces of the MFMA fragment layout.
    constexpr int ROWS = FM * 16, LDW = BN + 4, CH = BN / 8;
    float* ebuf = reinterpret_cast<float*>(smem);
    const int out_es = p.out_f32 ? 4 : EB;
    const bool wide_c = ((p.ldc * out_es) & 15) == 0 && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0 && (p.N & 7) == 0;
    const bool wide_r = p.resid && ((p.ldr * (long)EB) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
    // The shifts of this wave's FN column fragments, fetched in ONE batch ahead of the staging passes (they depend on j only).  Left
    // inside the (i, j) loop under its `if (p.bias && ...)` the compiler emitted one load + s_waitcnt vmcnt(0) per fragment: FM x FN
    // serialised L2 round trips at the top of every tile's epilogue (18 for the 144 x 256 shape).  Columns past N (never stored)
    // read the last valid float4.
    float4 bvj[FN];
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int c = n0 + (wn * FN + j) * 16 + frag_grp * 4;
        c = c < p.N ? c : p.N - 4;
        bvj[j] = *reinterpret_cast<const float4*>(p.bias + c);
      }
      if constexpr (SPLIT) {
        if (p.beta != 0.f) {
#pragma unroll
          for (int j = 0; j < FN; ++j) { bvj[j].x *= p.beta; bvj[j].y *= p.beta; bvj[j].z *= p.beta; bvj[j].w *= p.beta; }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j) bvj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // split half: a pass's residual rows ([hi | lo] planes, two 16-byte pieces per 8 columns) are requested in one burst at the top
    // of the pass, under the staging of the accumulators -- one overlapped round trip per pass instead of one per store-phase
    // iteration (what RESPRE does for the 2-byte formats)
    uint4 sres[SPLIT ? E_ITERS : 1][2];
#pragma unroll
    for (int pass = 0; pass < WM; ++pass) {
      if constexpr (SPLIT) {
        if (p.resid) {
#pragma unroll
          for (int it = 0; it < E_ITERS; ++it) {
            int c = it * NT + tid;
            c = c < E_ROWS * E_CH ? c : E_ROWS * E_CH - 1;
            const int r = c / E_CH, cc = c - r * E_CH;
            int m = m0 + pass * E_ROWS + r, n = n0 + cc * 8;
            m = m < p.M ? m : p.M - 1;
            n = n < p.N ? n : p.N - 8;
            const char* rp = reinterpret_cast<const char*>(p.resid) + (long)m * p.ldr * 4 + split_col_bytes(n);
            sres[it][0] = *reinterpret_cast<const uint4*>(rp);
            sres[it][1] = *reinterpret_cast<const uint4*>(rp + kSplitPlane);
          }
        }
      }
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int col = (wn * FN + j) * 16 + frag_grp * 4;
            f32x4 v = acc[i][j];
            if constexpr (SPLIT) {
              if (p.alpha != 0.f) v *= p.alpha;
            }
            v[0] += bvj[j].x; v[1] += bvj[j].y; v[2] += bvj[j].z; v[3] += bvj[j].w;
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
          float rv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) unpack2<T>(rres[pass][it][e], rv[2 * e], rv[2 * e + 1]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else if constexpr (SPLIT) {
          if (p.resid) {  // (the C ABI admits split operands in whole 8-column chunks only: `full` holds)
            const uint4 th = sres[it][0], tl = sres[it][1];
            float rv[8];
            merge2(th.x, tl.x, rv[0], rv[1]); merge2(th.y, tl.y, rv[2], rv[3]);
            merge2(th.z, tl.z, rv[4], rv[5]); merge2(th.w, tl.w, rv[6], rv[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
          }
        } else if (p.resid) {
          const T* rp = reinterpret_cast<const T*>(p.resid) + (long)m * p.ldr + n;
          float rv[8];
          if (full && wide_r) {
            if constexpr (sizeof(T) == 2) {
              const uint4 t = *reinterpret_cast<const uint4*>(rp);
              unpack2<T>(t.x, rv[0], rv[1]); unpack2<T>(t.y, rv[2], rv[3]);
              unpack2<T>(t.z, rv[4], rv[5]); unpack2<T>(t.w, rv[6], rv[7]);
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
        if (p.out_f32 || F32) {
          float* cp = reinterpret_cast<float*>(Cb) + (long)m * p.ldc + n;
          store4(cp, v);
          if (full) store4(cp + 4, v + 4);
        } else if constexpr (SPLIT) {
          char* cp = Cb + (long)m * p.ldc * 4 + split_col_bytes(n);
          uint4 h, l;
          split2(v[0], v[1], h.x, l.x); split2(v[2], v[3], h.y, l.y);
          split2(v[4], v[5], h.z, l.z); split2(v[6], v[7], h.w, l.w);
          *reinterpret_cast<uint4*>(cp) = h;
          *reinterpret_cast<uint4*>(cp + kSplitPlane) = l;
        } else if constexpr (sizeof(T) == 2) {
          T* cp = reinterpret_cast<T*>(Cb) + (long)m * p.ldc + n;
          if (full && wide_c) {
            *reinterpret_cast<uint4*>(cp) = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
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
#ifdef HVR_DBG_NOSCORE_EPI
    if (p.scale != 12345.f) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == 12345.f) mstat_g[0] = t;
      return;
    }
#endif
    // LDS scratch (the main loop is done): the P~ tile, staged so that it leaves in whole 16-byte row segments,
    // and [WN][BM] floats for the max / sum exchange between the column waves
    constexpr int PITCH = BN * (int)sizeof(T) + 8;  // bytes per staged row; 66 dwords: the 8-byte writes of a lane
                                                    // group (16 rows) land on 32 distinct banks
    char* pbuf = smem;
    float* red = reinterpret_cast<float*>(smem + BM * PITCH);
    const float sl2 = p.scale * 1.4426950408889634f;  // logits in log2 units
    const bool ragged = n0 + BN > p.N;                // only the last key tile masks columns
    float tmax[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + (wn * FN + j) * 16 + frag_grp * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = acc[i][j][r];  // raw dot products: the (positive) scale is folded into the exponent's FMA below
          if (ragged) s = (n + r < p.N) ? s : -INFINITY;
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
      tmax[i] *= sl2;  // log2 units from here on
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = (wm * FM + i) * 16 + frag_row;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = (wn * FN + j) * 16 + frag_grp * 4;
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], sl2, -tmax[i]));  // masked keys: exp2(-inf) = 0
        if constexpr (SPLIT) {
          // probabilities are small numbers (1 / keys on average): stored x 2^12 so that their lo halves stay normal; the apply
          // product takes the factor back (alpha, capi.hip), the statistics below are of the unscaled values
        }
        if constexpr (sizeof(T) == 2) {
          // (the row sum is taken before rounding: the rounding errors of a tile's 128 values average out far below
          // the bf16 resolution of the output, and re-expanding the packed values costs as much VALU as the exponentials)
          const uint32_t lo = pack2<T>(e[0], e[1]), hi = pack2<T>(e[2], e[3]);
          sum += (e[0] + e[1]) + (e[2] + e[3]);
          *reinterpret_cast<uint2*>(pbuf + row * PITCH + col * 2) = make_uint2(lo, hi);
        } else if constexpr (SPLIT) {
          uint32_t h0, l0, h1, l1;
          split2(e[0] * kSplitProbScale, e[1] * kSplitProbScale, h0, l0);
          split2(e[2] * kSplitProbScale, e[3] * kSplitProbScale, h1, l1);
          sum += (e[0] + e[1]) + (e[2] + e[3]);
          char* dst = pbuf + row * PITCH + split_col_bytes(col);   // the staged row has the memory layout of the 128-key tile
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dst + kSplitPlane) = make_uint2(l0, l1);
        } else {
          sum += (e[0] + e[1]) + (e[2] + e[3]);
          *reinterpret_cast<float2*>(pbuf + row * PITCH + col * 4) = make_float2(e[0], e[1]);
          *reinterpret_cast<float2*>(pbuf + row * PITCH + col * 4 + 8) = make_float2(e[2], e[3]);
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      if (frag_grp == 0) red[wn * BM + row] = sum;
    }
    __syncthreads();
    {  // P~ leaves in 16-byte row segments (ldc is a multiple of 128 keys: every column of the tile exists)
      constexpr int CH = BN * (int)sizeof(T) / 16;  // chunks per row
#pragma unroll
      for (int it = 0; it < (BM * CH + NT - 1) / NT; ++it) {
        const int c = it * NT + tid;
        if ((BM * CH) % NT != 0 && c >= BM * CH) continue;
        const int r = c / CH, cc = c - r * CH, m = m0 + r;
        const uint2 lo = *reinterpret_cast<const uint2*>(pbuf + r * PITCH + cc * 16);
        const uint2 hi = *reinterpret_cast<const uint2*>(pbuf + r * PITCH + cc * 16 + 8);
#ifdef HVR_DBG_NOPSTORE
        if (m < p.M && p.scale == 12345.f)
#else
        if (m < p.M)
#endif
          *reinterpret_cast<uint4*>(Cb + ((long)m * p.ldc + n0) * (long)sizeof(T) + cc * 16) =
              make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    if (wn == 0 && frag_grp == 0) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = (wm * FM + i) * 16 + frag_row, m = m0 + row;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) sum += red[w * BM + row];
        if (m < p.M) {
          const int t = n0 / 128;
          mstat_g[(long)m * p.ntile + t] = tmax[i];  // log2 units
          lstat_g[(long)m * p.ntile + t] = sum;
        }
      }
    }
  }
}

// ---------------- host-side dispatch ----------------
template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS, bool RESPRE = false, int NS = 2>
static hipError_t launch_tile_impl(const GemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
  constexpr size_t stage = NS * (size_t)(BM + BN) * 128;
  constexpr size_t epi = EPI == EPI_SCORES ? (size_t)BM * (BN * sizeof(T) + 8) + (size_t)WN * BM * 4  // P~ tile + exchange
                                           : (size_t)FM * 16 * (BN + 4) * 4;                      // one wave-row block, f32
  constexpr size_t lds = stage > epi ? stage : epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = tile_kernel<T, WM, WN, FM, FN, EPI, GLDS, RESPRE, NS>;
  static std::atomic<unsigned> attr_set{0};   // (the attribute is per device)
  per_device_once(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int extra = (EPI == EPI_SCORES && sizeof(T) == 2 && p.tr_blocks > 0) ? p.tr_blocks : 0;   // (V^T workgroups: tile_kernel, GemmParams::tr_*)
  hipLaunchKernelGGL(kern, dim3(tiles + extra, (EPI == EPI_LINEAR || EPI == EPI_APPLY) && p.ksplit_steps > 0 ? p.ksplit_count : 1, p.batch > 1 ? p.batch : 1),
                     dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError();
}

template <typename T, int WM, int WN, int FM, int FN, int EPI, bool GLDS, int NS = 2>
static hipError_t launch_tile(const GemmParams& p, hipStream_t stream) {
  if constexpr (EPI == EPI_LINEAR && GLDS && sizeof(T) == 2) {
    const bool pre = p.resid && (p.N & 7) == 0 && ((p.ldr * 2) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.resid) & 15) == 0;
    if (pre) return launch_tile_impl<T, WM, WN, FM, FN, EPI, GLDS, true, NS>(p, stream);
  }
  return launch_tile_impl<T, WM, WN, FM, FN, EPI, GLDS, false, NS>(p, stream);
}

template <typename T, int EPI, bool GLDS>
static hipError_t dispatch_tile(const GemmParams& p, int tile, hipStream_t stream) {
  if constexpr (GLDS && !std::is_same<T, float>::value) {
    switch (tile) {
      case 1: if constexpr (EPI != EPI_SCORES) return launch_tile<T, 4, 1, 2, 4, EPI, GLDS>(p, stream); break;
      case 2: if constexpr (EPI == EPI_LINEAR) return launch_tile<T, 3, 2, 3, 8, EPI, GLDS>(p, stream); break;
      case 3: return launch_tile<T, 3, 2, 3, 4, EPI, GLDS>(p, stream);
      case 4: return launch_tile<T, 4, 2, 4, 4, EPI, GLDS>(p, stream);
      case 5: if constexpr (EPI != EPI_APPLY) return launch_tile<T, 4, 2, 4, 4, EPI, GLDS, 3>(p, stream); break;
      // (not for split half: the 6-wave 144 x 256 ring has no registers for the split K-step's fragment sets -- it would spill them)
      case 6: if constexpr (EPI == EPI_LINEAR && !std::is_same<T, f16s_t>::value) return launch_tile<T, 3, 2, 3, 8, EPI, GLDS, 3>(p, stream); break;
      // (the pipelined apply loop is written for two K-steps per 128-key block: split half has four and takes the double-buffered shapes)
      case 7: if constexpr (!(EPI == EPI_APPLY && std::is_same<T, f16s_t>::value)) return launch_tile<T, 3, 2, 3, 4, EPI, GLDS, 4>(p, stream); break;
      // (no apply pass on this shape: never chosen for it, and its untracked block-weight loads do not survive every register allocation --
      // check_asm_waits.py flagged it after round 4's loop changes)
      case 8: if constexpr (EPI != EPI_APPLY) return launch_tile<T, 2, 2, 4, 4, EPI, GLDS, 4>(p, stream); break;
      case 9: if constexpr (EPI == EPI_LINEAR) return launch_tile<T, 4, 1, 2, 4, EPI, GLDS, 3>(p, stream); break;
      case 10: if constexpr (EPI == EPI_LINEAR) return launch_tile<T, 1, 8, 9, 2, EPI, GLDS, 3>(p, stream); break;
      case 11: if constexpr (!(EPI == EPI_APPLY && std::is_same<T, f16s_t>::value)) return launch_tile<T, 1, 8, 9, 1, EPI, GLDS, 4>(p, stream); break;
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

}  // namespace hvr
