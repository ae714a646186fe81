// Fused positional-encoding + FlexibleNeRFModel forward for gfx950 (reference run_network,
// nerf/train_utils.py:72-89; positional_encoding nerf/nerf_helpers.py:115-159; FlexibleNeRFModel.forward
// nerf/models.py:233-256).
//
// Design (see mlp_layout.h and DESIGN.md):
//   * persistent workgroups (one per CU); each wave64 owns 32 sample points and walks them through the
//     whole network with the activations resident in registers: the 32x32 accumulator tile of layer l
//     (rows = features, column = lane = point) is fed back as the B operand of layer l+1, so no
//     activation ever touches LDS or HBM;
//   * weights arrive as a linear stream of 1 KiB MFMA-A pieces, LDS-DMA'd (global_load_lds_dwordx4) from
//     L2 into a 4-slot x 16 KiB LDS ring shared by the waves of the workgroup; one counted
//     `s_waitcnt vmcnt(N)` + one raw `s_barrier` per 16 pieces, three phases of prefetch always in flight;
//   * positional encodings are computed in registers straight into B-fragment layout (the reference
//     materialises a (P,90) tensor and recomputes the direction encoding per sample);
//   * bf16 mode: v_mfma_f32_32x32x16_bf16, 8 waves x 32 points per workgroup, 2 waves / SIMD;
//     fp32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), 4 waves x 32 points, 1 wave / SIMD.
// MFMA-bound: 1,186,816 FLOP per point (D8/W256) against 16 B written per point.
#include <utility>

#include "mlp_layout.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr int kRingPhases = 4;
constexpr int kSlotBytes = kPhasePieces * kPieceBytes;  // 16 KiB
constexpr int kRingBytes = kRingPhases * kSlotBytes;    // 64 KiB

template <bool BF16> struct Prec;
template <> struct Prec<true> {
  using BPiece = bf16x8;
  static constexpr int EPP = 8;    // k-values (elements) per lane per piece
  static constexpr int PPT = 2;    // pieces per 32-row hidden tile
  static constexpr int WAVES = 8;
};
template <> struct Prec<false> {
  using BPiece = f32x4;
  static constexpr int EPP = 4;
  static constexpr int PPT = 4;
  static constexpr int WAVES = 4;
};

struct FwdParams {
  const char* packed;   // [bias region][pieces]
  int bias_bytes;
  int total_pieces;
  int D;
  unsigned skip_mask;
  int use_viewdirs;
  int mode;             // 0: rays + z_vals, 1: pts (+ viewdirs), 2: encoded rows
  const float* rays;
  int ray_stride;
  const float* z;
  const float* pts;
  const float* viewdirs;
  const float* enc;
  int enc_ld;
  long long n_points;
  int S;                // samples per ray (ray = point / S)
  long long n_tiles;
  float* out;
  float fx[16];
  float fd[8];
};

// ---- weight pipeline: LDS ring fed by LDS-DMA -------------------------------------------------------
template <int WAVES>
struct Pipe {
  static constexpr int PER_WAVE = kPhasePieces / WAVES;
  char* ring;           // LDS
  const char* wsrc;     // global pieces, + lane*16 folded in
  unsigned total_bytes; // stream length in bytes
  unsigned q_issue;     // byte offset of the next phase to DMA (wave-uniform)
  unsigned slot_wr, slot_rd;
  unsigned wave;
  const char* rd;       // LDS read pointer of the current phase (+ lane*16)
  unsigned lane16;

  __device__ __forceinline__ void issue_phase() {
    char* dst = ring + slot_wr * kSlotBytes + wave * (PER_WAVE * kPieceBytes);
    const char* src = wsrc + q_issue + wave * (PER_WAVE * kPieceBytes);
#pragma unroll
    for (int e = 0; e < PER_WAVE; ++e) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + e * kPieceBytes),
          (__attribute__((address_space(3))) void*)(dst + e * kPieceBytes), 16, 0, 0);
    }
    q_issue += kSlotBytes;
    if (q_issue >= total_bytes) q_issue = 0;
    slot_wr = (slot_wr + 1) & (kRingPhases - 1);
  }

  // Called at every 16-piece boundary of the (compile-time laid out) consumption sequence.
  __device__ __forceinline__ void phase_begin() {
    // own DMAs of this phase have landed once at most the two younger phases remain outstanding;
    // lgkmcnt(0): this wave's LDS reads of the previous phase are complete before its slot is recycled.
    if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_phase();  // phase p+3 into the slot phase p-1 just vacated
    rd = ring + slot_rd * kSlotBytes + lane16;
    slot_rd = (slot_rd + 1) & (kRingPhases - 1);
  }
};

template <bool BF16>
__device__ __forceinline__ f32x16 mma_piece(f32x16 acc, f32x4 a_raw, typename Prec<BF16>::BPiece b) {
  if constexpr (BF16) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, a_raw);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[3], b[3], acc, 0, 0, 0);
    return acc;
  }
}

// One GEMM stage: NT_OUT output tiles, KH hidden pieces + KP positional-encoding pieces per tile.
// POS0 = piece position (mod 16) at which the stage starts; phase boundaries are compile-time.
template <bool BF16, int NT_OUT, int KH, int KP, int POS0, class PipeT, class BH, class BP>
__device__ __forceinline__ void run_stage(PipeT& pipe, f32x16 (&acc)[NT_OUT], const BH& bh, const BP& bp,
                                          const char* bias_lds /* this lane-half's 64 B of tile 0 */) {
  constexpr int KT = KH + KP;
  static_for<NT_OUT>([&](auto nt_c) {
    constexpr int nt = decltype(nt_c)::value;
    const f32x4* bptr = reinterpret_cast<const f32x4*>(bias_lds + nt * 128);
    const f32x4 b0 = bptr[0], b1 = bptr[1], b2 = bptr[2], b3 = bptr[3];
    f32x16 a;
    a[0] = b0[0]; a[1] = b0[1]; a[2] = b0[2]; a[3] = b0[3];
    a[4] = b1[0]; a[5] = b1[1]; a[6] = b1[2]; a[7] = b1[3];
    a[8] = b2[0]; a[9] = b2[1]; a[10] = b2[2]; a[11] = b2[3];
    a[12] = b3[0]; a[13] = b3[1]; a[14] = b3[2]; a[15] = b3[3];
    static_for<KT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int pos = POS0 + nt * KT + k;
      if constexpr (pos % kPhasePieces == 0) pipe.phase_begin();
      const f32x4 araw = *reinterpret_cast<const f32x4*>(pipe.rd + (pos % kPhasePieces) * kPieceBytes);
      if constexpr (k < KH) a = mma_piece<BF16>(a, araw, bh[k]);
      else a = mma_piece<BF16>(a, araw, bp[k - KH]);
    });
    acc[nt] = a;
  });
}

// accumulator tiles -> next stage's B pieces (ReLU optional), in place of the register-resident chain
template <bool BF16, int NT, bool RELU, class BH>
__device__ __forceinline__ void acc_to_pieces(const f32x16 (&acc)[NT], BH& bh) {
  using P = Prec<BF16>;
  static_for<NT>([&](auto nt_c) {
    constexpr int nt = decltype(nt_c)::value;
    static_for<P::PPT>([&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      typename P::BPiece piece;
#pragma unroll
      for (int e = 0; e < P::EPP; ++e) {
        float v = acc[nt][s * P::EPP + e];
        if (RELU) v = fmaxf(v, 0.0f);
        if constexpr (BF16) piece[e] = static_cast<__bf16>(v); else piece[e] = v;
      }
      bh[nt * P::PPT + s] = piece;
    });
  });
}

// ---- positional encoding straight into B-piece layout -------------------------------------------------
// Slot u of this lane-half (mlp_layout.h pe_slot_col): u < 6*(L/2): sin/cos of this half's frequencies;
// then identity (half 0: x, y; half 1: z); rest zero padding.
template <bool BF16, int L, int NPIECES, class BP>
__device__ __forceinline__ void encode_pieces(const float (&x)[3], const float* freqs, int h, BP& bp) {
  using P = Prec<BF16>;
  constexpr int NF = L / 2;
  float sv[NF][3], cv[NF][3];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float fr = h ? freqs[NF + f] : freqs[f];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float arg = x[c] * fr;
      if constexpr (BF16) {
        // hardware sin/cos take revolutions in [-256, 256]: reduce with fract first (bf16 output precision)
        const float rev = __builtin_amdgcn_fractf(arg * 0.15915494309189535f);
        sv[f][c] = __builtin_amdgcn_sinf(rev);
        cv[f][c] = __builtin_amdgcn_cosf(rev);
      } else {
        sincosf(arg, &sv[f][c], &cv[f][c]);
      }
    }
  }
  static_for<NPIECES>([&](auto p_c) {
    constexpr int p = decltype(p_c)::value;
    typename P::BPiece piece;
    static_for<P::EPP>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value;
      constexpr int u = p * P::EPP + e;
      float v;
      if constexpr (u < 6 * NF) v = ((u % 6) < 3) ? sv[u / 6][u % 3] : cv[u / 6][u % 3];
      else if constexpr (u == 6 * NF) v = h ? x[2] : x[0];
      else if constexpr (u == 6 * NF + 1) v = h ? 0.0f : x[1];
      else v = 0.0f;
      if constexpr (BF16) piece[e] = static_cast<__bf16>(v); else piece[e] = v;
    });
    bp[p] = piece;
  });
}

// Encoded-rows input (FlexibleNeRFModel.forward(x) call surface): gather this lane's slots from x.
template <bool BF16, int L, int NPIECES, class BP>
__device__ __forceinline__ void gather_pieces(const float* row, int h, BP& bp) {
  using P = Prec<BF16>;
  static_for<NPIECES>([&](auto p_c) {
    constexpr int p = decltype(p_c)::value;
    typename P::BPiece piece;
#pragma unroll
    for (int e = 0; e < P::EPP; ++e) {
      const int col = pe_slot_col(L, h, p * P::EPP + e);
      const float v = (col >= 0) ? row[col] : 0.0f;
      if constexpr (BF16) piece[e] = static_cast<__bf16>(v); else piece[e] = v;
    }
    bp[p] = piece;
  });
}

template <int W, int LX, int LD, bool BF16>
__global__ __launch_bounds__(Prec<BF16>::WAVES * 64, BF16 ? 2 : 1) void mlp_forward_kernel(FwdParams p) {
  using P = Prec<BF16>;
  using BPiece = typename P::BPiece;
  constexpr int NT = W / 32;
  constexpr int KH = NT * P::PPT;                      // hidden pieces of a W-wide input
  constexpr int KXP = round_up(3 + 6 * LX, 16) / (2 * P::EPP);  // PE xyz pieces
  constexpr int KDP = round_up(3 + 6 * LD, 16) / (2 * P::EPP);  // PE dir pieces
  constexpr int WAVES = P::WAVES;
  constexpr int PTS_PER_WG = 32 * WAVES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* bias_lds = smem + kRingBytes;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;
  const int j = lane & 31;

  // biases -> LDS once per workgroup (fp32, pre-permuted [tile][half][16])
  {
    const f32x4* g = reinterpret_cast<const f32x4*>(p.packed);
    f32x4* l = reinterpret_cast<f32x4*>(bias_lds);
    for (int i = threadIdx.x; i < p.bias_bytes / 16; i += WAVES * 64) l[i] = g[i];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  Pipe<WAVES> pipe;
  pipe.ring = ring;
  pipe.lane16 = lane * 16;
  pipe.wsrc = p.packed + p.bias_bytes + lane * 16;
  pipe.total_bytes = static_cast<unsigned>(p.total_pieces) * kPieceBytes;
  pipe.q_issue = 0;
  pipe.slot_wr = 0;
  pipe.slot_rd = 0;
  pipe.wave = wave;
  pipe.rd = ring + lane * 16;
  pipe.issue_phase();
  pipe.issue_phase();
  pipe.issue_phase();

  const char* bias_half = bias_lds + h * 64;

  for (long long tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    // ---- inputs of this lane's point (lanes j and j+32 share a point) ----
    long long pt = tile * PTS_PER_WG + wave * 32 + j;
    const bool live = pt < p.n_points;
    if (!live) pt = p.n_points - 1;
    BPiece bx[KXP];
    float vd[3] = {0.f, 0.f, 0.f};
    if (p.mode == 2) {
      const float* row = p.enc + pt * p.enc_ld;
      gather_pieces<BF16, LX, KXP>(row, h, bx);
    } else {
      float x[3];
      if (p.mode == 0) {
        const long long ray = pt / p.S;
        const float* r = p.rays + ray * p.ray_stride;
        const float zv = p.z[pt];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = r[c] + r[3 + c] * zv;  // plain mul then add (train_utils.py:136)
        if (p.use_viewdirs) { vd[0] = r[8]; vd[1] = r[9]; vd[2] = r[10]; }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = p.pts[pt * 3 + c];
        if (p.use_viewdirs) {
          const long long ray = pt / p.S;
#pragma unroll
          for (int c = 0; c < 3; ++c) vd[c] = p.viewdirs[ray * 3 + c];
        }
      }
      encode_pieces<BF16, LX, KXP>(x, p.fx, h, bx);
    }
    // inputs are ordinary VMEM loads: drain so the counted vmcnt below only ever sees weight DMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    BPiece bh[KH];
    f32x16 acc[NT];
    int bias_tile = 0;
    // ---- layer1: xyz encoding -> W, no activation (models.py:238) ----
    {
      BPiece none[1];
      run_stage<BF16, NT, 0, KXP, 0>(pipe, acc, none, bx, bias_half);
      acc_to_pieces<BF16, NT, false>(acc, bh);
      bias_tile += NT;
    }
    // ---- layers_xyz[i]: (cat(x, xyz) on skip layers) -> W, ReLU (models.py:239-246) ----
    for (int i = 0; i < p.D - 1; ++i) {
      if ((p.skip_mask >> i) & 1u) {
        run_stage<BF16, NT, KH, KXP, 0>(pipe, acc, bh, bx, bias_half + bias_tile * 128);
      } else {
        BPiece none[1];
        run_stage<BF16, NT, KH, 0, 0>(pipe, acc, bh, none, bias_half + bias_tile * 128);
      }
      acc_to_pieces<BF16, NT, true>(acc, bh);
      bias_tile += NT;
    }
    float out4[4];
    if (p.use_viewdirs) {
      // ---- fc_alpha (extra tile, streamed first) + fc_feat with ReLU (models.py:248-249) ----
      BPiece none[1];
      constexpr int POS_A = 0;
      {
        f32x16 at[1];
        run_stage<BF16, 1, KH, 0, POS_A>(pipe, at, bh, none, bias_half + bias_tile * 128);
        out4[3] = at[0][0];  // row 0 of the tile: lanes 0..31, register 0
      }
      constexpr int POS_F = (POS_A + KH) % kPhasePieces;
      run_stage<BF16, NT, KH, 0, POS_F>(pipe, acc, bh, none, bias_half + (bias_tile + 1) * 128);
      acc_to_pieces<BF16, NT, true>(acc, bh);
      bias_tile += NT + 1;
      // ---- layers_dir[0] on cat(feat, view) -> W/2, ReLU (models.py:250-252) ----
      BPiece bd[KDP];
      if (p.mode == 2) gather_pieces<BF16, LD, KDP>(p.enc + pt * p.enc_ld + (3 + 6 * LX), h, bd);
      else encode_pieces<BF16, LD, KDP>(vd, p.fd, h, bd);
      if (p.mode == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      constexpr int POS_D = (POS_F + NT * KH) % kPhasePieces;
      f32x16 accd[NT / 2];
      run_stage<BF16, NT / 2, KH, KDP, POS_D>(pipe, accd, bh, bd, bias_half + bias_tile * 128);
      BPiece bg[KH / 2];
      acc_to_pieces<BF16, NT / 2, true>(accd, bg);
      bias_tile += NT / 2;
      // ---- fc_rgb (models.py:253) ----
      constexpr int POS_R = (POS_D + (NT / 2) * (KH + KDP)) % kPhasePieces;
      f32x16 ar[1];
      run_stage<BF16, 1, KH / 2, 0, POS_R>(pipe, ar, bg, none, bias_half + bias_tile * 128);
      out4[0] = ar[0][0]; out4[1] = ar[0][1]; out4[2] = ar[0][2];
      // the stream is padded to a whole number of phases; nothing to skip: every tail variant ends aligned
      static_assert((POS_R + KH / 2) % kPhasePieces == 0, "tail must end on a phase boundary");
    } else {
      BPiece none[1];
      f32x16 ao[1];
      run_stage<BF16, 1, KH, 0, 0>(pipe, ao, bh, none, bias_half + bias_tile * 128);
      out4[0] = ao[0][0]; out4[1] = ao[0][1]; out4[2] = ao[0][2]; out4[3] = ao[0][3];
      if constexpr (KH % kPhasePieces != 0) {
        // consume the padding phase remainder: nothing to read, positions are per-phase relative
      }
    }
    if (live && h == 0) {
      f32x4 o;
      o[0] = out4[0]; o[1] = out4[1]; o[2] = out4[2]; o[3] = out4[3];
      *reinterpret_cast<f32x4*>(p.out + pt * 4) = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ---- pack kernel: nn.Linear tensors -> bias tiles + MFMA-A piece stream ---------------------------------
struct PackPtrs {
  const float* w[kMaxStages];
  const float* b[kMaxStages];
};

template <bool BF16>
__global__ void pack_kernel(NetLayout L, PackPtrs ptrs, char* __restrict__ packed) {
  using P = Prec<BF16>;
  const int KX = round_up(3 + 6 * L.LX, 16), KD = round_up(3 + 6 * L.LD, 16);
  (void)KX; (void)KD;
  // bias tiles
  const int n_bias = L.total_bias_tiles * 32;
  float* bias_out = reinterpret_cast<float*>(packed);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < L.bias_bytes / 4; idx += gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (idx < n_bias) {
      const int tile = idx / 32, hh = (idx % 32) / 16, r = idx % 16;
      int s = 0;
      while (s + 1 < L.n_stages && L.st[s + 1].bias0 <= tile) ++s;
      const StageDesc& st = L.st[s];
      int ts = tile - st.bias0;
      const int row_in_tile = acc_row(r, hh);
      if (st.src2 >= 0) {
        if (ts == 0) v = (row_in_tile == 0) ? ptrs.b[st.src2][0] : 0.0f;
        else { const int n = (ts - 1) * 32 + row_in_tile; v = (n < st.n_real) ? ptrs.b[st.src][n] : 0.0f; }
      } else {
        const int n = ts * 32 + row_in_tile;
        v = (n < st.n_real) ? ptrs.b[st.src][n] : 0.0f;
      }
    }
    bias_out[idx] = v;
  }
  // weight pieces
  char* wout = packed + L.bias_bytes;
  const long long n_elems = static_cast<long long>(L.total_pieces) * 64 * P::EPP;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < n_elems;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int e = static_cast<int>(idx % P::EPP);
    const int lane = static_cast<int>((idx / P::EPP) % 64);
    const int piece = static_cast<int>(idx / (P::EPP * 64));
    const int i = lane & 31, hh = lane >> 5;
    float v = 0.0f;
    int s = 0;
    while (s + 1 < L.n_stages && L.st[s + 1].piece0 <= piece) ++s;
    const StageDesc& st = L.st[s];
    const int rel = piece - st.piece0;
    if (rel < st.n_tiles * st.pieces_per_tile) {
      const int ts = rel / st.pieces_per_tile;
      const int k = rel % st.pieces_per_tile;
      const int kh = st.hidden_in / (2 * P::EPP);  // hidden pieces
      int col;
      if (k < kh) {
        const int nt_in = k / P::PPT;
        const int r = (k % P::PPT) * P::EPP + e;
        col = st.col_hidden0 + nt_in * 32 + acc_row(r, hh);
      } else {
        const int u = (k - kh) * P::EPP + e;
        const int pc = pe_slot_col(st.pe_kind == 1 ? L.LX : L.LD, hh, u);
        col = (pc >= 0) ? st.col_pe0 + pc : -1;
      }
      if (col >= 0) {
        if (st.src2 >= 0) {
          if (ts == 0) v = (i == 0) ? ptrs.w[st.src2][col] : 0.0f;
          else { const int n = (ts - 1) * 32 + i; v = (n < st.n_real) ? ptrs.w[st.src][static_cast<long long>(n) * st.ld + col] : 0.0f; }
        } else {
          const int n = ts * 32 + i;
          v = (n < st.n_real) ? ptrs.w[st.src][static_cast<long long>(n) * st.ld + col] : 0.0f;
        }
      }
    }
    if constexpr (BF16) reinterpret_cast<__bf16*>(wout)[idx] = static_cast<__bf16>(v);
    else reinterpret_cast<float*>(wout)[idx] = v;
  }
}

template <int W, int LX, int LD, bool BF16>
static int launch_forward(const FwdParams& p, hipStream_t stream) {
  auto kern = mlp_forward_kernel<W, LX, LD, BF16>;
  const size_t lds = kRingBytes + p.bias_bytes;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return -static_cast<int>(e); }
    attr_set = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const long long grid = p.n_tiles < cus ? p.n_tiles : cus;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(Prec<BF16>::WAVES * 64), lds, stream, p);
  return check_launch("mlp_forward");
}

static int dispatch_forward(const dn_mlp_desc& d, int precision, FwdParams& p, hipStream_t stream) {
  const bool bf = precision == DN_PREC_BF16;
  p.n_tiles = (p.n_points + (bf ? 256 : 128) - 1) / (bf ? 256 : 128);
#define DN_CASE(W_, LX_)                                                                     \
  if (d.hidden_size == W_ && d.num_encoding_fn_xyz == LX_)                                   \
    return bf ? launch_forward<W_, LX_, 4, true>(p, stream) : launch_forward<W_, LX_, 4, false>(p, stream);
  DN_CASE(256, 10)
  DN_CASE(128, 10)
  DN_CASE(256, 6)
  DN_CASE(128, 6)
#undef DN_CASE
  set_error("mlp_forward: no kernel instance for W=%d L_xyz=%d", d.hidden_size, d.num_encoding_fn_xyz);
  return DN_E_UNSUPPORTED;
}

void fill_freqs(float* f, int num_fns, int log_sampling);  // rays_sampling.hip

static int setup_params(const dn_mlp_desc* desc, int precision, const void* packed, FwdParams* p) {
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  NetLayout L;
  build_layout(*desc, precision, &L);
  *p = FwdParams{};
  p->packed = static_cast<const char*>(packed);
  p->bias_bytes = L.bias_bytes;
  p->total_pieces = L.total_pieces;
  p->D = desc->num_layers;
  p->skip_mask = L.skip_mask;
  p->use_viewdirs = desc->use_viewdirs;
  fill_freqs(p->fx, desc->num_encoding_fn_xyz, desc->log_sampling_xyz);
  if (desc->use_viewdirs) fill_freqs(p->fd, desc->num_encoding_fn_dir, desc->log_sampling_dir);
  return 0;
}

}  // namespace dn

using namespace dn;

extern "C" size_t dn_mlp_packed_bytes(const dn_mlp_desc* desc, int precision) {
  if (validate_desc(desc, precision)) return 0;
  NetLayout L;
  build_layout(*desc, precision, &L);
  return static_cast<size_t>(L.bias_bytes) + static_cast<size_t>(L.total_pieces) * kPieceBytes;
}

extern "C" int dn_mlp_pack(const dn_mlp_desc* desc, int precision, const float* const* h_weights,
                           const float* const* h_biases, void* packed, dn_stream_t stream) {
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(h_weights && h_biases && packed, "dn_mlp_pack: NULL pointer");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "dn_mlp_pack: packed buffer must be 16-byte aligned");
  NetLayout L;
  build_layout(*desc, precision, &L);
  const int n_params = desc->num_layers + (desc->use_viewdirs ? 4 : 1);
  PackPtrs ptrs{};
  for (int i = 0; i < n_params; ++i) {
    DN_REQUIRE(h_weights[i] && h_biases[i], "dn_mlp_pack: parameter %d is NULL", i);
    ptrs.w[i] = h_weights[i];
    ptrs.b[i] = h_biases[i];
  }
  if (precision == DN_PREC_BF16)
    hipLaunchKernelGGL(pack_kernel<true>, dim3(512), dim3(256), 0, as_stream(stream), L, ptrs, static_cast<char*>(packed));
  else
    hipLaunchKernelGGL(pack_kernel<false>, dim3(512), dim3(256), 0, as_stream(stream), L, ptrs, static_cast<char*>(packed));
  return check_launch("dn_mlp_pack");
}

extern "C" int dn_run_network(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts,
                              const float* viewdirs, const float* rays, int ray_stride, const float* z_vals,
                              int64_t n_rays, int samples_per_ray, float* out, dn_stream_t stream) {
  FwdParams p;
  int rc = setup_params(desc, precision, packed, &p);
  if (rc) return rc;
  DN_REQUIRE(packed && out && n_rays >= 0 && samples_per_ray >= 1, "dn_run_network: bad arguments");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dn_run_network: out must be 16-byte aligned");
  if (pts != nullptr) {
    DN_REQUIRE(!desc->use_viewdirs || viewdirs, "dn_run_network: viewdirs required with use_viewdirs");
    p.mode = 1; p.pts = pts; p.viewdirs = viewdirs;
  } else {
    DN_REQUIRE(rays && z_vals, "dn_run_network: need pts, or rays + z_vals");
    DN_REQUIRE(ray_stride >= (desc->use_viewdirs ? 11 : 8), "dn_run_network: ray_stride too small");
    p.mode = 0; p.rays = rays; p.ray_stride = ray_stride; p.z = z_vals;
  }
  p.n_points = n_rays * samples_per_ray;
  p.S = samples_per_ray;
  p.out = out;
  if (p.n_points == 0) return 0;
  return dispatch_forward(*desc, precision, p, as_stream(stream));
}

extern "C" int dn_mlp_forward_encoded(const dn_mlp_desc* desc, int precision, const void* packed, const float* x,
                                      int64_t n_rows, float* out, dn_stream_t stream) {
  FwdParams p;
  int rc = setup_params(desc, precision, packed, &p);
  if (rc) return rc;
  DN_REQUIRE(packed && x && out && n_rows >= 0, "dn_mlp_forward_encoded: bad arguments");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dn_mlp_forward_encoded: out must be 16-byte aligned");
  p.mode = 2; p.enc = x;
  p.enc_ld = (3 + 6 * desc->num_encoding_fn_xyz) + (desc->use_viewdirs ? 3 + 6 * desc->num_encoding_fn_dir : 0);
  p.n_points = n_rows; p.S = 1; p.out = out;
  if (n_rows == 0) return 0;
  return dispatch_forward(*desc, precision, p, as_stream(stream));
}
