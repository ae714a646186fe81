// Training kernels of the fused network path: the backward-data (dL/dX) chain and the native->plain unpack.
//
// The training forward is mlp_forward_kernel<..., SAVE=true> (mlp_fused.hip): it keeps every stage's output
// pieces and a 128-bit ReLU mask word per lane per stage.  The backward chain below is the same register-resident
// MFMA chain run on the TRANSPOSED weight stream: for 32 points per wave,
//     dX^T[K x 32] = W^T[K x N] . dY^T[N x 32],   dY = dX_next (.) relu'(Y)
// so the gradient tile produced by one stage's MFMAs is (after masking) the B operand of the next stage, exactly
// like the activations in the forward.  Every masked gradient dL/d(pre-activation) is written out once in the
// wave-native piece layout; dW/db are formed from those and the saved activations by the weight-gradient kernel at
// the end of this file (bf16 MFMA, or exact-fp32 MFMA in the parity mode).
// In practice bound by the HBM write stream of the stored gradients (3.6 KiB/point at D8/W256), not by the MFMAs.
#include "mlp_internal.h"
#include "mlp_geo48.h"
#include <atomic>
#include <vector>
#include <cmath>

namespace dn {

struct BwdParams {
  const char* packed;      // backward (transposed) piece stream, no bias region
  int total_pieces;
  int D;
  int use_viewdirs;
  const float* g_out;      // (P,4) dL/d[r,g,b,sigma] of the raw radiance field
  const char* masks;       // ReLU mask words from the training forward
  int mask_words;
  long long n_points;
  long long n_tiles;
  char* grads;             // [tile32][grad_pieces][64][16 B]
  int grad_pieces;         // (S8 kernels: 1 KiB units = pairs of pieces)
  int gslot_dirout, gslot_feat, gslot_trunk0, gslot_layer1, gslot_out;
  float grad_scale;        // S8 kernels: the power of two the gradients are multiplied by before they are rounded to e5m2
};

// dn_set_s8_grad_scale: gradient scale of the 8-bit saved tensors.  Process-wide, not per thread: PyTorch runs the backward of a
// step on its autograd thread, and the backward-data kernel (which multiplies) and the weight-gradient kernel (which divides)
// must see the value the caller set on its own thread.
static std::atomic<float> g_s8_grad_scale{0.0f};   // 0: per launch, from the largest upstream gradient (safe for any loss reduction / scaling)

constexpr int kBwdWaveLds = 6 * kPieceBytes;  // per wave: 2 output-gradient slots + 4 mask-word slots (1 KiB each)

template <int W, int BF16, bool S8 = false>   // S8: the saved gradients at 8 bits (e5m2 x grad_scale), two pieces per 1 KiB unit
__global__ __launch_bounds__((waves_of<BF16, 1>() * 64), (BF16 ? 2 : 1)) void mlp_backward_kernel(BwdParams p) {
  using P = Prec<BF16>;
  using BPiece = typename P::BPiece;
  constexpr int PT = 1;
  constexpr int NT = W / 32;
  constexpr int KH = NT * P::PPT;
  constexpr int WAVES = waves_of<BF16, 1>();

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;
  const int j = lane & 31;
  // per-wave staging: the output gradient rows of the NEXT tile and the ReLU mask words two stages ahead arrive by
  // LDS-DMA; no VGPR-destination global load exists in the tile loop (a wait on one would also wait for every older
  // store - HBM latency - and for the weight DMAs)
  char* wbuf = smem + kRingBytes + wave * kBwdWaveLds;
  const unsigned wbuf_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)wbuf));
  const int n_masks = p.mask_words;          // stage q < n_masks applies mask word n_masks-1-q; stage n_masks: none

  auto issue_gout = [&](long long tile, int slot) {
    long long pt = (tile * WAVES + wave) * 32 + j;
    if (pt >= p.n_points) pt = p.n_points - 1;  // clamped lanes recompute a valid point; their stores hit padding tiles
    dma16_lanes(p.g_out + pt * 4, wbuf_addr + slot * kPieceBytes);
  };
  auto issue_mask = [&](long long tile, int q) {  // mask word of stage q of `tile` -> slot q & 3
    const char* src = p.masks + (((tile * WAVES + wave) * p.mask_words + (n_masks - 1 - q)) * 64 + lane) * 16;
    dma16_lanes(src, wbuf_addr + (2 + (q & 3)) * kPieceBytes);
  };

  Pipe<WAVES> pipe;
  pipe.ring = ring;
  pipe.ring_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring));
  pipe.lane16 = lane * 16;
  pipe.wsrc = p.packed;
  pipe.total_bytes = static_cast<unsigned>(p.total_pieces) * kPieceBytes;
  pipe.q_issue = 0;
  pipe.slot_wr = 0;
  pipe.wave = wave;
  issue_gout(blockIdx.x, 0);
  issue_mask(blockIdx.x, 0);
  if (n_masks > 1) issue_mask(blockIdx.x, 1);
#pragma unroll
  for (int ph = 0; ph < kRingPhases - 1; ++ph) pipe.issue_phase();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  pipe.slot_nxt = 0;
  pipe.rd_cur = ring + lane * 16;
  pipe.rd_nxt = ring + lane * 16;
#pragma unroll
  for (int e = 0; e < kPrefetch; ++e) pipe.af[e] = *reinterpret_cast<const f32x4*>(pipe.rd_nxt + e * kPieceBytes);

  int g_slot = 0;
  for (long long tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const long long tile32 = tile * WAVES + wave;
    const long long nxt = tile + gridDim.x;
    const f32x4 g = *reinterpret_cast<const f32x4*>(wbuf + g_slot * kPieceBytes + lane * 16);
    const char* grad_tile = uniform_ptr(p.grads + tile32 * p.grad_pieces * kPieceBytes);
    auto store_grad = [&](int slot, const BPiece& v) {
#ifndef DN_EXP_NOSAVE
      if constexpr (!S8) store16_uniform(grad_tile + static_cast<long long>(slot) * kPieceBytes, pipe.lane16, v);
#endif
    };
    // S8: pieces `slot` (even) and `slot + 1` as one unit of 8 + 8 bytes per lane
    auto store_grad_pair = [&](int slot, const BPiece& lo, const BPiece& hi) {
#ifndef DN_EXP_NOSAVE
      if constexpr (S8 && BF16 == 1) {
        unsigned w[4];
        piece_to_8bit<true>(lo, p.grad_scale, w[0], w[1]);
        piece_to_8bit<true>(hi, p.grad_scale, w[2], w[3]);
        store16_uniform(grad_tile + static_cast<long long>(slot >> 1) * kPieceBytes, pipe.lane16, make_uint4(w[0], w[1], w[2], w[3]));
      }
#endif
    };
    // Start of stage q: fetch this stage's mask word from its LDS slot, then stage what will be needed two stages on
    // (same tile, or the first two stages / the output gradient of the next tile) into the slot just read or one idle
    // for two stages.  The LDS read is complete (lgkmcnt(0)) before a DMA may overwrite the slot.
    auto stage_begin = [&](int q) {
      uint4 mw = make_uint4(~0u, ~0u, ~0u, ~0u);
      if (q < n_masks) mw = *reinterpret_cast<const uint4*>(wbuf + (2 + (q & 3)) * kPieceBytes + lane * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r = q + 2;
      if (r < n_masks) issue_mask(tile, r);
      else if (r > n_masks && nxt < p.n_tiles) {   // r == n_masks + 1 / + 2: stages 0 / 1 of the next tile
        const int r2 = r - (n_masks + 1);
        if (r2 < n_masks) issue_mask(nxt, r2);
        if (r2 == 0) issue_gout(nxt, g_slot ^ 1);
      }
      return mw;
    };

    // masked gradient tile -> next stage's B pieces + one store per piece
    auto emit_grad = [&](auto nt_c, const f32x16& acc_in, const uint4& mw, BPiece* bout, int gslot) {
      constexpr int nt = decltype(nt_c)::value;
      if constexpr (BF16) {
        // convert first, then mask the PACKED pairs: dword j of the tile's two pieces holds registers 2j / 2j+1, whose
        // mask bits sit at (j + 8*(nt&1)) and 16 + that (relu_mask_bit): (bits & 0x00010001) * 0xFFFF = the AND mask
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const unsigned word = nt / 2 == 0 ? mw.x : nt / 2 == 1 ? mw.y : nt / 2 == 2 ? mw.z : mw.w;
        static_for<P::PPT>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          u32x4 bits = __builtin_bit_cast(u32x4, make_piece<BF16, false, s>(acc_in));
#pragma unroll
          for (int d = 0; d < 4; ++d) bits[d] &= ((word >> (s * 4 + d + 8 * (nt & 1))) & 0x00010001u) * 0xFFFFu;
          const BPiece piece = __builtin_bit_cast(BPiece, bits);
          bout[nt * P::PPT + s] = piece;
          store_grad(gslot + nt * P::PPT + s, piece);
        });
        if constexpr (S8) store_grad_pair(gslot + nt * P::PPT, bout[nt * P::PPT], bout[nt * P::PPT + 1]);
      } else {
        f32x16 acc = acc_in;
        const unsigned words[4] = {mw.x, mw.y, mw.z, mw.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool on = (words[relu_mask_bit(nt, r) / 32] >> (relu_mask_bit(nt, r) % 32)) & 1u;
          acc[r] = on ? acc[r] : 0.0f;
        }
        static_for<P::PPT>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          const BPiece piece = make_piece<BF16, false, s>(acc);
          bout[nt * P::PPT + s] = piece;
          store_grad(gslot + nt * P::PPT + s, piece);
        });
      }
    };

    BPiece ba[PT][KH], bb[PT][KH];
    BPiece none[PT][1];
    auto no_pe = [&](int, int) { return BPiece{}; };
    int q = 0;  // stage counter of this tile
    if (p.use_viewdirs) {
      // custom input pieces: element (half 0, e) carries k = e
      BPiece crgb{}, calpha{};
      if (h == 0) {
        if constexpr (BF16) {
          crgb[0] = static_cast<__bf16>(g[0]); crgb[1] = static_cast<__bf16>(g[1]); crgb[2] = static_cast<__bf16>(g[2]);
          calpha[0] = static_cast<__bf16>(g[3]);
        } else {
          crgb[0] = g[0]; crgb[1] = g[1]; crgb[2] = g[2];
          calpha[0] = g[3];
        }
      }
      store_grad(p.gslot_out, crgb);        // for dW(fc_rgb)
      store_grad(p.gslot_out + 1, calpha);  // for dW(fc_alpha)
      if constexpr (S8) store_grad_pair(p.gslot_out, crgb, calpha);
      // ---- d g = fc_rgb^T d rgb, masked by relu'(layers_dir.0 out) ----
      uint4 mw = stage_begin(q++);
      auto c_rgb = [&](int, int) { return crgb; };
      run_stage<BF16, PT, NT / 2, 0, 1, 0, false>(pipe, none, c_rgb, nullptr, [&](auto nt_c, auto, const f32x16& acc) {
        emit_grad(nt_c, acc, mw, ba[0], p.gslot_dirout);
      });
      // ---- d feat = layers_dir.0[:, :W]^T d dirpre, masked by relu'(fc_feat out) ----
      constexpr int P1 = (NT / 2) % kPhasePieces;
      mw = stage_begin(q++);
      run_stage<BF16, PT, NT, KH / 2, 0, P1, false>(pipe, ba, no_pe, nullptr, [&](auto nt_c, auto, const f32x16& acc) {
        emit_grad(nt_c, acc, mw, bb[0], p.gslot_feat);
      });
      // ---- d h = fc_feat^T d featpre + fc_alpha^T d alpha, masked by relu'(layers_xyz[D-2] out) ----
      constexpr int P2 = (P1 + NT * (KH / 2)) % kPhasePieces;
      mw = stage_begin(q++);
      auto c_alpha = [&](int, int) { return calpha; };
      run_stage<BF16, PT, NT, KH, 1, P2, false>(pipe, bb, c_alpha, nullptr, [&](auto nt_c, auto, const f32x16& acc) {
        emit_grad(nt_c, acc, mw, ba[0], p.gslot_trunk0 + (p.D - 2) * KH);
      });
    } else {
      // ---- d h = fc_out^T d out, masked by relu'(layers_xyz[D-2] out) ----
      BPiece cout{};
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (BF16) cout[c] = static_cast<__bf16>(g[c]); else cout[c] = g[c];
        }
      }
      store_grad(p.gslot_out, cout);  // for dW(fc_out)
      if constexpr (S8) store_grad_pair(p.gslot_out, cout, BPiece{});
      const uint4 mw = stage_begin(q++);
      auto c_out = [&](int, int) { return cout; };
      run_stage<BF16, PT, NT, 0, 1, 0, false>(pipe, none, c_out, nullptr, [&](auto nt_c, auto, const f32x16& acc) {
        emit_grad(nt_c, acc, mw, ba[0], p.gslot_trunk0 + (p.D - 2) * KH);
      });
    }
    // ---- trunk, i = D-2 .. 0:  d x_i = layers_xyz[i][:, :W]^T d pre_i, masked by relu'(x_i) (x_0 = layer1 out: the
    //      last stage has no mask word - stage_begin hands back all ones).  The gradient sets ping-pong between `ba`
    //      and `bb`.  Positions: both head variants leave the same offset mod 16 per precision/W.
    constexpr int PV = ((NT / 2) + NT * (KH / 2) + NT * (KH + 1)) % kPhasePieces;  // viewdirs head
    constexpr int PN = NT % kPhasePieces;                                            // fc_out head
    auto trunk = [&](auto pos_c) {
      constexpr int POS = decltype(pos_c)::value;
      auto layer = [&](int i, const BPiece (&bin)[PT][KH], BPiece (&bout)[PT][KH]) __attribute__((always_inline)) {
        const uint4 mw = stage_begin(q++);
        const int gslot = i > 0 ? p.gslot_trunk0 + (i - 1) * KH : p.gslot_layer1;
        run_stage<BF16, PT, NT, KH, 0, POS, false>(pipe, bin, no_pe, nullptr, [&](auto nt_c, auto, const f32x16& acc) {
          emit_grad(nt_c, acc, mw, bout[0], gslot);
        });
      };
      int i = p.D - 2;
      for (; i >= 1; i -= 2) {
        layer(i, ba, bb);
        layer(i - 1, bb, ba);
      }
      if (i == 0) layer(0, ba, bb);
      if constexpr (POS % kPhasePieces != 0) pipe.template skip<POS, kPhasePieces - POS>();
    };
    static_assert((NT * KH) % kPhasePieces == 0, "trunk stages must preserve the phase offset");
    if (p.use_viewdirs) trunk(std::integral_constant<int, PV>{});
    else trunk(std::integral_constant<int, PN>{});
    g_slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ---- native piece layout -> plain (P, width) fp32 rows -------------------------------------------------------
// kind 0: hidden vector (feature = 32*(q/PPT) + acc_row(...)); kind 1/2: xyz / dir positional encoding.
template <int BF16>
__global__ void unpack_kernel(const char* __restrict__ native, int pieces_per_tile, int slot0, int n_pieces, int kind,
                              int L, long long n_points, float* __restrict__ out, int ld_out, int col0) {
  using P = Prec<BF16>;
  const long long total = ((n_points + 31) / 32) * n_pieces * 64 * P::EPP;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int e = static_cast<int>(idx % P::EPP);
    const int lane = static_cast<int>((idx / P::EPP) % 64);
    const int q = static_cast<int>((idx / (P::EPP * 64)) % n_pieces);
    const long long tile32 = idx / (static_cast<long long>(P::EPP) * 64 * n_pieces);
    const int j = lane & 31, hh = lane >> 5;
    const long long pt = tile32 * 32 + j;
    if (pt >= n_points) continue;
    int col;
    if (kind == 0) col = (q / P::PPT) * 32 + acc_row((q % P::PPT) * P::EPP + e, hh);
    else col = pe_slot_col(L, hh, q * P::EPP + e);
    if (col < 0) continue;
    const char* src = native + ((tile32 * pieces_per_tile + slot0 + q) * 64 + lane) * 16;
    float v;
    if constexpr (BF16) v = static_cast<float>(reinterpret_cast<const __bf16*>(src)[e]);
    else v = reinterpret_cast<const float*>(src)[e];
    out[pt * ld_out + col0 + col] = v;
  }
}

template <int W, int BF16, bool S8 = false>
static int launch_backward(BwdParams p, hipStream_t stream) {
  auto kern = mlp_backward_kernel<W, BF16, S8>;
  constexpr int WAVES = waves_of<BF16, 1>();
  p.n_tiles = (p.n_points + WAVES * 32 - 1) / (WAVES * 32);
  if (int rc = ensure_big_lds(reinterpret_cast<const void*>(kern))) return rc;
  const int cus = device_cus();
  const long long grid = p.n_tiles < cus ? p.n_tiles : cus;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(WAVES * 64), kRingBytes + WAVES * kBwdWaveLds, stream, p);
  return check_launch("mlp_backward");
}

static long long padded_tiles(long long n_points, int precision) {
  // both training kernels process whole workgroup tiles; buffers are sized for the padded tile count
  const int per_wg = 32 * (precision != DN_PREC_F32 ? 8 : 4);
  return (n_points + per_wg - 1) / per_wg * (per_wg / 32);
}

}  // namespace dn

using namespace dn;

extern "C" int dn_mlp_train_sizes(const dn_mlp_desc* desc, int precision, int64_t n_points, size_t* act_bytes,
                                  size_t* mask_bytes, size_t* grad_bytes) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(precision != DN_PREC_F16, "training kernels exist for fp32 and bf16 (fp16 is a render-only mode)");
  DN_REQUIRE(n_points >= 0 && act_bytes && mask_bytes && grad_bytes, "dn_mlp_train_sizes: bad arguments");
  DN_REQUIRE(desc->num_encoding_fn_xyz == 10, "dn_mlp_train_sizes: training kernels are built for L_xyz = 10");
  if (s8) {   // s8-48 layout (mlp_geo48.h): units per 16-point group, two groups per 32-point record, whole 384-point tiles
    if (!g48_train_supported(*desc)) {
      set_error("dn_mlp_train_sizes: the 8-bit-saved-tensor training kernels run the 48-point geometry (W in {128, 256}, L_xyz = 10, a depth whose bias rows fit its LDS); train this network with DN_PREC_BF16");
      return DN_E_UNSUPPORTED;
    }
    TrainLayout48 t8;
    build_train_layout48(*desc, &t8);
    const size_t records = static_cast<size_t>(g48_padded_records(n_points));
    *act_bytes = records * 2 * t8.act_units * kPieceBytes;
    *mask_bytes = static_cast<size_t>(g48_mask_wave_tiles(n_points)) * t8.mask_stages * 2 * kPieceBytes;
    *grad_bytes = records * 2 * t8.grad_units * kPieceBytes + kS8BlockBytes;   // + the statistics / scale record (mlp_geo48.h)
    return 0;
  }
  TrainLayout t;
  build_train_layout(*desc, precision, &t);
  const size_t tiles = static_cast<size_t>(padded_tiles(n_points, precision));
  *act_bytes = tiles * t.act_pieces * kPieceBytes;
  *mask_bytes = tiles * t.mask_words * kPieceBytes;
  *grad_bytes = tiles * t.grad_pieces * kPieceBytes;
  return 0;
}

extern "C" size_t dn_mlp_backward_packed_bytes(const dn_mlp_desc* desc, int precision) {
  if (precision == DN_PREC_BF16_S8) {   // the 48-point chain's stream
    if (validate_desc(desc, DN_PREC_BF16) || !g48_train_supported(*desc)) return 0;
    NetLayout L48;
    build_backward_layout48(*desc, &L48);
    return static_cast<size_t>(L48.total_pieces) * kPieceBytes;
  }
  if (validate_desc(desc, precision)) return 0;
  NetLayout L;
  build_backward_layout(*desc, precision, &L);
  return static_cast<size_t>(L.total_pieces) * kPieceBytes;
}

extern "C" int dn_mlp_pack_backward(const dn_mlp_desc* desc, int precision, const float* const* h_weights, void* packed,
                                    dn_stream_t stream) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(h_weights && packed, "dn_mlp_pack_backward: NULL pointer");
  const int n_params = desc->num_layers + (desc->use_viewdirs ? 4 : 1);
  PackPtrs ptrs{};
  for (int i = 0; i < n_params; ++i) {
    DN_REQUIRE(h_weights[i], "dn_mlp_pack_backward: parameter %d is NULL", i);
    ptrs.w[i] = h_weights[i];
    ptrs.b[i] = h_weights[i];  // unused (no bias tiles in the backward stream)
  }
  if (s8) {
    DN_REQUIRE(g48_train_supported(*desc), "dn_mlp_pack_backward: no 8-bit-saved-tensor training kernels for this network");
    return launch_pack48_backward(*desc, ptrs, static_cast<char*>(packed), as_stream(stream));
  }
  NetLayout L;
  build_backward_layout(*desc, precision, &L);
  return launch_pack(L, ptrs, packed, precision, as_stream(stream));
}

// Both streams a DN_PREC_BF16_S8 training step reads - the 48-point forward stream (inside `packed`, behind the core one) and the
// transposed backward stream - for TWO networks of one architecture (the coarse and the fine net), in two launches instead of four.
extern "C" int dn_mlp_pack_train_pair(const dn_mlp_desc* desc, const float* const* h_weights_a, const float* const* h_biases_a,
                                      void* packed_a, void* packed_bwd_a, const float* const* h_weights_b,
                                      const float* const* h_biases_b, void* packed_b, void* packed_bwd_b, dn_stream_t stream) {
  int rc = validate_desc(desc, DN_PREC_BF16);
  if (rc) return rc;
  DN_REQUIRE(g48_train_supported(*desc), "dn_mlp_pack_train_pair: no 8-bit-saved-tensor training kernels for this network");
  DN_REQUIRE(h_weights_a && h_biases_a && packed_a && packed_bwd_a && h_weights_b && h_biases_b && packed_b && packed_bwd_b,
             "dn_mlp_pack_train_pair: NULL pointer");
  const int n_params = desc->num_layers + (desc->use_viewdirs ? 4 : 1);
  PackPtrs a{}, b{};
  for (int i = 0; i < n_params; ++i) {
    DN_REQUIRE(h_weights_a[i] && h_biases_a[i] && h_weights_b[i] && h_biases_b[i], "dn_mlp_pack_train_pair: parameter %d is NULL", i);
    a.w[i] = h_weights_a[i]; a.b[i] = h_biases_a[i];
    b.w[i] = h_weights_b[i]; b.b[i] = h_biases_b[i];
  }
  NetLayout L;
  build_layout(*desc, DN_PREC_BF16, &L);   // the 48-point region starts behind the core stream
  const size_t core = static_cast<size_t>(L.bias_bytes) + static_cast<size_t>(L.total_pieces) * kPieceBytes;
  if ((rc = launch_pack48_pair(*desc, a, b, static_cast<char*>(packed_a) + core, static_cast<char*>(packed_b) + core, as_stream(stream)))) return rc;
  return launch_pack48_backward_pair(*desc, a, b, static_cast<char*>(packed_bwd_a), static_cast<char*>(packed_bwd_b), as_stream(stream));
}

extern "C" int dn_run_network_train(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts,
                                    const float* viewdirs, const float* rays, int ray_stride, const float* z_vals,
                                    int64_t n_rays, int samples_per_ray, float* out, void* act, void* masks,
                                    dn_stream_t stream) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;
  FwdParams p;
  int rc = setup_params(desc, precision, packed, &p);
  if (rc) return rc;
  DN_REQUIRE(packed && out && act && masks && n_rays >= 0 && samples_per_ray >= 1, "dn_run_network_train: bad arguments");
  DN_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(masks)) & 15) == 0,
             "dn_run_network_train: buffers must be 16-byte aligned");
  if (pts != nullptr) {
    DN_REQUIRE(!desc->use_viewdirs || viewdirs, "dn_run_network_train: viewdirs required with use_viewdirs");
    p.mode = 1; p.pts = pts; p.viewdirs = viewdirs;
  } else {
    DN_REQUIRE(rays && z_vals, "dn_run_network_train: need pts, or rays + z_vals");
    DN_REQUIRE(ray_stride >= (desc->use_viewdirs ? 11 : 8), "dn_run_network_train: ray_stride too small");
    p.mode = 0; p.rays = rays; p.ray_stride = ray_stride; p.z = z_vals;
  }
  p.n_points = n_rays * samples_per_ray;
  p.S = samples_per_ray;
  p.out = out;
  p.act = static_cast<char*>(act);
  p.masks = static_cast<char*>(masks);
  if (s8) {   // the 48-point training forward: slots / strides in units of the s8-48 layout
    DN_REQUIRE(g48_train_supported(*desc), "dn_run_network_train: no 8-bit-saved-tensor training kernels for this network (see dn_mlp_train_sizes)");
    TrainLayout48 t8;
    build_train_layout48(*desc, &t8);
    p.act_pieces = t8.act_units; p.mask_words = t8.mask_stages;
    p.slot_xyz = t8.slot_xyz; p.slot_dir = t8.slot_dir; p.slot_layer1 = t8.slot_layer1; p.slot_trunk0 = t8.slot_trunk0;
    p.slot_feat = t8.slot_feat; p.slot_dirout = t8.slot_dirout;
    p.save8 = 1;
  } else {
    TrainLayout t;
    build_train_layout(*desc, precision, &t);
    p.act_pieces = t.act_pieces; p.mask_words = t.mask_words;
    p.slot_xyz = t.slot_xyz; p.slot_dir = t.slot_dir; p.slot_layer1 = t.slot_layer1; p.slot_trunk0 = t.slot_trunk0;
    p.slot_feat = t.slot_feat; p.slot_dirout = t.slot_dirout;
  }
  if (p.n_points == 0) return 0;
  return dispatch_forward(*desc, precision, p, as_stream(stream));
}

extern "C" int dn_mlp_backward_data(const dn_mlp_desc* desc, int precision, const void* packed_bwd, const float* g_out,
                                    const void* masks, int64_t n_points, void* grads, dn_stream_t stream) {
  return dn::mlp_backward_data_partials(desc, precision, packed_bwd, g_out, masks, n_points, grads, nullptr, 0, stream);
}

bool dn::s8_scale_is_per_launch() { return g_s8_grad_scale.load() == 0.0f; }

int dn::mlp_backward_data_partials(const dn_mlp_desc* desc, int precision, const void* packed_bwd, const float* g_out, const void* masks,
                                   int64_t n_points, void* grads, const unsigned* partials, int n_partials, dn_stream_t stream) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(precision != DN_PREC_F16, "dn_mlp_backward_data: fp16 is a render-only mode");
  DN_REQUIRE(packed_bwd && g_out && masks && grads && n_points >= 0, "dn_mlp_backward_data: bad arguments");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(g_out) & 15) == 0, "dn_mlp_backward_data: g_out must be 16-byte aligned");
  if (n_points == 0) return 0;
  if (s8) {   // the 48-point chain (mlp_train48.hip)
    DN_REQUIRE(g48_train_supported(*desc), "dn_mlp_backward_data: no 8-bit-saved-tensor training kernels for this network (see dn_mlp_train_sizes)");
    DN_REQUIRE(n_points < (1LL << 31) - 1024, "dn_mlp_backward_data (8-bit saved tensors): at most 2^31 - 1024 points per call");
    return backward48_entry(desc, packed_bwd, g_out, masks, n_points, grads, g_s8_grad_scale.load(), as_stream(stream), partials, n_partials);
  }
  NetLayout L;
  build_backward_layout(*desc, precision, &L);
  TrainLayout t;
  build_train_layout(*desc, precision, &t);
  BwdParams p{};
  p.packed = static_cast<const char*>(packed_bwd);
  p.total_pieces = L.total_pieces;
  p.D = desc->num_layers;
  p.use_viewdirs = desc->use_viewdirs;
  p.g_out = g_out;
  p.masks = static_cast<const char*>(masks);
  p.mask_words = t.mask_words;
  p.n_points = n_points;
  p.grads = static_cast<char*>(grads);
  p.grad_pieces = t.grad_pieces;
  p.gslot_dirout = t.gslot_dirout; p.gslot_feat = t.gslot_feat; p.gslot_trunk0 = t.gslot_trunk0; p.gslot_layer1 = t.gslot_layer1;
  p.gslot_out = t.gslot_out;
  const bool bf = precision == DN_PREC_BF16;
  if (desc->hidden_size == 256) return bf ? launch_backward<256, true>(p, as_stream(stream)) : launch_backward<256, false>(p, as_stream(stream));
  if (desc->hidden_size == 128) return bf ? launch_backward<128, true>(p, as_stream(stream)) : launch_backward<128, false>(p, as_stream(stream));
  set_error("dn_mlp_backward_data: no kernel instance for W=%d", desc->hidden_size);
  return DN_E_UNSUPPORTED;
}

extern "C" int dn_mlp_unpack(const dn_mlp_desc* desc, int precision, int which, const void* native, int64_t n_points,
                             int slot, int width, int kind, float* out, int ld_out, int col0, dn_stream_t stream) {
  if (precision == DN_PREC_BF16_S8) {   // the 8-bit units of the 48-point training kernels (kind 3: the custom output-gradient unit)
    int rc8 = validate_desc(desc, DN_PREC_BF16);
    if (rc8) return rc8;
    DN_REQUIRE(native && out && n_points >= 0 && width > 0 && ld_out >= col0 + 1 && (which == 0 || which == 1) && kind >= 0 && kind <= 3 &&
               g48_train_supported(*desc), "dn_mlp_unpack (8-bit layout): bad arguments");
    if (n_points == 0) return 0;
    return unpack48_entry(desc, which, native, n_points, slot, width, kind, out, ld_out, col0, as_stream(stream));
  }
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(native && out && n_points >= 0 && width > 0 && ld_out >= col0 + 1 && (which == 0 || which == 1) && kind >= 0 && kind <= 2,
             "dn_mlp_unpack: bad arguments");
  if (n_points == 0) return 0;
  TrainLayout t;
  build_train_layout(*desc, precision, &t);
  const int per_tile = which == 0 ? t.act_pieces : t.grad_pieces;
  const int n_pieces = (kind == 0) ? width / t.kpp : (kind == 1 ? t.kxp : t.kdp);
  const int L = kind == 1 ? desc->num_encoding_fn_xyz : desc->num_encoding_fn_dir;
  DN_REQUIRE(slot >= 0 && slot + n_pieces <= per_tile, "dn_mlp_unpack: slot range outside the tile");
  if (precision == DN_PREC_BF16)
    hipLaunchKernelGGL(unpack_kernel<true>, dim3(2048), dim3(256), 0, as_stream(stream), static_cast<const char*>(native),
                       per_tile, slot, n_pieces, kind, L, n_points, out, ld_out, col0);
  else
    hipLaunchKernelGGL(unpack_kernel<false>, dim3(2048), dim3(256), 0, as_stream(stream), static_cast<const char*>(native),
                       per_tile, slot, n_pieces, kind, L, n_points, out, ld_out, col0);
  return check_launch("dn_mlp_unpack");
}

// ==============================================================================================================
// Weight / bias gradients straight from the wave-native buffers (bf16):  dW[n][k] += sum_p dY[p][n] X[p][k].
// The contraction runs over POINTS, which both saved tensors keep on the lane axis, so each 1 KiB native piece
// (64 lanes x 8 features) is staged in LDS as is (LDS-DMA, lane-linear) and read back TRANSPOSED with
// ds_read_b64_tr_b16: a 16-lane group fetches 4 points x 16 features and every lane receives one feature of those
// 4 points - two reads build the 8-point MFMA fragment of one feature row.  One workgroup (8 waves, one per CU:
// the accumulators take the register file) owns the whole (N x K) gradient of a layer for a strided set of
// 32-point tiles and adds its partial with fp32 atomics at the end.  An all-ones B tile yields db.
// The layer shape is a template parameter: the tile loop is straight-line code (a runtime-shaped version spent
// 5x the MFMA time in scalar branches), and the LDS not needed for two tile buffers is used for depth: up to 16
// tile buffers, all but one in flight, with counted vmcnt waits.
// HBM-bound: (N + K) x 2 B per point against 2 N K FLOP per point (146 FLOP/B at 256 x 320).
// ==============================================================================================================
#ifndef DN_WG_EPI
#define DN_WG_EPI 0
#endif
// cache policy of the weight-gradient kernel's streaming loads: non-temporal (read-once data; -DDN_WG_LOAD_POLICY_ID=0
// plain / 2 sc1 are ablation hooks: 2.11 / 1.98 vs 1.92 ms for all layers at 786 k points)
#if !defined(DN_WG_LOAD_POLICY_ID) || DN_WG_LOAD_POLICY_ID == 1
#define DN_WG_LOAD_POLICY " nt"
#elif DN_WG_LOAD_POLICY_ID == 0
#define DN_WG_LOAD_POLICY ""
#else
#define DN_WG_LOAD_POLICY " sc1"
#endif
namespace dn {

struct WgParams {
  const char* act;
  const char* grads;
  int act_pieces, grad_pieces;
  long long n_points;
  int g_slot;                     // dY pieces: g_slot .. (2 per 32-row tile, or ONE custom piece)
  int custom_rows;                // custom dY piece: element (half h, e) = output row 8h+e, rows < custom_rows real
  int x_slot;                     // hidden X pieces (2 per 32-feature tile)
  int pe_slot, pe_L;              // positional-encoding pieces appended to X
  int pe_kind;                    // 1 xyz, 2 view direction (the 8-bit layout keeps the two panels differently: mlp_geo48.h)
  float* dW;
  int ldw, col_pe0;
  float* db;
  int shape;                      // index into the instantiation table (wg_shape_index)
  // deterministic reduction (dn_*_ws entry points): workgroup `wg` of the unit stores its partial - [n_real x ldw] dW then [n_real]
  // db - at part + wg * part_stride with plain stores, and wg_reduce_kernel adds the partials in workgroup order (NULL: fp32 atomics
  // straight into dW / db, whose order - and so the sum's last bits - changes from launch to launch)
  float* part;
  int part_stride;                // floats between consecutive workgroups' partials
  int n_real;                     // rows of dW / db that exist (custom dY: custom_rows, else the layer's output width)
  // 8-bit saved tensors (DN_PREC_BF16_S8): slots / strides above count 1 KiB UNITS (= two 8-byte-per-lane pieces side by
  // side); dy_odd = which half of its unit a custom dY piece is; out_scale = 1 / (the power of two the gradients were
  // multiplied by before they were rounded to e5m2)
  int dy_odd;
  const unsigned* scale_word;     // bits of the scale the backward-data launch recorded behind the saved gradients (kS8BlockScale)
  unsigned* stats_block;          // that record: the 8-bit kernel counts saturated / floor-level gradient bytes into it (mlp_geo48.h)
#ifdef DN_WG_STAMP
  unsigned long long* stamp;      // diagnostic build: [workgroup][wave][8] accumulated s_memtime ticks
#endif
};

constexpr int kWgLdsBytes = 144 * 1024;
// bf16 weight-gradient kernel: pieces are staged 1088 bytes apart (64 B more than their size), so that the two pieces of a pair sit on
// different halves of the 32 LDS banks (see weight_grad_unit), and its tile buffers may use 158 KiB
constexpr int kWgPieceStride16 = kPieceBytes + 64;
constexpr int kWgLdsBytes16 = 158 * 1024;

// Tiles consumed per barrier: a small tile (few pieces) is a few MFMAs per wave, so the barrier + wait + LDS latency of
// an iteration is amortised over 2 or 4 of them (the as-shipped 4x128 nets are all "small").
constexpr int wg_tiles_per_iter(int pieces) { return pieces <= 12 ? 4 : (pieces <= 24 ? 2 : 1); }

// the small layer shapes of the 8-bit kernel (the W = 128 nets') also exist on HALF the LDS, two workgroups per CU: their tile loop is
// bound by the wait / barrier / LDS round trip of a tile, which a second resident workgroup hides (weight_grad_batch_kernel_s8_small)
constexpr int kWgLdsBytesSmall = 72 * 1024;
constexpr int wg_stages_for(int pieces, int tpi, bool s8, int lds_s8 = kWgLdsBytes) {
  int s = s8 ? lds_s8 / (pieces * kPieceBytes) : kWgLdsBytes16 / (pieces * kWgPieceStride16);
  if (s > 16) s = 16;
  const int per_wave = (pieces + 7) / 8;
  while (s > 2 * tpi && (s - 2 * tpi) * per_wave > 48) --s;  // counted-wait range
  return s < 2 * tpi ? 2 * tpi : s;
}

// NTN: 32-row tiles of the output (dY) width; XT: 32-column tiles of the hidden input; PET: 32-column tiles of the
// appended positional encoding; CUSTOM: dY is the single custom output-gradient piece (fc_rgb / fc_alpha / fc_out)
template <int NTN_, int XT_, int PET_, bool CUSTOM_, bool S8_ = false, int LDSB_ = kWgLdsBytes>
struct WgShape {
  static constexpr int LDS_BYTES = LDSB_;                   // (8-bit kernel) dynamic LDS of the launch this shape is compiled for
  static constexpr int NTN = NTN_, XT = XT_, PET = PET_;
  static constexpr bool CUSTOM = CUSTOM_;
  static constexpr bool S8 = S8_;                          // 8-bit saved tensors: a staged 1 KiB unit holds BOTH pieces of a 32-feature tile
  static constexpr int KT = XT + PET + 1;                  // k-tiles incl. the all-ones (bias) tile
  static constexpr int KGROUPS = 8 / NTN;                  // waves sharing one n-tile split the k-tiles
  static constexpr int J = (KT + KGROUPS - 1) / KGROUPS;   // k-tiles (accumulators) per wave
  static constexpr int UPT = S8 ? 1 : 2;                   // staged 1 KiB units per 32-feature tile
  // 8-bit buffers (s8-48 layout, mlp_geo48.h): a 32-point record holds, per 64-feature slot, the unit of its first and of its
  // second 16-point group side by side - so units come in pairs: a 32-feature tile is half of the two units of its slot, and a
  // lone custom dY piece / the one-piece view-direction panel still stage a whole pair
  static constexpr int N_DY = CUSTOM ? (S8 ? 2 : 1) : UPT * NTN;
  static constexpr int N_X = UPT * XT, N_PE = S8 ? 2 * ((PET + 1) / 2) : UPT * PET;
  static_assert(!S8 || (XT % 2 == 0 && (CUSTOM || NTN % 2 == 0)), "8-bit layout: whole 64-feature slots");
  static constexpr int PIECES = N_DY + N_X + N_PE;         // 1 KiB pieces staged per 32-point tile
  static constexpr int PER_WAVE = (PIECES + 7) / 8;        // DMAs per tile of the busiest wave
  // 8-bit kernel: PAIRS of tiles contracted by the K = 64 fp8 MFMA (twice the K = 16 rate) wherever 24 operand registers fit
  // beside the accumulators (J <= 9: all but the skip layer of a W = 256 net, which keeps the K = 16 form)
  static constexpr bool K64 = S8 && J <= 11;
  // tiles per barrier: by the MFMA count of a tile, not by its bytes
  static constexpr int TPI0 = wg_tiles_per_iter(S8 ? 2 * PIECES : PIECES);
  static constexpr int TPI = (K64 && TPI0 < 2) ? 2 : TPI0;
  static constexpr int STAGES = wg_stages_for(PIECES, TPI, S8, LDSB_); // tile buffers in LDS; STAGES - TPI tiles in flight
  static constexpr int PSTRIDE = S8 ? kPieceBytes : kWgPieceStride16;   // LDS distance of consecutive staged pieces
  // Cycles one 32-point tile costs a workgroup of this shape in the 8-bit kernel, fitted to -DDN_WG_STAMP runs (profiles/r02_train_s8.md;
  // measured / model for the W = 256 shapes: (8,0,2) 678 / 746, (8,8,0) 1131 / 1134, (4,8,1) 940 / 1000, (1,8,0) 868 / 904, (8,8,2)
  // 2123 / 2138): wait + barrier, ~100 issue cycles per LDS-DMA of the busiest wave, and the MFMAs of the two waves of a SIMD - K = 64:
  // 64 cycles per accumulator tile and PAIR of tiles, but never less than the LDS round trips of a tile (~260); K = 16: 128 per tile.
  // The 8-bit launch divides its workgroups among the layers by this, not by bytes: it is not HBM-bound.
  static constexpr int COST = K64 ? 250 + 12 * J + 100 * PER_WAVE + (64 * J > 260 ? 64 * J : 260) + (KGROUPS > 1 ? 170 : 0) - (TPI >= 4 ? 100 : 0)
                                  : 430 / TPI + 100 * PER_WAVE + 128 * J;
  static_assert(STAGES >= 2 * TPI && STAGES * PIECES * PSTRIDE <= 160 * 1024, "LDS budget");
  static_assert(XT + PET >= 1 && NTN * KGROUPS == 8, "shape");
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const char* piece_lane /* piece base + this lane's transposing offset */,
                                          int point0 /* first of the 8 points, multiple of 4 */) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(piece_lane + point0 * 16));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(piece_lane + (point0 + 4) * 16));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// 8-bit form (ds_read_b64_tr_b8; lane / byte mapping measured: profiles/r02_tr8_probe.md): a 16-lane group reads an
// 8-row x 16-byte block - lane li supplies the address of the 8-byte chunk (row li >> 1, half li & 1) - and receives byte
// column li, 8 rows.  Rows = points, byte columns = the 16 features of one piece (half h = the native piece's lane half):
// ONE read is the 8-point operand block of v_mfma_f32_32x32x16_bf8_fp8 (the 16-bit form needs two).
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long tr8_frag(const char* unit_lane /* unit base + odd * 8 + this lane's chunk offset */, int point0) {
  const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(unit_lane + point0 * 16));
  return __builtin_bit_cast(long, v);
}

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14])
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// One unit = one nn.Linear's (dW, db); workgroup `wg` of the `n_wg` that share the unit takes 32-point tiles
// wg, wg + n_wg, ...
template <class S>
__device__ __forceinline__ void weight_grad_unit(const WgParams& p, int wg, int n_wg, char* smem) {
  const long long tiles = (p.n_points + 31) / 32;
  if (wg >= tiles) return;  // nothing to add (workgroup-uniform)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int BUF = S::PIECES * S::PSTRIDE;
  constexpr int PS = S::PSTRIDE;

  // wave -> (n-tile, subset of k-tiles)
  const int ntile = wave % S::NTN;
  const int kgroup = S::KGROUPS == 1 ? 0 : wave / S::NTN;   // (a constant where every wave has its own n-tile: the epilogue then knows which k-tiles a wave holds)

  // transposing read: lane (16-lane group g, li) supplies row (li>>2) / column chunk (li&3) of a 4 x 16 block and
  // receives feature column li; group g covers feature sub-block fs = g&1 and k-half hh = g>>1 of the MFMA operand
  const int li = lane & 15, grp = lane >> 4;
  const int fs = grp & 1, hh = grp >> 1;
  // (8-bit units: a group reads the 8 x 16-byte rows [8 bytes of piece 2u | 8 bytes of piece 2u+1] that the lanes (point, lane
  // half fs) stored - 128 CONTIGUOUS bytes, every LDS bank once.  Reading the two lane halves of one piece side by side, as the
  // bf16 form has to, puts a group's chunk pairs 512 bytes apart = on the same bank: PMC showed one conflict cycle per two
  // LDS cycles.  The price is a permuted feature order inside the 32-wide tile: s8_feature below.)
  // (the 8-bit kernel's MFMA contracts 64 points = a PAIR of tiles: hh selects the tile, the four reads of a lane its 32 points)
  // bf16 pieces: a 16-lane group reads 4 rows x 32 bytes = [16 bytes of piece 2t | 16 bytes of piece 2t+1] of the lanes (point, lane
  // half fs) - the same pairing as the 8-bit units - and pieces are staged 1088 bytes apart, so the two 16-byte halves of a row sit 64
  // bytes apart modulo the 128-byte bank period: a group covers all 32 banks once.  (Reading the two lane halves of ONE piece side by
  // side put them 512 B apart = on the same 16 banks: stamps showed 2,242 cycles of `consume` per 256 x 256 tile for 1,152 cycles
  // of MFMAs, PMC three conflict cycles in four LDS cycles.)
  // 8-bit units (s8-48 layout): unit rows are (lane group g, point j of the 16-point group), 16 bytes each = the 8 + 8 bytes of
  // the slot's two pieces; a 32-feature fragment of parity par takes rows g = 2 par + fs, a 16-lane group reads 8 of them (128
  // contiguous bytes, every LDS bank once): lane li supplies the 8-byte chunk (row li >> 1, half li & 1) and receives byte
  // column li.  Parity 1 sits 512 bytes further on; the record's second group one unit (PS) further on.
  // The odd lane groups' rows are stored swizzled (point j at row j ^ 8: mlp_device.h store16_unit48): this lane's fs = 1 reads start
  // 128 bytes in and step back, so the two 16-lane groups a 32-lane LDS cycle serves touch different halves of the 64 banks.
  const int lane_off = S::S8 ? ((fs * 16 + (li >> 1)) * 16) + (li & 1) * 8 + fs * 128
                             : ((li & 3) >> 1) * PS + ((fs * 32 + (li >> 2) + 8 * hh) * 16) + ((li & 3) & 1) * 8;
  const int swz = 128 - 256 * fs;   // (8-bit kernel) byte step from a group's first eight points to its second eight

  // ---- staging: each 1 KiB piece is one LDS-DMA (opaque asm: the counted waits below are ours; hipcc would drain
  // with vmcnt(0) at every barrier).  This wave stages pieces wave, wave + 8, ... of every tile.
  const unsigned lane16 = lane * 16;
  const unsigned smem_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const char* src0[S::PER_WAVE];
  long long stride[S::PER_WAVE];
  static_for<S::PER_WAVE>([&](auto e_c) {
    constexpr int e = decltype(e_c)::value;
    const int pi = wave + 8 * e;
    if (pi < S::N_DY) { src0[e] = p.grads + static_cast<long long>(p.g_slot + pi) * kPieceBytes; stride[e] = static_cast<long long>(p.grad_pieces) * kPieceBytes; }
    else if (pi < S::N_DY + S::N_X) { src0[e] = p.act + static_cast<long long>(p.x_slot + pi - S::N_DY) * kPieceBytes; stride[e] = static_cast<long long>(p.act_pieces) * kPieceBytes; }
    else { src0[e] = p.act + static_cast<long long>(p.pe_slot + pi - S::N_DY - S::N_X) * kPieceBytes; stride[e] = static_cast<long long>(p.act_pieces) * kPieceBytes; }
  });
  auto stage = [&](long long tile32, int buf) {
    if (tile32 >= tiles) tile32 = tiles - 1;  // past the end: harmless re-load, keeps the DMA count per stage constant
    static_for<S::PER_WAVE>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value;
      const unsigned long long src_bits = reinterpret_cast<unsigned long long>(src0[e] + tile32 * stride[e]);
      const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits));
      const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits >> 32));
      const char* usrc = reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
      const unsigned lds = __builtin_amdgcn_readfirstlane(smem_addr + buf * BUF + (wave + 8 * e) * PS);
      const unsigned go = __builtin_amdgcn_readfirstlane((wave + 8 * e < S::PIECES) ? 1u : 0u);
      const unsigned voff = lane16;  // (asm operands do not capture: name a local)
      unsigned keep;
#ifdef DN_WG_NOSTAGE
      asm volatile("" : [keep] "=s"(keep) : [go] "s"(go), [lds] "s"(lds), [voff] "v"(voff), [sbase] "s"(usrc) : "memory");
#else
      asm volatile(
          "s_cmp_lg_u32 %[go], 0\n\t"
          "s_cbranch_scc0 .Ldn_wg_skip%=\n\t"
          "s_mov_b32 %[keep], m0\n\t"
          "s_mov_b32 m0, %[lds]\n\t"
          "s_nop 1\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase]" DN_WG_LOAD_POLICY "\n\t"
          "s_mov_b32 m0, %[keep]\n"
          ".Ldn_wg_skip%=:"
          : [keep] "=&s"(keep)
          : [go] "s"(go), [lds] "s"(lds), [voff] "v"(voff), [sbase] "s"(usrc)
          : "memory", "scc");
#endif
    });
  };
  // this wave issues PER_WAVE or PER_WAVE-1 DMAs per tile; while TPI tiles are consumed, STAGES - 2*TPI younger tiles may
  // stay in flight (in-order retirement)
  const bool full = (wave + 8 * (S::PER_WAVE - 1)) < S::PIECES;
  auto wait_tiles = [&]() {
    if (full) wait_vmcnt<(S::STAGES - 2 * S::TPI) * S::PER_WAVE>();
    else wait_vmcnt<(S::STAGES - 2 * S::TPI) * (S::PER_WAVE - 1)>();
  };

  f32x16 acc[S::J];
#pragma unroll
  for (int k = 0; k < S::J; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  unsigned s8_stats[3] = {0u, 0u, 0u};   // (8-bit kernel) saturated / floor-level / non-zero gradient bytes this lane counted
  bf16x8 ones, zeros;
#pragma unroll
  for (int e = 0; e < 8; ++e) { ones[e] = static_cast<__bf16>(1.0f); zeros[e] = static_cast<__bf16>(0.0f); }

  // one staged tile: A = dY^T fragments of this wave's n-tile, B = the X / PE / all-ones k-tiles, 2 MFMAs per k-tile
  auto consume_16 = [&](long long tile, int cb) {
    const char* base = smem + cb * BUF + lane_off;
    // A = dY^T fragments of this wave's n-tile, two 16-point k-steps (fragment row 16 fs + li = piece li >> 3 of the pair, lane half fs,
    // element li & 7).  A custom dY is a single piece: the columns read from its neighbour are zeroed.
    const char* dy = base + (S::CUSTOM ? 0 : 2 * ntile) * PS;
    bf16x8 a0 = tr_frag(dy, 0);
    bf16x8 a1 = tr_frag(dy, 16);
    if constexpr (S::CUSTOM) {
      a0 = (li >> 3) ? zeros : a0;
      a1 = (li >> 3) ? zeros : a1;
    }
    const long long valid = p.n_points - tile * 32;  // points of this tile that exist (the rest are padding copies)
    if (valid < 32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (8 * hh + e >= valid) a0[e] = static_cast<__bf16>(0.0f);
        if (16 + 8 * hh + e >= valid) a1[e] = static_cast<__bf16>(0.0f);
      }
    }
    static_for<S::J>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      bf16x8 b0, b1;
      if constexpr (S::KGROUPS == 1) {
        if constexpr (j == S::KT - 1) { b0 = ones; b1 = ones; }
        else {
#ifndef DN_WG_NOREAD
          const char* pb = base + (S::N_DY + 2 * j) * PS;
          b0 = tr_frag(pb, 0);
          b1 = tr_frag(pb, 16);
#else
          b0 = ones; b1 = ones;
#endif
        }
      } else {
        const int kt = kgroup + j * S::KGROUPS;           // wave-uniform; kt >= KT: an unused accumulator
        const int ktr = kt < S::KT - 1 ? kt : S::KT - 2;  // a piece that exists
        const char* pb = base + (S::N_DY + 2 * ktr) * PS;
        b0 = tr_frag(pb, 0);
        b1 = tr_frag(pb, 16);
        const bool is_ones = kt >= S::KT - 1;
        b0 = is_ones ? ones : b0;
        b1 = is_ones ? ones : b1;
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
    });
  };
  // 8-bit buffers: tiles `tile0` (buffer cb0) and `tile1` (buffer cb1; tile1 >= tiles: absent) as ONE K = 64 contraction with
  // v_mfma_f32_32x32x64_f8f6f4 (A = e5m2, B = e4m3, no scaling: twice the rate of the K = 16 fp8 / bf16 MFMAs).  Lane (row or
  // column 16 fs + li, K block hh) holds 32 consecutive K slots = the 32 points of tile hh, four 8-point reads; A and B use the
  // same slot order, which is all a contraction needs.
  auto consume_pair_s8 = [&](long long tile0, int cb0, long long tile1, int cb1) {
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    const char* base = smem + (hh ? cb1 : cb0) * BUF + lane_off;
    constexpr int kOnes = 0x38383838;   // 1.0 in e4m3, four times
    // this lane's 32 K slots of one operand = the 32 points of its record: four 8-point transposing reads, two in the unit of
    // the record's first 16-point group, two in the second group's (the next staged unit)
    auto read32 = [swz](const char* unit_lane) {
      i32x8 o;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        // (f & 1: the second eight points of the group - 128 bytes further on for an even lane group, 128 bytes BACK for an odd one,
        // whose rows are stored swizzled: `swz` below)
        const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(unit_lane + (f >> 1) * kPieceBytes + (f & 1) * swz));
        o[2 * f] = v[0]; o[2 * f + 1] = v[1];
      }
      return o;
    };
    // 32-feature fragment `frag` of a staged vector that starts at unit `first`: slot frag / 2 (two staged units), row parity frag % 2
    auto frag_at = [&](int first, int frag) { return base + (first + 2 * (frag >> 1)) * kPieceBytes + (frag & 1) * 512; };
    // A = dY^T (fragment row 16 fs + li = lane group 2 (ntile % 2) + fs, byte li of the unit row: piece li >> 3, element li & 7)
    i32x8 av = read32(frag_at(0, S::CUSTOM ? 0 : ntile));
    const long long mine = hh ? tile1 : tile0;
    // statistics of the saved gradients (mlp_geo48.h): every dY byte of a record is in exactly one lane of the waves with k-group
    // 0; one record in sixteen is counted - byte-parallel compares on the operand registers, this kernel has the issue slots
    if (kgroup == 0 && (mine & 15) == 0 && mine < tiles) {
      auto zero_bytes = [](unsigned x) { return __builtin_popcount(((x - 0x01010101u) & ~x & 0x80808080u)); };   // (exact for our inputs: no byte borrows past a zero byte)
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const unsigned mag = static_cast<unsigned>(av[d]) & 0x7F7F7F7Fu;
        s8_stats[0] += static_cast<unsigned>(zero_bytes(mag ^ 0x7B7B7B7Bu));
        s8_stats[1] += static_cast<unsigned>(zero_bytes(mag ^ 0x01010101u));
        s8_stats[2] += 4u - static_cast<unsigned>(zero_bytes(mag));
      }
    }
    // points of this lane's tile that exist (the rest: padding copies, or a re-load standing in for an absent tile); a custom dY
    // is piece dy_odd of its unit: the columns of the other piece are not its rows
    int valid = mine < tiles ? static_cast<int>(p.n_points - mine * 32 < 32 ? p.n_points - mine * 32 : 32) : 0;
    if constexpr (S::CUSTOM) valid = ((li >> 3) != p.dy_odd) ? 0 : valid;
    if (valid < 32) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int n = valid - 4 * d;   // K slots 4d .. 4d+3 (one per byte)
        av[d] &= n >= 4 ? -1 : (n <= 0 ? 0 : static_cast<int>((1u << (8 * n)) - 1u));
      }
    }
    auto load_b = [&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      i32x8 bv;
      if constexpr (S::KGROUPS == 1) {
        if constexpr (j == S::KT - 1) {
#pragma unroll
          for (int d = 0; d < 8; ++d) bv[d] = kOnes;
        } else {
#ifndef DN_WG_NOREAD
          bv = read32(frag_at(j < S::XT ? S::N_DY : S::N_DY + S::N_X, j < S::XT ? j : j - S::XT));
#else
#pragma unroll
          for (int d = 0; d < 8; ++d) bv[d] = kOnes;
#endif
        }
      } else {
        const int kt = kgroup + j * S::KGROUPS;           // wave-uniform; kt >= KT: an unused accumulator
        const int ktr = kt < S::KT - 1 ? kt : S::KT - 2;  // a fragment that exists
        bv = read32(frag_at(ktr < S::XT ? S::N_DY : S::N_DY + S::N_X, ktr < S::XT ? ktr : ktr - S::XT));
        const bool is_ones = kt >= S::KT - 1;
#pragma unroll
        for (int d = 0; d < 8; ++d) bv[d] = is_ones ? kOnes : bv[d];
      }
      return bv;
    };
    // two k-tiles' operands ahead of the MFMA that uses them (the LDS round trip of four transposing reads is longer than one
    // 16-pass MFMA), and no further: left alone the scheduler hoists all J x 8 operand registers above the first MFMA
    // (the widest layer, 11 accumulator tiles, has room for ONE operand set ahead)
    if constexpr (S::J > 9) {
      i32x8 bcur = load_b(std::integral_constant<int, 0>{});
      static_for<S::J>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        i32x8 bnext = bcur;
        if constexpr (j + 1 < S::J) bnext = load_b(std::integral_constant<int, (j + 1 < S::J ? j + 1 : 0)>{});
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bcur, acc[j], 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bcur = bnext;
      });
    } else {
      i32x8 bq0 = load_b(std::integral_constant<int, 0>{});
      i32x8 bq1 = bq0;
      if constexpr (S::J > 1) bq1 = load_b(std::integral_constant<int, (S::J > 1 ? 1 : 0)>{});
      static_for<S::J>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        i32x8 bnew = bq1;
        if constexpr (j + 2 < S::J) bnew = load_b(std::integral_constant<int, (j + 2 < S::J ? j + 2 : 0)>{});
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq0, acc[j], 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bq0 = bq1;
        bq1 = bnew;
      });
    }
  };
  static_assert(!S::S8 || S::K64, "every 8-bit shape contracts pairs of tiles (K = 64)");
  auto consume = [&](long long tile, int cb) {
    if constexpr (!S::S8) consume_16(tile, cb);
  };
  // tile k of this workgroup's sequence (k = 0, 1, ...) is 32-point tile wg + k * n_wg and lives in buffer k % STAGES
  int buf = 0;
  if constexpr (S::TPI == 1) {
    // one tile per barrier (the big shapes: their accumulators leave no register to spare - keep this loop minimal)
    long long tile = wg;
#pragma unroll 1
    for (int st = 0; st + 1 < S::STAGES; ++st) stage(tile + static_cast<long long>(st) * n_wg, st);
#ifdef DN_WG_STAMP
    unsigned long long st_wait = 0, st_bar = 0, st_stage = 0, st_cons = 0, st_n = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (; tile < tiles; tile += n_wg) {
#ifdef DN_WG_STAMP
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      // this wave's DMAs of tile `tile` are done, and so are its LDS reads of the previous tile ...
      wait_tiles();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef DN_WG_STAMP
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
      // ... after the barrier everyone's are: the tile is resident and the previous tile's buffer is free
      __builtin_amdgcn_s_barrier();
#ifdef DN_WG_STAMP
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
      {
        int nb = buf + S::STAGES - 1;
        if (nb >= S::STAGES) nb -= S::STAGES;
        stage(tile + static_cast<long long>(S::STAGES - 1) * n_wg, nb);
      }
#ifdef DN_WG_STAMP
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
      consume(tile, buf);
#ifdef DN_WG_STAMP
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      st_wait += t1 - t0; st_bar += t2 - t1; st_stage += t3 - t2; st_cons += t4 - t3; st_n += 1;
#endif
      buf = (buf + 1 == S::STAGES) ? 0 : buf + 1;
    }
#ifdef DN_WG_STAMP
    if (lane == 0 && p.stamp != nullptr) {
      unsigned long long* o = p.stamp + (static_cast<long long>(blockIdx.x) * 8 + wave) * 8;
      o[0] = st_wait; o[1] = st_bar; o[2] = st_stage; o[3] = st_cons; o[4] = st_n; o[5] = __builtin_amdgcn_s_memtime() - st_begin;
      o[6] = static_cast<unsigned long long>(p.shape); o[7] = static_cast<unsigned long long>(S::J);
    }
#endif
  } else if constexpr (S::K64 && S::TPI == 2) {
    // one PAIR of tiles per barrier, K = 64 (the big 8-bit shapes: the same minimal loop as above, two tiles at a time)
    long long tile = wg;
#pragma unroll 1
    for (int st = 0; st + 2 < S::STAGES; ++st) stage(tile + static_cast<long long>(st) * n_wg, st);
#ifdef DN_WG_STAMP
    unsigned long long st_wait = 0, st_bar = 0, st_stage = 0, st_cons = 0, st_n = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (; tile < tiles; tile += 2LL * n_wg) {
#ifdef DN_WG_STAMP
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      wait_tiles();   // this pair landed; STAGES - 4 younger tiles may be in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef DN_WG_STAMP
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef DN_WG_STAMP
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
      int nb = buf + S::STAGES - 2;
      if (nb >= S::STAGES) nb -= S::STAGES;
      stage(tile + static_cast<long long>(S::STAGES - 2) * n_wg, nb);
      nb = (nb + 1 == S::STAGES) ? 0 : nb + 1;
      stage(tile + static_cast<long long>(S::STAGES - 1) * n_wg, nb);
      const int cb1 = (buf + 1 == S::STAGES) ? 0 : buf + 1;
#ifdef DN_WG_STAMP
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
      consume_pair_s8(tile, buf, tile + n_wg, cb1);
#ifdef DN_WG_STAMP
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      st_wait += t1 - t0; st_bar += t2 - t1; st_stage += t3 - t2; st_cons += t4 - t3; st_n += 2;   // (per TILE: two per iteration)
#endif
      buf += 2;
      if (buf >= S::STAGES) buf -= S::STAGES;
    }
#ifdef DN_WG_STAMP
    if (lane == 0 && p.stamp != nullptr) {
      unsigned long long* o = p.stamp + (static_cast<long long>(blockIdx.x) * 8 + wave) * 8;
      o[0] = st_wait; o[1] = st_bar; o[2] = st_stage; o[3] = st_cons; o[4] = st_n; o[5] = __builtin_amdgcn_s_memtime() - st_begin;
      o[6] = static_cast<unsigned long long>(p.shape); o[7] = static_cast<unsigned long long>(S::J);
    }
#endif
  } else {
    // TPI tiles per barrier (small shapes: a tile is a few MFMAs per wave, the barrier + wait + LDS latency dominate)
    auto tile_of = [&](long long k) { return wg + k * n_wg; };
#pragma unroll 1
    for (int st = 0; st < S::STAGES - S::TPI; ++st) stage(tile_of(st), st);
#pragma unroll 1
    for (long long k0 = 0; tile_of(k0) < tiles; k0 += S::TPI) {
      wait_tiles();   // tiles k0 .. k0+TPI-1 landed; STAGES - 2*TPI younger ones may be in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int t = 0; t < S::TPI; ++t) {
        int nb = buf + S::STAGES - S::TPI + t;
        if (nb >= S::STAGES) nb -= S::STAGES;
        stage(tile_of(k0 + S::STAGES - S::TPI + t), nb);
      }
      if constexpr (S::K64) {
        static_assert(S::TPI % 2 == 0, "the K = 64 form consumes pairs of tiles");
        static_for<S::TPI / 2>([&](auto t_c) {
          constexpr int tsub = 2 * decltype(t_c)::value;
          int cb0 = buf + tsub;
          if (cb0 >= S::STAGES) cb0 -= S::STAGES;
          const int cb1 = (cb0 + 1 == S::STAGES) ? 0 : cb0 + 1;
          if (tile_of(k0 + tsub) < tiles) consume_pair_s8(tile_of(k0 + tsub), cb0, tile_of(k0 + tsub + 1), cb1);
        });
      } else {
        static_for<S::TPI>([&](auto t_c) {
          constexpr int tsub = decltype(t_c)::value;
          int cb = buf + tsub;
          if (cb >= S::STAGES) cb -= S::STAGES;
          if (tile_of(k0 + tsub) < tiles) consume(tile_of(k0 + tsub), cb);
        });
      }
      buf += S::TPI;
      if (buf >= S::STAGES) buf -= S::STAGES;
    }
  }
  wait_vmcnt<0>();  // the trailing re-loads
  float out_scale = 1.0f;
  if constexpr (S::S8) {
    out_scale = 1.0f / __uint_as_float(*p.scale_word);
    // statistics: wave totals -> LDS -> three atomics per WORKGROUP into one of the record's replicas (same-address atomics
    // serialise: one per wave on one set of counters cost the step 40-90 us)
    unsigned* wg_stats = reinterpret_cast<unsigned*>(smem + S::LDS_BYTES - 128);
    __syncthreads();   // every wave has left the tile loop: the tile buffers are free
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned tot = static_cast<unsigned>(wave_sum(static_cast<double>(s8_stats[k])));
      if (lane == 0) wg_stats[wave * 4 + k] = tot;
    }
    __syncthreads();
    if (wave == 0 && lane < 3) {
      unsigned tot = 0u;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) tot += wg_stats[w8 * 4 + lane];
      if (tot != 0u) atomicAdd(p.stats_block + kS8BlockStats + 4 * (static_cast<int>(blockIdx.x) % kS8BlockReplicas) + lane, tot);
    }
  }
  if constexpr (S::S8) {   // (in place: the epilogue below only moves values)
#pragma unroll
    for (int k = 0; k < S::J; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] *= out_scale;
  }
  // ---- add this workgroup's partial: D[i][j] sits in lane (j = lane&31, half = lane>>5), register r, i = acc_row(r, half)
  const int jl = lane & 31, half = lane >> 5;
  // fragment row / column i (0..31) = (piece s of the tile's two, lane half h, element e): i = 16 h + 8 s + e (see lane_off)
  auto piece_of = [](int i) { return (i >> 3) & 1; };
  auto half_of = [](int i) { return i >> 4; };
  auto feature_of = [&](int i) { return acc_row(piece_of(i) * 8 + (i & 7), half_of(i)); };
  // 8-bit layout: fragment `frag` of a hidden vector, index i = 16 fs + li -> lane group 2 (frag % 2) + fs, piece
  // 2 (frag / 2) + li / 8, element li % 8 (g48_hidden_col); relative to the vector's first feature
  auto feature48 = [&](int frag, int i) { return g48_hidden_col(2 * (frag >> 1) + piece_of(i), 2 * (frag & 1) + half_of(i), i & 7); };
  // dW goes to memory through LDS: in the accumulator layout a lane's 16 values are 16 rows and its neighbours hold 4-float runs
  // scattered over a 64-column span - as atomics that is a 16-byte request each, and the reduction over the workgroups of a layer
  // was a third of the as-shipped nets' launch (scripts/micro/reduce_probe.hip: 12x between scattered and 256-byte-contiguous
  // float atomics).  A round = the k-tiles that make up to 64 consecutive dW columns - hidden inputs: k-tiles 2r and 2r + 1; the
  // encoding panel: its one or two k-tiles, laid down by dW column.  The waves holding those k-tiles write their 32 x 32 blocks
  // into a [rows][64] image (the tile buffers are free by now), then every wave adds whole rows: one instruction = consecutive
  // floats of one dW row.
  constexpr int IMG_LD = 64 + 4;   // floats per image row (+4: rows 32 banks apart would serialise the column writes)
  constexpr int IMG_ROWS = S::CUSTOM ? 8 : 32 * S::NTN;   // (custom dY: fc_rgb / fc_alpha / fc_out, at most 4 rows)
  static_assert(IMG_ROWS * IMG_LD * 4 <= S::STAGES * BUF, "weight_grad: the reduction image fits the tile buffers");
  float* img = reinterpret_cast<float*>(smem);
  const int n_rows = S::CUSTOM ? p.custom_rows : 32 * S::NTN;
  // image row of accumulator register r (custom dY: rows beyond custom_rows - the other piece of the pair, padding - get -1)
  auto row_of = [&](int r) {
    const int irow = acc_row(r, half);
    if constexpr (S::CUSTOM) {
      // custom piece: row = 8 (lane half / lane group) + e (bf16: piece 0 of the pair read; 8-bit: piece dy_odd of the unit - the
      // other one was zeroed)
      const bool mine = piece_of(irow) == (S::S8 ? p.dy_odd : 0);
      const int n = half_of(irow) * 8 + (irow & 7);
      return (mine && n < p.custom_rows) ? n : -1;
    } else {
      return S::S8 ? feature48(ntile, irow) : 32 * ntile + feature_of(irow);
    }
  };
  // lay the block of k-tile KT_ (held in acc[KT_ / KGROUPS] by the waves of k-group KT_ % KGROUPS) into the image at column `col`
  auto lay = [&](auto kt_c, int col) {
    constexpr int kt = decltype(kt_c)::value;
    constexpr int j = kt / S::KGROUPS;
    if (kgroup != kt % S::KGROUPS || col < 0) return;
    if constexpr (S::CUSTOM) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = row_of(r);
        if (n >= 0) img[n * IMG_LD + col] = acc[j][r];
      }
    } else {
      // row of register r = row of register 0 + a compile-time distance (both layouts place r's bits apart from the lane half's
      // and the n-tile's): one address register, immediate offsets
      float* at = img + row_of(0) * IMG_LD + col;
      static_for<16>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        constexpr int i = acc_row(r, 0);
        constexpr int dn = S::S8 ? g48_hidden_col((i >> 3) & 1, i >> 4, i & 7) - g48_hidden_col(0, 0, 0)
                                 : acc_row(((i >> 3) & 1) * 8 + (i & 7), i >> 4) - acc_row(0, 0);
        at[dn * IMG_LD] = acc[j][r];
      });
    }
  };
  // add the image's first n_cols columns to dW[:, col0 ...]: wave w takes rows w, w + 8, ...
  // (the workgroups of a layer finish together and would walk the rows in step - every one queueing on the same cache line while
  // the memory-side atomic units of the others idle: workgroup `wg` starts wg row-groups further on)
  constexpr int ROW_GROUPS = (IMG_ROWS + 7) / 8;
  const int q0 = wg % ROW_GROUPS;
  float* part_w = p.part != nullptr ? p.part + static_cast<long long>(wg) * p.part_stride : nullptr;
  auto add_rows = [&](int col0, int n_cols) {
#pragma unroll
    for (int q = 0; q < ROW_GROUPS; ++q) {
      const int qq = (q + q0 >= ROW_GROUPS) ? q + q0 - ROW_GROUPS : q + q0;
      const int n = wave + 8 * qq;
      if (n < n_rows && lane < n_cols) {
#if DN_WG_EPI == 1
        if (img[n * IMG_LD + lane] == 1.2345f) p.dW[static_cast<long long>(n) * p.ldw + col0 + lane] = 1.0f;
#else
        if (p.part != nullptr) part_w[static_cast<long long>(n) * p.ldw + col0 + lane] = img[n * IMG_LD + lane];   // 256 contiguous bytes of this workgroup's partial
        else atomicAdd(p.dW + static_cast<long long>(n) * p.ldw + col0 + lane, img[n * IMG_LD + lane]);
#endif
      }
    }
  };
  static_for<S::XT / 2>([&](auto r_c) {
    constexpr int rr = decltype(r_c)::value;
    __syncthreads();   // the tile buffers / the previous round's image are done with
    static_for<2>([&](auto par_c) {
      constexpr int kt = 2 * rr + decltype(par_c)::value;
      lay(std::integral_constant<int, kt>{}, (S::S8 ? feature48(kt, jl) : 32 * kt + feature_of(jl)) - 64 * rr);
    });
    __syncthreads();
    add_rows(64 * rr, 64);
  });
  if constexpr (S::PET > 0) {
    __syncthreads();
    static_for<S::PET>([&](auto t_c) {
      constexpr int kt = S::XT + decltype(t_c)::value;
      int pc;   // dW column (relative to the panel's first) of this lane's slot, -1: padding
      if constexpr (S::S8) {
        // xyz panel (two fragments): lane group 2 (fragment) + fs, slot = byte li; view-direction panel (one fragment): lanes of
        // groups 0 / 1 hold [their 8 slots | those of groups 2 / 3]
        const int li_c = jl & 15, fs_c = jl >> 4;
        pc = p.pe_kind == 1 ? g48_pe_col(1, 2 * (kt - S::XT) + fs_c, li_c, p.pe_L) : g48_pe_col(2, fs_c + 2 * (li_c >> 3), li_c & 7, p.pe_L);
      } else {
        const int pe_piece = 2 * (kt - S::XT) + piece_of(jl);
        pc = pe_slot_col(p.pe_L, half_of(jl), pe_piece * 8 + (jl & 7));
      }
      lay(std::integral_constant<int, kt>{}, pc);
    });
    __syncthreads();
    add_rows(p.col_pe0, 3 + 6 * p.pe_L);
  }
  // the all-ones k-tile: its column 0 is the bias gradient - one value per row, added 64 rows per instruction (from the registers
  // it was two lanes per instruction, every workgroup of the layer on the same four cache lines: same-line atomics serialise)
  if (p.db != nullptr) {
    __syncthreads();
    lay(std::integral_constant<int, S::KT - 1>{}, jl == 0 ? 0 : -1);
    __syncthreads();
    const int n = 64 * wave + lane;
    if (n < n_rows) {
#if DN_WG_EPI != 1
      if (p.part != nullptr) part_w[static_cast<long long>(p.n_real) * p.ldw + n] = img[n * IMG_LD];
      else atomicAdd(p.db + n, img[n * IMG_LD]);
#endif
    }
  }
}

// ---- exact-fp32 variant (the parity mode) ---------------------------------------------------------------------
// Same decomposition on the fp32 buffers (4 pieces of 64 lanes x 4 floats per 32-feature tile) with
// v_mfma_f32_32x32x2_f32: every MFMA contracts TWO points, A lane (i, kk) = dY[point 2m+kk][feature i], B lane (j, kk) =
// X[point 2m+kk][feature j], both fetched from the staged pieces with ds_read_b32 (gfx950's transposing reads stop at 16
// bits).  Feature f of a tile sits in piece f/8, lane half (f%8)/4, element f%4 (acc_row), so the operand rows / columns
// are the natural feature order.  Pieces are staged 1056 bytes apart: the 32-byte skew spreads the four pieces of a
// tile over the LDS banks (bank = 8*piece + 4*kk + element: 32 distinct, 2-way on the lane half only).
// MFMA-bound at the fp32 matrix rate (16 MFMAs of 64 cycles per tile pair per 32 points), two tile buffers in LDS.
constexpr int kWg32PieceStride = kPieceBytes + 32;

template <int NTN_, int XT_, int PET_, bool CUSTOM_>
struct WgShape32 {
  static constexpr int NTN = NTN_, XT = XT_, PET = PET_;
  static constexpr bool CUSTOM = CUSTOM_;
  static constexpr int KT = XT + PET + 1;
  static constexpr int KGROUPS = 8 / NTN;
  static constexpr int J = (KT + KGROUPS - 1) / KGROUPS;
  static constexpr int N_DY = CUSTOM ? 1 : 4 * NTN;
  static constexpr int N_X = 4 * XT, N_PE = 4 * PET;
  static constexpr int PIECES = N_DY + N_X + N_PE;
  static constexpr int PER_WAVE = (PIECES + 7) / 8;
  static constexpr int BUF = PIECES * kWg32PieceStride;
  static constexpr int STAGES = (kWgLdsBytes + 8 * 1024) / BUF >= 3 ? 3 : 2;   // 152 KiB budget; deeper does not fit the big shapes
  static_assert(STAGES * BUF <= 158 * 1024 && (STAGES - 2) * PER_WAVE <= 48, "LDS / counted-wait budget");
};

template <class S>
__device__ __forceinline__ void weight_grad_unit_f32(const WgParams& p, int wg, int n_wg, char* smem) {
  const long long tiles = (p.n_points + 31) / 32;
  if (wg >= tiles) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntile = wave % S::NTN;
  const int kgroup = S::KGROUPS == 1 ? 0 : wave / S::NTN;   // (a constant where every wave has its own n-tile: the epilogue then knows which k-tiles a wave holds)
  const int i = lane & 31, kk = lane >> 5;                       // operand row / column and point parity of this lane
  // byte offset of feature i's element inside a tile's 4-piece group, for point parity kk (point 2m+kk adds 32*m bytes)
  const int feat_off = (i >> 3) * kWg32PieceStride + (((i & 7) >> 2) * 32 + kk) * 16 + (i & 3) * 4;
  const unsigned lane16 = lane * 16;
  const unsigned smem_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const char* src0[S::PER_WAVE];
  long long stride[S::PER_WAVE];
  static_for<S::PER_WAVE>([&](auto e_c) {
    constexpr int e = decltype(e_c)::value;
    const int pi = wave + 8 * e;
    if (pi < S::N_DY) { src0[e] = p.grads + static_cast<long long>(p.g_slot + pi) * kPieceBytes; stride[e] = static_cast<long long>(p.grad_pieces) * kPieceBytes; }
    else if (pi < S::N_DY + S::N_X) { src0[e] = p.act + static_cast<long long>(p.x_slot + pi - S::N_DY) * kPieceBytes; stride[e] = static_cast<long long>(p.act_pieces) * kPieceBytes; }
    else { src0[e] = p.act + static_cast<long long>(p.pe_slot + pi - S::N_DY - S::N_X) * kPieceBytes; stride[e] = static_cast<long long>(p.act_pieces) * kPieceBytes; }
  });
  auto stage = [&](long long tile32, int buf) {
    if (tile32 >= tiles) tile32 = tiles - 1;
    static_for<S::PER_WAVE>([&](auto e_c) {
      constexpr int e = decltype(e_c)::value;
      const unsigned long long src_bits = reinterpret_cast<unsigned long long>(src0[e] + tile32 * stride[e]);
      const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits));
      const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits >> 32));
      const char* usrc = reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
      const unsigned lds = __builtin_amdgcn_readfirstlane(smem_addr + buf * S::BUF + (wave + 8 * e) * kWg32PieceStride);
      const unsigned go = __builtin_amdgcn_readfirstlane((wave + 8 * e < S::PIECES) ? 1u : 0u);
      const unsigned voff = lane16;
      unsigned keep;
      asm volatile(
          "s_cmp_lg_u32 %[go], 0\n\t"
          "s_cbranch_scc0 .Ldn_wg32_skip%=\n\t"
          "s_mov_b32 %[keep], m0\n\t"
          "s_mov_b32 m0, %[lds]\n\t"
          "s_nop 1\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase]" DN_WG_LOAD_POLICY "\n\t"
          "s_mov_b32 m0, %[keep]\n"
          ".Ldn_wg32_skip%=:"
          : [keep] "=&s"(keep)
          : [go] "s"(go), [lds] "s"(lds), [voff] "v"(voff), [sbase] "s"(usrc)
          : "memory", "scc");
    });
  };
  const bool full = (wave + 8 * (S::PER_WAVE - 1)) < S::PIECES;
  auto wait_tile = [&]() {
    if (full) wait_vmcnt<(S::STAGES - 2) * S::PER_WAVE>();
    else wait_vmcnt<(S::STAGES - 2) * (S::PER_WAVE - 1)>();
  };

  f32x16 acc[S::J];
#pragma unroll
  for (int k = 0; k < S::J; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

  long long tile = wg;
  int buf = 0;
#pragma unroll 1
  for (int st = 0; st + 1 < S::STAGES; ++st) stage(tile + static_cast<long long>(st) * n_wg, st);
#pragma unroll 1
  for (; tile < tiles; tile += n_wg) {
    wait_tile();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      int nb = buf + S::STAGES - 1;
      if (nb >= S::STAGES) nb -= S::STAGES;
      stage(tile + static_cast<long long>(S::STAGES - 1) * n_wg, nb);
    }
    const char* base = smem + buf * S::BUF;
    // A: this wave's 32 output features (a custom dY is one piece: rows 4h+e of lanes i < 8, the rest zero)
    const char* pa = S::CUSTOM ? base + ((i >> 2) & 1) * 32 * 16 + kk * 16 + (i & 3) * 4
                               : base + ntile * 4 * kWg32PieceStride + feat_off;
    const long long valid = p.n_points - tile * 32;
    float a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      a[m] = *reinterpret_cast<const float*>(pa + m * 32);
      if (S::CUSTOM && i >= 8) a[m] = 0.0f;
      if (2 * m + kk >= valid) a[m] = 0.0f;      // padding points of the last tile
    }
    static_for<S::J>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int kt = (S::KGROUPS == 1) ? j : kgroup + j * S::KGROUPS;   // kt >= KT: an unused accumulator
      const int ktr = kt < S::KT - 1 ? kt : S::KT - 2;
      // B: k-tile ktr.  Hidden tiles: 4 pieces each in feature order; PE tiles: a 64-wide panel is two tiles (lane half
      // = tile), a 32-wide one a single tile (lane half = column / 16)
      const char* pb;
      if (ktr < S::XT) {
        pb = base + (S::N_DY + 4 * ktr) * kWg32PieceStride + feat_off;
      } else {
        const int t = ktr - S::XT;
        const int u = (S::PET == 2) ? i : (i & 15);
        const int hh = (S::PET == 2) ? t : (i >> 4);
        pb = base + (S::N_DY + S::N_X + (u >> 2)) * kWg32PieceStride + (hh * 32 + kk) * 16 + (u & 3) * 4;
      }
      const bool is_ones = kt >= S::KT - 1;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        float b = *reinterpret_cast<const float*>(pb + m * 32);
        b = is_ones ? 1.0f : b;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b, acc[j], 0, 0, 0);
      }
    });
    buf = (buf + 1 == S::STAGES) ? 0 : buf + 1;
  }
  wait_vmcnt<0>();
  // ---- epilogue: D[row][col] in lane (col = lane & 31, half = lane >> 5), register r, row = acc_row(r, half); operand rows
  // and columns are in natural feature order here
  const int col_l = lane & 31, half = lane >> 5;
  static_for<S::J>([&](auto j_c) {
    constexpr int j = decltype(j_c)::value;
    const int kt = (S::KGROUPS == 1) ? j : kgroup + j * S::KGROUPS;
    if (kt < S::KT) {
      int col;
      if (kt < S::XT) col = 32 * kt + col_l;
      else if (kt < S::KT - 1) {
        const int t = kt - S::XT;
        const int u = (S::PET == 2) ? col_l : (col_l & 15);
        const int hh = (S::PET == 2) ? t : (col_l >> 4);
        const int pc = pe_slot_col(p.pe_L, hh, u);
        col = pc >= 0 ? p.col_pe0 + pc : -1;
      } else col = (col_l == 0) ? -2 : -1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int n = 32 * ntile + acc_row(r, half);
        if constexpr (S::CUSTOM) {
          n = acc_row(r, half);
          if (n >= p.custom_rows) continue;
        }
        const float val = acc[j][r];
        if (p.part != nullptr) {
          float* part_w = p.part + static_cast<long long>(wg) * p.part_stride;
          if (col >= 0) part_w[static_cast<long long>(n) * p.ldw + col] = val;
          else if (col == -2 && p.db != nullptr) part_w[static_cast<long long>(p.n_real) * p.ldw + n] = val;
        } else if (col >= 0) atomicAdd(p.dW + static_cast<long long>(n) * p.ldw + col, val);
        else if (col == -2 && p.db != nullptr) atomicAdd(p.db + n, val);
      }
    }
  });
}

// the instantiated layer shapes: W = 256 and W = 128 nets (L_xyz = 10: a 64-wide xyz panel = 2 tiles; L_dir: 1 tile)
#define DN_WG_SHAPES(X)                                                                                        \
  X(0, 8, 0, 2, false) X(1, 8, 8, 0, false) X(2, 8, 8, 2, false) X(3, 4, 8, 1, false) X(4, 1, 8, 0, true)     \
  X(5, 1, 4, 0, true)  X(6, 4, 0, 2, false) X(7, 4, 4, 0, false) X(8, 4, 4, 2, false) X(9, 2, 4, 1, false)    \
  X(10, 1, 2, 0, true)

static int wg_shape_index(int ntn, int xt, int pet, bool custom) {
#define X(id, a, b, c, d) if (ntn == a && xt == b && pet == c && custom == d) return id;
  DN_WG_SHAPES(X)
#undef X
  return -1;
}
static int wg_shape_pieces(int shape) {
#define X(id, a, b, c, d) if (shape == id) return WgShape<a, b, c, d>::PIECES;
  DN_WG_SHAPES(X)
#undef X
  return 0;
}
static int wg_shape_lds(int shape) {
#define X(id, a, b, c, d) if (shape == id) return WgShape<a, b, c, d>::STAGES * WgShape<a, b, c, d>::PIECES * WgShape<a, b, c, d>::PSTRIDE;
  DN_WG_SHAPES(X)
#undef X
  return 0;
}

__device__ __forceinline__ void weight_grad_dispatch(const WgParams& p, int wg, int n_wg, char* smem) {
  switch (p.shape) {  // workgroup-uniform
#ifdef DN_WG_ONLY   // developer hook: compile a single shape (register-pressure bisection)
#define X(id, a, b, c, d) case id: if constexpr (id == DN_WG_ONLY) weight_grad_unit<WgShape<a, b, c, d>>(p, wg, n_wg, smem); break;
#else
#define X(id, a, b, c, d) case id: weight_grad_unit<WgShape<a, b, c, d>>(p, wg, n_wg, smem); break;
#endif
    DN_WG_SHAPES(X)
#undef X
    default: break;
  }
}

__device__ __forceinline__ void weight_grad_dispatch_s8(const WgParams& p, int wg, int n_wg, char* smem) {
  switch (p.shape) {
#define X(id, a, b, c, d) case id: weight_grad_unit<WgShape<a, b, c, d, true>>(p, wg, n_wg, smem); break;
    DN_WG_SHAPES(X)
#undef X
    default: break;
  }
}
// shapes 5 .. 10: the layers of a W = 128 net (at most 3 accumulator tiles per wave): 128 VGPRs, two workgroups per CU
#define DN_WG_SMALL_SHAPES(X) X(5, 1, 4, 0, true) X(6, 4, 0, 2, false) X(7, 4, 4, 0, false) X(8, 4, 4, 2, false) X(9, 2, 4, 1, false) X(10, 1, 2, 0, true)
constexpr int kWgFirstSmallShape = 5;
__device__ __forceinline__ void weight_grad_dispatch_s8_small(const WgParams& p, int wg, int n_wg, char* smem) {
  switch (p.shape) {
#define X(id, a, b, c, d) case id: weight_grad_unit<WgShape<a, b, c, d, true, kWgLdsBytesSmall>>(p, wg, n_wg, smem); break;
    DN_WG_SMALL_SHAPES(X)
#undef X
    default: break;
  }
}
static int wg_shape_pieces_s8(int shape) {
#define X(id, a, b, c, d) if (shape == id) return WgShape<a, b, c, d, true>::PIECES;
  DN_WG_SHAPES(X)
#undef X
  return 0;
}
static int wg_shape_cost_s8(int shape) {
#define X(id, a, b, c, d) if (shape == id) return WgShape<a, b, c, d, true>::COST;
  DN_WG_SHAPES(X)
#undef X
  return 0;
}

__device__ __forceinline__ void weight_grad_dispatch_f32(const WgParams& p, int wg, int n_wg, char* smem) {
  switch (p.shape) {
#define X(id, a, b, c, d) case id: weight_grad_unit_f32<WgShape32<a, b, c, d>>(p, wg, n_wg, smem); break;
    DN_WG_SHAPES(X)
#undef X
    default: break;
  }
}
static int wg_shape_pieces_f32(int shape) {
#define X(id, a, b, c, d) if (shape == id) return WgShape32<a, b, c, d>::PIECES;
  DN_WG_SHAPES(X)
#undef X
  return 0;
}

__global__ __launch_bounds__(512, 2) void weight_grad_kernel(WgParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  weight_grad_dispatch(p, blockIdx.x, gridDim.x, smem);
}

// All linear layers of a network in ONE launch: the workgroups are divided among the units in proportion to the
// bytes each unit streams, so a unit's gradient is the sum of a few dozen partials instead of one per workgroup of
// a whole-chip launch (the fp32 atomics of the epilogue are expensive: ~1 lane-op per L2 channel per clock).
constexpr int kWgMaxUnits = 32;   // two D <= 12 networks of one training step in one batch
struct WgBatch {
  int n_units;
  int wg_begin[kWgMaxUnits + 1];  // unit u owns workgroups [wg_begin[u], wg_begin[u+1])
  WgParams u[kWgMaxUnits];
};

__global__ __launch_bounds__(512, 2) void weight_grad_batch_kernel(WgBatch b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int u = 0;
  while (u + 1 < b.n_units && static_cast<int>(blockIdx.x) >= b.wg_begin[u + 1]) ++u;
  const WgParams p = b.u[u];  // by value: the fields live in SGPRs, not behind kernarg loads inside the tile loop
  weight_grad_dispatch(p, static_cast<int>(blockIdx.x) - b.wg_begin[u], b.wg_begin[u + 1] - b.wg_begin[u], smem);
}

__global__ __launch_bounds__(512, 2) void weight_grad_batch_kernel_s8(WgBatch b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int u = 0;
  while (u + 1 < b.n_units && static_cast<int>(blockIdx.x) >= b.wg_begin[u + 1]) ++u;
  const WgParams p = b.u[u];
  weight_grad_dispatch_s8(p, static_cast<int>(blockIdx.x) - b.wg_begin[u], b.wg_begin[u + 1] - b.wg_begin[u], smem);
}

__global__ __launch_bounds__(512, 4) void weight_grad_batch_kernel_s8_small(WgBatch b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int u = 0;
  while (u + 1 < b.n_units && static_cast<int>(blockIdx.x) >= b.wg_begin[u + 1]) ++u;
  const WgParams p = b.u[u];
  weight_grad_dispatch_s8_small(p, static_cast<int>(blockIdx.x) - b.wg_begin[u], b.wg_begin[u + 1] - b.wg_begin[u], smem);
}

__global__ __launch_bounds__(512, 2) void weight_grad_batch_kernel_f32(WgBatch b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int u = 0;
  while (u + 1 < b.n_units && static_cast<int>(blockIdx.x) >= b.wg_begin[u + 1]) ++u;
  const WgParams p = b.u[u];
  weight_grad_dispatch_f32(p, static_cast<int>(blockIdx.x) - b.wg_begin[u], b.wg_begin[u + 1] - b.wg_begin[u], smem);
}

// second phase of the deterministic reduction: dW / db += the partials of the unit's workgroups, in workgroup order.  One element per
// thread and step, consecutive threads on consecutive floats of every partial; blockIdx.y = the unit.
__global__ __launch_bounds__(256) void wg_reduce_kernel(WgBatch b) {
  const int u = static_cast<int>(blockIdx.y);
  const float* part = b.u[u].part;
  if (part == nullptr) return;
  const int n_wg = b.wg_begin[u + 1] - b.wg_begin[u];
  const long long stride = b.u[u].part_stride;
  const long long n_w = static_cast<long long>(b.u[u].n_real) * b.u[u].ldw;
  const long long total = n_w + (b.u[u].db != nullptr ? b.u[u].n_real : 0);
  float* dW = b.u[u].dW;
  float* db = b.u[u].db;
  // (the order is fixed, not sequential: four running sums over the partials w = 0, 1, 2, 3 (mod 4), combined at the end - four loads in
  // flight per thread instead of a chain of dependent ones)
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int w = 0;
    for (; w + 3 < n_wg; w += 4) {
      const float a = part[w * stride + e], b2 = part[(w + 1) * stride + e], c = part[(w + 2) * stride + e], d = part[(w + 3) * stride + e];
      s0 += a; s1 += b2; s2 += c; s3 += d;
    }
    for (; w < n_wg; ++w) s0 += part[w * stride + e];
    const float sum = (s0 + s1) + (s2 + s3);
    if (e < n_w) dW[e] += sum;
    else db[e - n_w] += sum;
  }
}

static int wg_fill(const dn_mlp_desc* desc, const TrainLayout& t, const void* act, const void* grads, int64_t n_points,
                   int g_slot, int n_out, int x_slot, int x_width, int pe_kind, float* dW, int ldw, float* db, WgParams* out,
                   bool s8 = false) {
  const int custom_rows = (n_out < 32) ? n_out : 0;  // fc_rgb (3) / fc_alpha (1) / fc_out (4): one custom dY piece
  const int n_real = n_out;
  if (custom_rows) n_out = 32;
  DN_REQUIRE(n_out % 32 == 0 && x_width % 32 == 0 && pe_kind >= 0 && pe_kind <= 2, "weight_grad: bad layer shape");
  WgParams p{};
  p.act = static_cast<const char*>(act);
  p.grads = static_cast<const char*>(grads);
  p.act_pieces = t.act_pieces; p.grad_pieces = t.grad_pieces;
  p.n_points = n_points;
  p.g_slot = g_slot; p.custom_rows = custom_rows;
  p.x_slot = x_slot;
  p.pe_slot = pe_kind == 1 ? t.slot_xyz : t.slot_dir;
  const int pe_tiles = pe_kind == 0 ? 0 : (pe_kind == 1 ? t.kxp : t.kdp) / t.ppt;   // pieces per 32-column tile: 2 (bf16) / 4 (fp32)
  p.pe_L = pe_kind == 1 ? desc->num_encoding_fn_xyz : desc->num_encoding_fn_dir;
  p.pe_kind = pe_kind;
  p.dW = dW; p.ldw = ldw; p.col_pe0 = x_width; p.db = db;
  p.n_real = n_real; p.part = nullptr; p.part_stride = 0;
  p.shape = wg_shape_index(n_out / 32, x_width / 32, pe_tiles, custom_rows > 0);
  if (p.shape < 0) {
    set_error("weight_grad: no kernel instance for a %d x (%d + %d) layer%s", n_out, x_width, 32 * pe_tiles, custom_rows ? " (custom dY)" : "");
    return DN_E_UNSUPPORTED;
  }
  if (s8) {
    // 8-bit buffers (s8-48 layout): `t` carries the layout in HALF-unit numbers (wg_layout_s8): two pieces side by side per 1 KiB
    // unit - every slot / stride below becomes a count of staged units; all slots of a layer are even except a custom dY piece
    // (fc_rgb: even half, fc_alpha: odd half of the custom unit)
    DN_REQUIRE(custom_rows || (p.g_slot % 2) == 0, "weight_grad (8-bit buffers): odd gradient slot");
    DN_REQUIRE((p.x_slot % 2) == 0 && (p.pe_slot % 2) == 0, "weight_grad (8-bit buffers): odd activation slot");
    p.dy_odd = p.g_slot & 1;
    p.g_slot /= 2; p.x_slot /= 2; p.pe_slot /= 2;
    p.act_pieces = (t.act_pieces + 1) / 2; p.grad_pieces = (t.grad_pieces + 1) / 2;
    p.stats_block = reinterpret_cast<unsigned*>(const_cast<char*>(static_cast<const char*>(grads)) +
                                                g48_padded_records(n_points) * 2 * (t.grad_pieces / 4) * kPieceBytes);
    p.scale_word = p.stats_block + kS8BlockScale;
  }
  *out = p;
  return 0;
}

// The s8-48 layout (TrainLayout48: units per 16-point group) in the numbering wg_fill halves for the 8-bit kernel: a record of two
// groups holds 2 x units_per_group staged units, the two groups' units of slot s at staged positions 2 s and 2 s + 1 - so slot s is
// "half-unit" 4 s, a hidden vector spans 4 kh_u of them, and the custom unit's second piece (d alpha) is half-unit 4 s + 1.
static void wg_layout_s8(const dn_mlp_desc& d, TrainLayout* t) {
  TrainLayout48 u;
  build_train_layout48(d, &u);
  *t = TrainLayout{};
  t->kpp = 16; t->epp = 8; t->ppt = 2;
  t->kxp = 4; t->kdp = d.use_viewdirs ? 2 : 0;   // -> 2 / 1 32-column tiles (wg_fill)
  t->kh = 4 * u.kh_u;
  t->slot_xyz = 4 * u.slot_xyz; t->slot_dir = 4 * u.slot_dir; t->slot_layer1 = 4 * u.slot_layer1; t->slot_trunk0 = 4 * u.slot_trunk0;
  t->slot_feat = 4 * u.slot_feat; t->slot_dirout = 4 * u.slot_dirout;
  t->act_pieces = 4 * u.act_units;
  t->mask_words = u.mask_stages;
  t->gslot_dirout = 4 * u.gslot_dirout; t->gslot_feat = 4 * u.gslot_feat; t->gslot_trunk0 = 4 * u.gslot_trunk0; t->gslot_layer1 = 4 * u.gslot_layer1;
  t->gslot_out = 4 * u.gslot_out;
  t->grad_pieces = 4 * u.grad_units;
}

template <class K>
static int wg_attr(K kern) { return ensure_big_lds(reinterpret_cast<const void*>(kern)); }

}  // namespace dn

namespace dn {

// append the (dW, db) units of ONE network to a batch; unit_tiles[u] = its 32-point tiles (networks of a batch differ in points)
static int wg_add_network(const dn_mlp_desc* desc, int precision, bool s8, const void* act, const void* grads, int64_t n_points,
                          float* const* h_dW, float* const* h_db, WgBatch& b, long long* unit_tiles, const char* who) {
  const bool f32 = precision == DN_PREC_F32;
  DN_REQUIRE(act && grads && h_dW && h_db && n_points > 0, "%s: bad arguments", who);
  DN_REQUIRE(desc->num_encoding_fn_xyz == 10, "%s: training kernels are built for L_xyz = 10", who);
  int rc;
  TrainLayout t;
  if (s8) {
    DN_REQUIRE(g48_train_supported(*desc), "%s: no 8-bit-saved-tensor training kernels for this network (see dn_mlp_train_sizes)", who);
    wg_layout_s8(*desc, &t);
  } else {
    build_train_layout(*desc, precision, &t);
  }
  (void)f32;
  NetLayout L;
  build_layout(*desc, precision, &L);
  const int W = desc->hidden_size, D = desc->num_layers;
  const int dim_xyz = 3 + 6 * desc->num_encoding_fn_xyz, dim_dir = 3 + 6 * desc->num_encoding_fn_dir;
  const int n_units = D + (desc->use_viewdirs ? 4 : 1);
  const int u0 = b.n_units;
  DN_REQUIRE(u0 + n_units <= kWgMaxUnits, "%s: too many layers (%d)", who, u0 + n_units);
  for (int i = 0; i < n_units; ++i) DN_REQUIRE(h_dW[i] && h_db[i], "%s: gradient tensor %d is NULL", who, i);
  // parameter order: layer1, layers_xyz[0..D-2], then layers_dir.0, fc_alpha, fc_rgb, fc_feat | fc_out (models.py:207-229)
  int u = b.n_units;
  if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_layer1, W, 0, 0, 1, h_dW[u - u0], dim_xyz, h_db[u - u0], &b.u[u], s8))) return rc;
  ++u;
  int x_slot = t.slot_layer1;
  for (int i = 0; i + 1 < D; ++i, ++u) {
    const bool skip = (L.skip_mask >> i) & 1u;
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_trunk0 + i * t.kh, W, x_slot, W, skip ? 1 : 0, h_dW[u - u0],
                      W + (skip ? dim_xyz : 0), h_db[u - u0], &b.u[u], s8)))
      return rc;
    x_slot = t.slot_trunk0 + i * t.kh;
  }
  if (desc->use_viewdirs) {
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_dirout, W / 2, t.slot_feat, W, 2, h_dW[u - u0], W + dim_dir, h_db[u - u0], &b.u[u], s8))) return rc;
    ++u;
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_out + 1, 1, x_slot, W, 0, h_dW[u - u0], W, h_db[u - u0], &b.u[u], s8))) return rc;
    ++u;
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_out, 3, t.slot_dirout, W / 2, 0, h_dW[u - u0], W / 2, h_db[u - u0], &b.u[u], s8))) return rc;
    ++u;
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_feat, W, x_slot, W, 0, h_dW[u - u0], W, h_db[u - u0], &b.u[u], s8))) return rc;
    ++u;
  } else {
    if ((rc = wg_fill(desc, t, act, grads, n_points, t.gslot_out, 4, x_slot, W, 0, h_dW[u - u0], W, h_db[u - u0], &b.u[u], s8))) return rc;
    ++u;
  }
  for (int i = u0; i < u0 + n_units; ++i) unit_tiles[i] = (n_points + 31) / 32;
  b.n_units = u0 + n_units;
  return 0;
}

bool weight_grad_pair_fits(const dn_mlp_desc& d) { return 2 * (d.num_layers + (d.use_viewdirs ? 4 : 1)) <= kWgMaxUnits; }

// share the workgroups (one per CU) among the batch's units and launch
constexpr long long wg_part_stride(int n_real, int ldw) { return ((static_cast<long long>(n_real) * (ldw + 1) + 63) / 64) * 64; }

static int wg_launch_batch(WgBatch& b, const long long* unit_tiles, bool f32, bool s8, dn_stream_t stream, void* scratch = nullptr,
                           size_t scratch_bytes = 0) {
  int rc;
  const int n_units = b.n_units;
  // One workgroup per CU - or, for an 8-bit batch of small layer shapes only (the W = 128 nets), two on half the LDS each: those
  // launches are bound by the wait / barrier / LDS round trip of a tile, which the second resident workgroup hides.  (Round 3 tried
  // this with the atomics reduction and dropped it - the tile loop got 13-26 us shorter, the reduction of twice as many partials
  // 50 us longer, HISTORY.md section 4.7c; with partial slabs and the fixed-order second launch the reduction no longer grows that way.)
  bool small = s8 && !f32 && std::getenv("DEXNERF_WG_ONE_PER_CU") == nullptr;
  for (int i = 0; i < n_units; ++i) small = small && b.u[i].shape >= kWgFirstSmallShape;
  int total_wg = device_cus() * (small ? 2 : 1);
  if (total_wg < n_units) total_wg = n_units;
  long long cost[kWgMaxUnits], cost_sum = 0;
  for (int i = 0; i < n_units; ++i) {
    // (per-tile cost of the layer's shape x its tiles: the networks of one batch may differ in points)
    cost[i] = (f32 ? wg_shape_pieces_f32(b.u[i].shape) : (s8 ? wg_shape_cost_s8(b.u[i].shape) : wg_shape_pieces(b.u[i].shape))) * unit_tiles[i];
    cost_sum += cost[i];
  }
  int share[kWgMaxUnits], given = 0;
  long long rem[kWgMaxUnits];
  for (int i = 0; i < n_units; ++i) {
    share[i] = static_cast<int>(cost[i] * total_wg / cost_sum);
    rem[i] = cost[i] * total_wg % cost_sum;
    if (share[i] < 1) { share[i] = 1; rem[i] = 0; }
    given += share[i];
  }
  while (given < total_wg) {
    int best = 0;
    for (int i = 1; i < n_units; ++i) if (rem[i] > rem[best]) best = i;
    ++share[best]; rem[best] = -1; ++given;
  }
  b.wg_begin[0] = 0;
  for (int i = 0; i < n_units; ++i) {
    if (share[i] > unit_tiles[i]) share[i] = static_cast<int>(unit_tiles[i]);  // idle workgroups would exit at once anyway
    b.wg_begin[i + 1] = b.wg_begin[i] + share[i];
  }
  // deterministic reduction: a partial per workgroup in the caller's scratch, added up in workgroup order by a second launch
  bool two_phase = false;
  if (scratch != nullptr) {
    long long floats = 0;
    for (int i = 0; i < n_units; ++i) floats += static_cast<long long>(b.wg_begin[i + 1] - b.wg_begin[i]) * wg_part_stride(b.u[i].n_real, b.u[i].ldw);
    DN_REQUIRE(static_cast<size_t>(floats) * sizeof(float) <= scratch_bytes && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0,
               "weight gradients: the reduction scratch holds %zu bytes, %lld are needed (dn_mlp_weight_grad_scratch_bytes), 256-byte aligned",
               scratch_bytes, floats * 4);
    float* at = static_cast<float*>(scratch);
    for (int i = 0; i < n_units; ++i) {
      b.u[i].part = at;
      b.u[i].part_stride = static_cast<int>(wg_part_stride(b.u[i].n_real, b.u[i].ldw));
      at += static_cast<long long>(b.wg_begin[i + 1] - b.wg_begin[i]) * b.u[i].part_stride;
    }
    two_phase = true;
  }
  if ((rc = f32 ? wg_attr(weight_grad_batch_kernel_f32) : (small ? wg_attr(weight_grad_batch_kernel_s8_small) : (s8 ? wg_attr(weight_grad_batch_kernel_s8) : wg_attr(weight_grad_batch_kernel))))) return rc;
#ifdef DN_WG_STAMP   // diagnostic build: synchronous, allocates, prints - never part of the shipped library
  static unsigned long long* stamp_buf = nullptr;
  const size_t stamp_words = static_cast<size_t>(b.wg_begin[n_units]) * 8 * 8;
  if (!stamp_buf) (void)hipMalloc(&stamp_buf, 1024 * 8 * 8 * sizeof(unsigned long long));
  (void)hipMemsetAsync(stamp_buf, 0, stamp_words * sizeof(unsigned long long), as_stream(stream));
  for (int i = 0; i < n_units; ++i) b.u[i].stamp = stamp_buf;
#endif
  if (small)
    hipLaunchKernelGGL(weight_grad_batch_kernel_s8_small, dim3(static_cast<unsigned>(b.wg_begin[n_units])), dim3(512), kWgLdsBytesSmall,
                       as_stream(stream), b);
  else if (s8)
    hipLaunchKernelGGL(weight_grad_batch_kernel_s8, dim3(static_cast<unsigned>(b.wg_begin[n_units])), dim3(512), kWgLdsBytes,
                       as_stream(stream), b);
  else if (f32)
    hipLaunchKernelGGL(weight_grad_batch_kernel_f32, dim3(static_cast<unsigned>(b.wg_begin[n_units])), dim3(512), 158 * 1024,
                       as_stream(stream), b);
  else
    hipLaunchKernelGGL(weight_grad_batch_kernel, dim3(static_cast<unsigned>(b.wg_begin[n_units])), dim3(512), kWgLdsBytes16,
                       as_stream(stream), b);
#ifdef DN_WG_STAMP
  {
    (void)hipStreamSynchronize(as_stream(stream));
    std::vector<unsigned long long> h(stamp_words);
    (void)hipMemcpy(h.data(), stamp_buf, stamp_words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (int i = 0; i < n_units; ++i) {
      double a[6] = {}; int waves = 0; double shape = -1, jj = 0;
      for (int wgi = b.wg_begin[i]; wgi < b.wg_begin[i + 1]; ++wgi)
        for (int w = 0; w < 8; ++w) {
          const unsigned long long* o = &h[(static_cast<size_t>(wgi) * 8 + w) * 8];
          if (o[4] == 0) continue;
          for (int k = 0; k < 6; ++k) a[k] += static_cast<double>(o[k]);
          shape = static_cast<double>(o[6]); jj = static_cast<double>(o[7]); ++waves;
        }
      if (waves)
        fprintf(stderr, "[wg-stamp] unit %2d shape %2.0f J %2.0f, %3d workgroups: per tile: wait %.0f, barrier %.0f, stage %.0f, consume %.0f ticks; %.0f tiles per wave, loop %.0f ticks per wave\n",
                i, shape, jj, b.wg_begin[i + 1] - b.wg_begin[i], a[0] / a[4], a[1] / a[4], a[2] / a[4], a[3] / a[4], a[4] / waves, a[5] / waves);
    }
  }
#endif
  if ((rc = check_launch("dn_mlp_weight_grad_all"))) return rc;
  if (two_phase) {
    hipLaunchKernelGGL(wg_reduce_kernel, dim3(48, static_cast<unsigned>(n_units)), dim3(256), 0, as_stream(stream), b);
    return check_launch("weight-gradient reduction");
  }
  return 0;
}

}  // namespace dn

// Bytes of reduction scratch dn_mlp_weight_grad_all_ws (n_networks = 1) / dn_mlp_weight_grad_pair_ws (2) need for networks of this
// architecture: one partial (dW + db of the layer it works on) per workgroup of the launch, every workgroup priced at the largest layer.
extern "C" size_t dn_mlp_weight_grad_scratch_bytes(const dn_mlp_desc* desc, int n_networks) {
  if (desc == nullptr || n_networks < 1 || n_networks > 2) return 0;
  const int W = desc->hidden_size;
  const int dim_xyz = 3 + 6 * desc->num_encoding_fn_xyz, dim_dir = 3 + 6 * desc->num_encoding_fn_dir;
  const int widest = W + (dim_xyz > dim_dir ? dim_xyz : dim_dir);
  const int n_units = n_networks * (desc->num_layers + (desc->use_viewdirs ? 4 : 1));
  int total_wg = dn::device_cus() * (W <= 128 ? 2 : 1);   // (the W = 128 nets' 8-bit launch: two workgroups per CU)
  if (total_wg < n_units) total_wg = n_units;
  return static_cast<size_t>(total_wg + n_units) * static_cast<size_t>(dn::wg_part_stride(W, widest)) * sizeof(float);
}

extern "C" int dn_mlp_weight_grad_all_ws(const dn_mlp_desc* desc, int precision, const void* act, const void* grads, int64_t n_points,
                                         float* const* h_dW, float* const* h_db, void* scratch, size_t scratch_bytes, dn_stream_t stream) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;   // same layouts and slots, half-size pieces
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(precision == DN_PREC_BF16 || precision == DN_PREC_F32, "dn_mlp_weight_grad_all: bf16 or fp32 buffers (fp16 is a render-only mode)");
  DN_REQUIRE(n_points >= 0, "dn_mlp_weight_grad_all: bad arguments");
  if (n_points == 0) return 0;
  WgBatch b{};
  long long unit_tiles[kWgMaxUnits];
  if ((rc = wg_add_network(desc, precision, s8, act, grads, n_points, h_dW, h_db, b, unit_tiles, "dn_mlp_weight_grad_all"))) return rc;
  return wg_launch_batch(b, unit_tiles, precision == DN_PREC_F32, s8, stream, scratch, scratch_bytes);
}

extern "C" int dn_mlp_weight_grad_all(const dn_mlp_desc* desc, int precision, const void* act, const void* grads,
                                      int64_t n_points, float* const* h_dW, float* const* h_db, dn_stream_t stream) {
  return dn_mlp_weight_grad_all_ws(desc, precision, act, grads, n_points, h_dW, h_db, nullptr, 0, stream);
}

// The weight gradients of TWO networks of one architecture - the coarse and the fine network of a training step - in ONE launch:
// the workgroups are shared among all their layers by cost x points, so a layer's gradient is the sum of half as many partials as
// with two launches and the launch's fixed costs (pipeline fill, the reduction epilogue, the launch itself) are paid once.
extern "C" int dn_mlp_weight_grad_pair_ws(const dn_mlp_desc* desc, int precision, const void* act_a, const void* grads_a, int64_t n_points_a,
                                          float* const* h_dW_a, float* const* h_db_a, const void* act_b, const void* grads_b,
                                          int64_t n_points_b, float* const* h_dW_b, float* const* h_db_b, void* scratch, size_t scratch_bytes,
                                          dn_stream_t stream) {
  const bool s8 = precision == DN_PREC_BF16_S8;
  if (s8) precision = DN_PREC_BF16;
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(precision == DN_PREC_BF16 || precision == DN_PREC_F32, "dn_mlp_weight_grad_pair: bf16 or fp32 buffers (fp16 is a render-only mode)");
  DN_REQUIRE(n_points_a > 0 && n_points_b > 0, "dn_mlp_weight_grad_pair: both networks need points");
  WgBatch b{};
  long long unit_tiles[kWgMaxUnits];
  if ((rc = wg_add_network(desc, precision, s8, act_a, grads_a, n_points_a, h_dW_a, h_db_a, b, unit_tiles, "dn_mlp_weight_grad_pair"))) return rc;
  if ((rc = wg_add_network(desc, precision, s8, act_b, grads_b, n_points_b, h_dW_b, h_db_b, b, unit_tiles, "dn_mlp_weight_grad_pair"))) return rc;
  return wg_launch_batch(b, unit_tiles, precision == DN_PREC_F32, s8, stream, scratch, scratch_bytes);
}

extern "C" int dn_mlp_weight_grad_pair(const dn_mlp_desc* desc, int precision, const void* act_a, const void* grads_a, int64_t n_points_a,
                                       float* const* h_dW_a, float* const* h_db_a, const void* act_b, const void* grads_b,
                                       int64_t n_points_b, float* const* h_dW_b, float* const* h_db_b, dn_stream_t stream) {
  return dn_mlp_weight_grad_pair_ws(desc, precision, act_a, grads_a, n_points_a, h_dW_a, h_db_a, act_b, grads_b, n_points_b, h_dW_b, h_db_b,
                                    nullptr, 0, stream);
}

extern "C" int dn_mlp_weight_grad(const dn_mlp_desc* desc, int precision, const void* act, const void* grads,
                                  int64_t n_points, int g_slot, int n_out, int x_slot, int x_width, int pe_kind,
                                  float* dW, int ldw, float* db, dn_stream_t stream) {
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(precision == DN_PREC_BF16, "dn_mlp_weight_grad: bf16 buffers only (fp32 mode forms dW with library GEMMs)");
  DN_REQUIRE(act && grads && dW && n_points >= 0, "dn_mlp_weight_grad: bad arguments");
  if (n_points == 0) return 0;
  TrainLayout t;
  build_train_layout(*desc, precision, &t);
  WgParams p{};
  if ((rc = wg_fill(desc, t, act, grads, n_points, g_slot, n_out, x_slot, x_width, pe_kind, dW, ldw, db, &p))) return rc;
  if ((rc = wg_attr(weight_grad_kernel))) return rc;
  const int cus = device_cus();
  const long long tiles = (n_points + 31) / 32;
  const long long grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(weight_grad_kernel, dim3(static_cast<unsigned>(grid)), dim3(512), wg_shape_lds(p.shape), as_stream(stream), p);
  return check_launch("dn_mlp_weight_grad");
}

extern "C" int dn_set_s8_grad_scale(float scale) {
  int e = 0;
  DN_REQUIRE(scale == 0.0f || (scale > 0.0f && std::isfinite(scale) && std::frexp(scale, &e) == 0.5f),
             "dn_set_s8_grad_scale: the scale must be a power of two (the scaling and its inverse are then exact), or 0 = chosen per launch from the largest upstream gradient");
  g_s8_grad_scale.store(scale);
  return 0;
}
