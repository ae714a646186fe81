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
//     L2 into a 5-slot x 16 KiB LDS ring shared by the waves of the workgroup; one counted
//     `s_waitcnt vmcnt(N)` + one raw `s_barrier` per 16 pieces; the next phase is always fully landed, two more are
//     in flight;
//   * positional encodings are computed in registers straight into B-fragment layout (the reference
//     materialises a (P,90) tensor and recomputes the direction encoding per sample);
//   * bf16 / fp16 modes: v_mfma_f32_32x32x16_{bf16,f16}, 8 waves x 32 points per workgroup, 2 waves / SIMD;
//     fp32 mode: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), 4 waves x 32 points, 1 wave / SIMD.
// MFMA-bound: 1,186,816 FLOP per point (D8/W256) against 16 B written per point.  SAVE = the training forward: the same
// chain also streams every stage's output pieces (non-temporal, scalar-base stores) and ReLU mask words to HBM.
#include "mlp_device.h"
#include "mlp_internal.h"
#include "mlp_geo48.h"

namespace dn {

// SAVE = training forward: every stage's output pieces (and both encodings) are also written to `p.act` in the
// wave-native piece layout, plus one 128-bit ReLU mask word per lane per masked stage to `p.masks`.
template <int W, int LX, int LD, int BF16, int PT, int SAVE>   // SAVE: 0 inference, 1 training forward (bf16 / fp32 pieces), 2 training forward with 8-bit saved pieces
__global__ __launch_bounds__((waves_of<BF16, PT>() * 64), ((BF16 && PT == 1) ? 2 : 1)) void mlp_forward_kernel(FwdParams p) {
  using P = Prec<BF16>;
  using BPiece = typename P::BPiece;
  constexpr int NT = W / 32;
  constexpr int KH = NT * P::PPT;                      // hidden pieces of a W-wide input
  constexpr int KXP = kXyzPanel / (2 * P::EPP);                 // PE xyz pieces (fixed 64-wide panel)
  static_assert(3 + 6 * LX <= kXyzPanel, "xyz encoding wider than its K panel");
  constexpr int KDP = round_up(3 + 6 * LD, 16) / (2 * P::EPP);  // PE dir pieces
  constexpr int WAVES = waves_of<BF16, PT>();
  constexpr int PTS_PER_WAVE = 32 * PT;
  constexpr int PTS_PER_WG = PTS_PER_WAVE * WAVES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* bias_lds = smem + kRingBytes;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;
  const int j = lane & 31;
  // per-wave input staging rows (PTS_PER_WAVE floats each): 0-2 origin / point, 3-5 direction, 6 depth, 7-9 view dir
  float* inbuf = reinterpret_cast<float*>(smem + kRingBytes + p.bias_bytes) + wave * (kInRows * PTS_PER_WAVE);
  // per-wave copy of the xyz-encoding B pieces (re-read at layer1 and at the skip layers instead of pinning
  // registers for the whole trunk)
  char* pex = smem + kRingBytes + p.bias_bytes + WAVES * kInRows * PTS_PER_WAVE * 4 +
              wave * (PT * (KXP + KDP) * kPieceBytes) + lane * 16;
  char* ped = pex + PT * KXP * kPieceBytes;  // the view-direction encoding pieces, used once near the end of the tile

  // Stage the inputs of a tile by LDS-DMA (4 B per lane, per-lane source address): no VGPR-destination load is
  // ever in flight next to the weight DMAs, so the compiler never drains the pipeline with vmcnt(0).
  // Lane l (< PTS_PER_WAVE) stages point l of this wave's PTS_PER_WAVE points.
  auto issue_inputs = [&](long long tile) {
    long long pt = tile * PTS_PER_WG + wave * PTS_PER_WAVE + lane;
    if (pt >= p.n_points) pt = p.n_points - 1;
    auto dma = [&](const float* src, int row) {
      if (lane < PTS_PER_WAVE)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(inbuf + row * PTS_PER_WAVE), 4, 0, 0);
    };
    if (p.mode == 0) {
      const float* r = p.rays + (pt / p.S) * p.ray_stride;
#pragma unroll
      for (int c = 0; c < 6; ++c) dma(r + c, c);
      dma(p.z + pt, 6);
      if (p.use_viewdirs) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dma(r + 8 + c, 7 + c);
      }
    } else if (p.mode == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dma(p.pts + pt * 3 + c, c);
      if (p.use_viewdirs) {
        const float* v = p.viewdirs + (pt / p.S) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dma(v + c, 7 + c);
      }
    }
  };

  // biases -> LDS once per workgroup (fp32, pre-permuted [tile][half][16])
  {
    const f32x4* g = reinterpret_cast<const f32x4*>(p.packed);
    f32x4* l = reinterpret_cast<f32x4*>(bias_lds);
    for (int i = threadIdx.x; i < p.bias_bytes / 16; i += WAVES * 64) l[i] = g[i];
  }
  issue_inputs(blockIdx.x);

  Pipe<WAVES> pipe;
  pipe.ring = ring;
  pipe.ring_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring));
  pipe.lane16 = lane * 16;
  pipe.wsrc = p.packed + p.bias_bytes;
  pipe.total_bytes = static_cast<unsigned>(p.total_pieces) * kPieceBytes;
  pipe.q_issue = 0;
  pipe.slot_wr = 0;
  pipe.wave = wave;
#pragma unroll
  for (int ph = 0; ph < kRingPhases - 1; ++ph) pipe.issue_phase();
  // one-time full drain: biases, first inputs and the first five phases are resident
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  pipe.slot_nxt = 0;
  pipe.rd_cur = ring + lane * 16;
  pipe.rd_nxt = ring + lane * 16;  // phase_begin() of phase 0 turns this into rd_cur
#pragma unroll
  for (int e = 0; e < kPrefetch; ++e)
    pipe.af[e] = *reinterpret_cast<const f32x4*>(pipe.rd_nxt + e * kPieceBytes);

  const char* bias_half = bias_lds + h * 64;

  for (long long tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    // ---- inputs: lanes j and j+32 share the points {t*32 + j} of this wave ----
    // Everything a lane derives from its inputs (both encodings) goes to LDS here, so no VGPR is carried across
    // the trunk: spilled carries would be reloaded with vmcnt(0) waits that drain the weight pipeline.
    if (p.mode != 2) {
      // staged by this wave's own DMAs one tile ago; VMEM ops retire in order, so every counted wait since then
      // (>= 70 phases, each leaving at most 6/12 younger ops outstanding) has covered them
      float in[PT][10];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < 10; ++c) in[t][c] = inbuf[c * PTS_PER_WAVE + t * 32 + j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long nxt = tile + gridDim.x;
      if (nxt < p.n_tiles) issue_inputs(nxt);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)  // plain mul then add (train_utils.py:136)
          x[c] = (p.mode == 0) ? in[t][c] + in[t][3 + c] * in[t][6] : in[t][c];
        BPiece bx[KXP];
        encode_pieces<BF16, LX, KXP>(x, p.fx, h, bx);
#pragma unroll
        for (int k = 0; k < KXP; ++k) *reinterpret_cast<BPiece*>(pex + (t * KXP + k) * kPieceBytes) = bx[k];
        if (p.use_viewdirs) {
          float vdir[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) vdir[c] = in[t][7 + c];
          BPiece bd[KDP];
          encode_pieces<BF16, LD, KDP>(vdir, p.fd, h, bd);
#pragma unroll
          for (int k = 0; k < KDP; ++k) *reinterpret_cast<BPiece*>(ped + (t * KDP + k) * kPieceBytes) = bd[k];
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        long long pt = tile * PTS_PER_WG + wave * PTS_PER_WAVE + j + t * 32;
        if (pt >= p.n_points) pt = p.n_points - 1;
        BPiece bx[KXP];
        gather_pieces<BF16, LX, KXP>(p.enc + pt * p.enc_ld, h, bx);
        BPiece bd[KDP];
        if (p.use_viewdirs) gather_pieces<BF16, LD, KDP>(p.enc + pt * p.enc_ld + (3 + 6 * LX), h, bd);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < KXP; ++k) *reinterpret_cast<BPiece*>(pex + (t * KXP + k) * kPieceBytes) = bx[k];
        if (p.use_viewdirs) {
#pragma unroll
          for (int k = 0; k < KDP; ++k) *reinterpret_cast<BPiece*>(ped + (t * KDP + k) * kPieceBytes) = bd[k];
        }
      }
    }
    auto pe_xyz = [&](int t, int k) { return *reinterpret_cast<const BPiece*>(pex + (t * KXP + k) * kPieceBytes); };
    auto no_pe = [&](int, int) { return BPiece{}; };
    // training forward: where this wave's 32-point tile t keeps its saved pieces / mask words
    // this wave's saved-activation record(s) and mask words of the tile: wave-uniform bases, lane offset = lane * 16
    const char* act_tile[PT];
    const char* mask_tile_base[PT];
    if constexpr (SAVE) {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const long long tile32 = (tile * WAVES + wave) * PT + t;
        act_tile[t] = uniform_ptr(p.act + tile32 * p.act_pieces * kPieceBytes);
        mask_tile_base[t] = uniform_ptr(p.masks + tile32 * p.mask_words * kPieceBytes);
      }
    }
    auto save_piece = [&](int t, int slot, const BPiece& v) {
      store16_uniform(act_tile[t] + static_cast<long long>(slot) * kPieceBytes, pipe.lane16, v);
    };
    // SAVE == 2: pieces `slot` (even) and `slot + 1` as ONE 1 KiB unit of 8 + 8 bytes per lane (e4m3)
    auto save_pair = [&](int t, int slot, const BPiece& lo, const BPiece& hi) {
      if constexpr (SAVE == 2 && BF16 == 1) {
        unsigned w[4];
        piece_to_8bit<false>(lo, 1.0f, w[0], w[1]);
        piece_to_8bit<false>(hi, 1.0f, w[2], w[3]);
        store16_uniform(act_tile[t] + static_cast<long long>(slot >> 1) * kPieceBytes, pipe.lane16, make_uint4(w[0], w[1], w[2], w[3]));
      }
    };
    // (the piece arrays are passed by reference to their array type and indexed with compile-time constants only: a
    // decayed pointer sends the whole register-resident activation set to scratch memory in the fp32 instances)
    auto save_pieces = [&](auto nt_c, int t, int slot0, const auto& pieces) {
#ifndef DN_EXP_NOSAVE
      if constexpr (SAVE == 2) {
        constexpr int nt = decltype(nt_c)::value;
        static_assert(P::PPT == 2, "8-bit saved tensors pair the two pieces of a 32-row tile");
        save_pair(t, slot0 + nt * 2, pieces[nt * 2], pieces[nt * 2 + 1]);
      } else if constexpr (SAVE) {
        constexpr int nt = decltype(nt_c)::value;
        static_for<P::PPT>([&](auto s_c) {
          constexpr int s2 = decltype(s_c)::value;
          save_piece(t, slot0 + nt * P::PPT + s2, pieces[nt * P::PPT + s2]);
        });
      }
#endif
    };
    unsigned maskw[PT][4];
    auto mask_clear = [&]() {
#pragma unroll
      for (int t = 0; t < PT; ++t) { maskw[t][0] = 0u; maskw[t][1] = 0u; maskw[t][2] = 0u; maskw[t][3] = 0u; }
    };
    // ReLU mask bits of output tile nt (layout: relu_mask_bit).  16-bit modes read them off the packed ReLU outputs
    // (non-zero <=> pre-activation > 0 in the arithmetic the kernel actually ran): one v_pk_min_u16 + one shift-or per
    // dword instead of compare + select + or per element with sixteen bit constants held in VGPRs.
    auto mask_tile = [&](auto nt_c, int t, const f32x16& acc, const auto& pieces) {
      if constexpr (SAVE) {
        constexpr int nt = decltype(nt_c)::value;
        if constexpr (BF16) {
          typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
          static_for<2>([&](auto s_c) {
            constexpr int s2 = decltype(s_c)::value;
            const u32x4 m = __builtin_bit_cast(u32x4, __builtin_elementwise_min(__builtin_bit_cast(u16x8, pieces[nt * 2 + s2]), one));
#pragma unroll
            for (int d = 0; d < 4; ++d) maskw[t][nt / 2] |= m[d] << (s2 * 4 + d + 8 * (nt & 1));
          });
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            maskw[t][relu_mask_bit(nt, r) / 32] |= (acc[r] > 0.0f) ? (1u << (relu_mask_bit(nt, r) % 32)) : 0u;
        }
      }
    };
    auto mask_store = [&](int word) {
      if constexpr (SAVE) {
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const uint4 v = make_uint4(maskw[t][0], maskw[t][1], maskw[t][2], maskw[t][3]);
          store16_uniform(mask_tile_base[t] + static_cast<long long>(word) * kPieceBytes, pipe.lane16, v);
        }
      }
    };
    if constexpr (SAVE) {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        if constexpr (SAVE == 2) {
          static_assert(KXP % 2 == 0 && KDP % 2 == 0, "encoding panels are whole units");
#pragma unroll
          for (int k = 0; k < KXP; k += 2) save_pair(t, p.slot_xyz + k, pe_xyz(t, k), pe_xyz(t, k + 1));
          if (p.use_viewdirs) {
#pragma unroll
            for (int k = 0; k < KDP; k += 2)
              save_pair(t, p.slot_dir + k, *reinterpret_cast<const BPiece*>(ped + (t * KDP + k) * kPieceBytes),
                        *reinterpret_cast<const BPiece*>(ped + (t * KDP + k + 1) * kPieceBytes));
          }
        } else {
#pragma unroll
        for (int k = 0; k < KXP; ++k) save_piece(t, p.slot_xyz + k, pe_xyz(t, k));
        if (p.use_viewdirs) {
#pragma unroll
          for (int k = 0; k < KDP; ++k)
            save_piece(t, p.slot_dir + k, *reinterpret_cast<const BPiece*>(ped + (t * KDP + k) * kPieceBytes));
        }
        }
      }
    }

    BPiece ba[PT][KH], bb[PT][KH];
    BPiece none[PT][1];
    int bias_tile = 0;
    // One trunk layer: layers_xyz[i] on (cat(x, xyz) when it is a skip layer) -> W, ReLU (models.py:239-246)
    // always_inline: left as a call (hipcc does that for the large fp32 instances) the register-resident activation sets
    // would have to live in scratch memory to be passed by reference
    auto trunk_layer = [&](int i, const BPiece (&bin)[PT][KH], BPiece (&bout)[PT][KH]) __attribute__((always_inline)) {
      auto emit = [&](auto nt_c, auto t_c, const f32x16& acc) {
        constexpr int t = decltype(t_c)::value;
        emit_pieces<BF16, true, decltype(nt_c)::value>(acc, bout[t]);
        save_pieces(nt_c, t, p.slot_trunk0 + i * KH, bout[t]);
        mask_tile(nt_c, t, acc, bout[t]);
      };
      mask_clear();
      if ((p.skip_mask >> i) & 1u)
        run_stage<BF16, PT, NT, KH, KXP, 0>(pipe, bin, pe_xyz, bias_half + bias_tile * 128, emit);
      else
        run_stage<BF16, PT, NT, KH, 0, 0>(pipe, bin, no_pe, bias_half + bias_tile * 128, emit);
      mask_store(i);
      bias_tile += NT;
    };
    // ---- layer1: xyz encoding -> W, no activation (models.py:238) ----
    run_stage<BF16, PT, NT, 0, KXP, 0>(pipe, none, pe_xyz, bias_half, [&](auto nt_c, auto t_c, const f32x16& acc) {
      constexpr int t = decltype(t_c)::value;
      emit_pieces<BF16, false, decltype(nt_c)::value>(acc, ba[t]);
      save_pieces(nt_c, t, p.slot_layer1, ba[t]);
    });
    bias_tile += NT;
    // ---- trunk, two layers per iteration so the activations ping-pong between two register sets ----
    int i = 0;
    for (; i + 1 < p.D - 1; i += 2) {
      trunk_layer(i, ba, bb);
      trunk_layer(i + 1, bb, ba);
    }
    if (i < p.D - 1) {
      trunk_layer(i, ba, bb);
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int k = 0; k < KH; ++k) ba[t][k] = bb[t][k];
    }
    float out4[PT][4];
    if (p.use_viewdirs) {
      // ---- fc_alpha (extra tile, streamed first) + fc_feat with ReLU (models.py:248-249) ----
      constexpr int POS_A = 0;
      run_stage<BF16, PT, 1, KH, 0, POS_A>(pipe, ba, no_pe, bias_half + bias_tile * 128,
                                            [&](auto, auto t_c, const f32x16& acc) {
                                              out4[decltype(t_c)::value][3] = acc[0];  // row 0: lanes 0..31, reg 0
                                            });
      constexpr int POS_F = (POS_A + KH) % kPhasePieces;
      mask_clear();
      run_stage<BF16, PT, NT, KH, 0, POS_F>(pipe, ba, no_pe, bias_half + (bias_tile + 1) * 128,
                                             [&](auto nt_c, auto t_c, const f32x16& acc) {
                                               constexpr int t = decltype(t_c)::value;
                                               emit_pieces<BF16, true, decltype(nt_c)::value>(acc, bb[t]);
                                               save_pieces(nt_c, t, p.slot_feat, bb[t]);
                                               mask_tile(nt_c, t, acc, bb[t]);
                                             });
      mask_store(p.D - 1);
      bias_tile += NT + 1;
      // ---- layers_dir[0] on cat(feat, view) -> W/2, ReLU (models.py:250-252) ----
      constexpr int POS_D = (POS_F + NT * KH) % kPhasePieces;
      BPiece bg[PT][KH / 2];
      auto pe_dir = [&](int t, int k) { return *reinterpret_cast<const BPiece*>(ped + (t * KDP + k) * kPieceBytes); };
      mask_clear();
      run_stage<BF16, PT, NT / 2, KH, KDP, POS_D>(pipe, bb, pe_dir, bias_half + bias_tile * 128,
                                                   [&](auto nt_c, auto t_c, const f32x16& acc) {
                                                     constexpr int t = decltype(t_c)::value;
                                                     emit_pieces<BF16, true, decltype(nt_c)::value>(acc, bg[t]);
                                                     save_pieces(nt_c, t, p.slot_dirout, bg[t]);
                                                     mask_tile(nt_c, t, acc, bg[t]);
                                                   });
      mask_store(p.D);
      bias_tile += NT / 2;
      // ---- fc_rgb (models.py:253) ----
      constexpr int POS_R = (POS_D + (NT / 2) * (KH + KDP)) % kPhasePieces;
      run_stage<BF16, PT, 1, KH / 2, 0, POS_R>(pipe, bg, no_pe, bias_half + bias_tile * 128,
                                                [&](auto, auto t_c, const f32x16& acc) {
                                                  constexpr int t = decltype(t_c)::value;
                                                  out4[t][0] = acc[0]; out4[t][1] = acc[1]; out4[t][2] = acc[2];
                                                });
      static_assert((POS_R + KH / 2) % kPhasePieces == 0, "tail must end on a phase boundary");
    } else {
      // ---- fc_out (models.py:256); the stream is padded to a whole phase after it ----
      run_stage<BF16, PT, 1, KH, 0, 0>(pipe, ba, no_pe, bias_half + bias_tile * 128,
                                        [&](auto, auto t_c, const f32x16& acc) {
                                          constexpr int t = decltype(t_c)::value;
                                          out4[t][0] = acc[0]; out4[t][1] = acc[1]; out4[t][2] = acc[2]; out4[t][3] = acc[3];
                                        });
      if constexpr (KH % kPhasePieces != 0) pipe.template skip<KH % kPhasePieces, kPhasePieces - KH % kPhasePieces>();
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const long long pt = tile * PTS_PER_WG + wave * PTS_PER_WAVE + j + t * 32;
      if (pt < p.n_points && h == 0) {
        f32x4 o;
        o[0] = out4[t][0]; o[1] = out4[t][1]; o[2] = out4[t][2]; o[3] = out4[t][3];
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p.out + pt * 4));  // write-once stream: keep it out of the weight stream's L2
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ---- pack kernel: nn.Linear tensors -> bias tiles + MFMA-A piece stream ---------------------------------

template <int BF16>
__global__ void pack_kernel(NetLayout L, PackPtrs ptrs, char* __restrict__ packed) {  // also used by mlp_train.hip
  using P = Prec<BF16>;
  const int KX = kXyzPanel, KD = round_up(3 + 6 * L.LD, 16);
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
    if (st.transposed) {
      // backward-data stage: A[row][k] = W[k][row0 + row]; k runs over the forward layer's outputs in accumulator
      // order (hidden pieces) and then over one optional "custom" piece (half hh, element e -> k = hh*EPP + e)
      if (rel < st.n_tiles * st.pieces_per_tile) {
        const int ts = rel / st.pieces_per_tile;
        const int k = rel % st.pieces_per_tile;
        const int kh = st.hidden_in / (2 * P::EPP);
        const int row = ts * 32 + i;
        if (row < st.n_real) {
          if (k < kh) {
            const int kout = (k / P::PPT) * 32 + acc_row((k % P::PPT) * P::EPP + e, hh);
            v = ptrs.w[st.src][static_cast<long long>(kout) * st.ld + st.col_hidden0 + row];
          } else {
            const int kc = hh * P::EPP + e;
            if (kc < st.custom_k) {
              const int srcw = st.src2 >= 0 ? st.src2 : st.src;
              v = ptrs.w[srcw][static_cast<long long>(kc) * st.ld + st.col_hidden0 + row];
            }
          }
        }
      }
    } else if (rel < st.n_tiles * st.pieces_per_tile) {
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
    reinterpret_cast<typename P::Elem*>(wout)[idx] = static_cast<typename P::Elem>(v);
  }
}

template <int W, int LX, int LD, int BF16, int PT, int SAVE = 0>
static int launch_forward(FwdParams p, hipStream_t stream) {
  auto kern = mlp_forward_kernel<W, LX, LD, BF16, PT, SAVE>;
  constexpr int WAVES = waves_of<BF16, PT>();
  constexpr int PTS_PER_WG = WAVES * 32 * PT;
  constexpr int KXP = kXyzPanel / (2 * Prec<BF16>::EPP);
  p.n_tiles = (p.n_points + PTS_PER_WG - 1) / PTS_PER_WG;
  constexpr int KDP = round_up(3 + 6 * LD, 16) / (2 * Prec<BF16>::EPP);
  const size_t lds = kRingBytes + p.bias_bytes + WAVES * (kInRows * 32 * PT * sizeof(float) + PT * (KXP + KDP) * kPieceBytes);
  if (int rc = ensure_big_lds(reinterpret_cast<const void*>(kern))) return rc;
  const int cus = device_cus();
  const long long grid = p.n_tiles < cus ? p.n_tiles : cus;
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(WAVES * 64), lds, stream, p);
  return check_launch("mlp_forward");
}

int dispatch_forward(const dn_mlp_desc& d, int precision, FwdParams& p, hipStream_t stream, const CompParams* comp, int* composited) {
  const bool bf = precision == DN_PREC_BF16;
  const bool hf = precision == DN_PREC_F16;
  // bf16 geometry: PT=1 (8 waves x 32 points, two waves per SIMD) measured fastest (1327 vs 1277 TFLOP/s for
  // PT=2 = 4 waves x 64 points, one wave per SIMD); DEXNERF_BF16_PT=2 selects the latter for experiments
  static const int bf16_pt = [] { const char* e = getenv("DEXNERF_BF16_PT"); return (e && atoi(e) == 2) ? 2 : 1; }();
  // bf16 / fp16 inference from rays / points: the 48-points-per-wave geometry (mlp_fused48.hip) when the net fits it;
  // DEXNERF_BF16_GEOM=32 keeps the 32-point kernels (same results up to bf16 accumulation order and the cosine's phase form)
  const char* geom_env = getenv("DEXNERF_BF16_GEOM");   // read per call: tests and probes switch it within one process
  const bool geom48 = !(geom_env && atoi(geom_env) == 32);
  if ((bf || hf) && geom48 && bf16_pt == 1 && p.act == nullptr && p.mode != 2 && p.n_points < (1LL << 31) - 1024 && g48_supported(d, precision))
    return launch_forward48(d, precision, p, p.packed + p.bias_bytes + static_cast<size_t>(p.total_pieces) * kPieceBytes, stream, comp, composited);
  if (p.act != nullptr && hf) { set_error("mlp_forward(train): fp16 is a render-only mode"); return DN_E_UNSUPPORTED; }
  if (p.act != nullptr && p.save8) {   // training forward with 8-bit saved units: the 48-point geometry (mlp_fused48.hip, SAVE = 2)
    if (!bf || p.mode == 2 || !g48_train_supported(d) || p.n_points >= (1LL << 31) - 1024) {
      set_error("mlp_forward(train, 8-bit saved tensors): bf16 arithmetic, rays / points input, W in {128, 256}, L_xyz = 10, a depth the 48-point kernel holds in LDS");
      return DN_E_UNSUPPORTED;
    }
    return launch_forward48(d, precision, p, p.packed + p.bias_bytes + static_cast<size_t>(p.total_pieces) * kPieceBytes, stream);
  }
  if (p.act != nullptr) {  // training forward: LX=10 nets, PT=1
    if (d.num_encoding_fn_xyz == 10 && d.hidden_size == 256)
      return bf ? launch_forward<256, 10, 4, true, 1, true>(p, stream) : launch_forward<256, 10, 4, false, 1, true>(p, stream);
    if (d.num_encoding_fn_xyz == 10 && d.hidden_size == 128)
      return bf ? launch_forward<128, 10, 4, true, 1, true>(p, stream) : launch_forward<128, 10, 4, false, 1, true>(p, stream);
    set_error("mlp_forward(train): no kernel instance for W=%d L_xyz=%d", d.hidden_size, d.num_encoding_fn_xyz);
    return DN_E_UNSUPPORTED;
  }
  if (hf) {  // fp16 render kernels: the L_xyz = 10 nets
    if (d.num_encoding_fn_xyz == 10 && d.hidden_size == 256) return launch_forward<256, 10, 4, 2, 1>(p, stream);
    if (d.num_encoding_fn_xyz == 10 && d.hidden_size == 128) return launch_forward<128, 10, 4, 2, 1>(p, stream);
    set_error("mlp_forward(fp16): no kernel instance for W=%d L_xyz=%d", d.hidden_size, d.num_encoding_fn_xyz);
    return DN_E_UNSUPPORTED;
  }
#define DN_CASE(W_, LX_)                                                                     \
  if (d.hidden_size == W_ && d.num_encoding_fn_xyz == LX_)                                   \
    return bf ? launch_forward<W_, LX_, 4, true, 1>(p, stream) : launch_forward<W_, LX_, 4, false, 1>(p, stream);
  if (bf && bf16_pt == 2 && d.hidden_size == 256 && d.num_encoding_fn_xyz == 10)
    return launch_forward<256, 10, 4, true, 2>(p, stream);  // experimental 4-wave x 64-point geometry
  DN_CASE(256, 10)
  DN_CASE(128, 10)
  DN_CASE(256, 6)
  DN_CASE(128, 6)
#undef DN_CASE
  set_error("mlp_forward: no kernel instance for W=%d L_xyz=%d", d.hidden_size, d.num_encoding_fn_xyz);
  return DN_E_UNSUPPORTED;
}

void fill_freqs(float* f, int num_fns, int log_sampling);  // rays_sampling.hip

int setup_params(const dn_mlp_desc* desc, int precision, const void* packed, FwdParams* p) {
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

int launch_pack(const NetLayout& L, const PackPtrs& ptrs, void* packed, int precision, hipStream_t stream) {
  if (precision == DN_PREC_BF16)
    hipLaunchKernelGGL(pack_kernel<1>, dim3(512), dim3(256), 0, stream, L, ptrs, static_cast<char*>(packed));
  else if (precision == DN_PREC_F16)
    hipLaunchKernelGGL(pack_kernel<2>, dim3(512), dim3(256), 0, stream, L, ptrs, static_cast<char*>(packed));
  else
    hipLaunchKernelGGL(pack_kernel<0>, dim3(512), dim3(256), 0, stream, L, ptrs, static_cast<char*>(packed));
  return check_launch("mlp_pack");
}

}  // namespace dn

using namespace dn;

extern "C" size_t dn_mlp_packed_bytes(const dn_mlp_desc* desc, int precision) {
  if (validate_desc(desc, precision)) return 0;
  NetLayout L;
  build_layout(*desc, precision, &L);
  return static_cast<size_t>(L.bias_bytes) + static_cast<size_t>(L.total_pieces) * kPieceBytes +
         (g48_supported(*desc, precision) ? g48_region_bytes(*desc) : 0);
}

extern "C" int dn_fp16_range_guard(const dn_mlp_desc* desc) {
  if (validate_desc(desc, DN_PREC_F16)) return 0;
  return (std::getenv("DEXNERF_G48_RUNTIME_SHAPE") == nullptr && std::getenv("DEXNERF_BF16_GEOM") == nullptr && g48_range_guard_complete(*desc)) ? 1 : 0;
}

extern "C" int dn_mlp_pack(const dn_mlp_desc* desc, int precision, const float* const* h_weights,
                           const float* const* h_biases, void* packed, dn_stream_t stream) {
  return dn_mlp_pack_parts(desc, precision, h_weights, h_biases, packed, DN_PACK_ALL, stream);
}

extern "C" int dn_mlp_pack_parts(const dn_mlp_desc* desc, int precision, const float* const* h_weights,
                                 const float* const* h_biases, void* packed, int parts, dn_stream_t stream) {
  int rc = validate_desc(desc, precision);
  if (rc) return rc;
  DN_REQUIRE(h_weights && h_biases && packed, "dn_mlp_pack: NULL pointer");
  DN_REQUIRE(parts != 0 && (parts & ~DN_PACK_ALL) == 0, "dn_mlp_pack_parts: parts must be a non-empty mask of DN_PACK_CORE | DN_PACK_G48");
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
  if (parts & DN_PACK_CORE) rc = launch_pack(L, ptrs, packed, precision, as_stream(stream));
  if (rc == 0 && (parts & DN_PACK_G48) && g48_supported(*desc, precision))
    rc = launch_pack48(*desc, precision, ptrs, static_cast<char*>(packed) + L.bias_bytes + static_cast<size_t>(L.total_pieces) * kPieceBytes,
                       as_stream(stream));
  return rc;
}

extern "C" int dn_run_network(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts,
                              const float* viewdirs, const float* rays, int ray_stride, const float* z_vals,
                              int64_t n_rays, int samples_per_ray, float* out, dn_stream_t stream) {
  return dn::run_network_flagged(desc, precision, packed, pts, viewdirs, rays, ray_stride, z_vals, n_rays, samples_per_ray, out,
                                 nullptr, stream);
}

// dn_run_network + the fp16 range flag of the 48-point kernel (FwdParams::range_flag; ignored by every other kernel)
int dn::run_network_flagged(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts,
                            const float* viewdirs, const float* rays, int ray_stride, const float* z_vals,
                            int64_t n_rays, int samples_per_ray, float* out, unsigned* range_flag, dn_stream_t stream,
                            const CompParams* comp, int* composited) {
  if (composited) *composited = 0;
  FwdParams p;
  int rc = setup_params(desc, precision, packed, &p);
  if (rc) return rc;
  if (n_rays == 0) return 0;
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
  p.range_flag = range_flag;
  if (p.n_points == 0) return 0;
  return dispatch_forward(*desc, precision, p, as_stream(stream), comp, composited);
}

extern "C" int dn_mlp_forward_encoded(const dn_mlp_desc* desc, int precision, const void* packed, const float* x,
                                      int64_t n_rows, float* out, dn_stream_t stream) {
  FwdParams p;
  int rc = setup_params(desc, precision, packed, &p);
  if (rc) return rc;
  if (n_rows == 0) return 0;
  DN_REQUIRE(packed && x && out && n_rows >= 0, "dn_mlp_forward_encoded: bad arguments");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dn_mlp_forward_encoded: out must be 16-byte aligned");
  p.mode = 2; p.enc = x;
  p.enc_ld = (3 + 6 * desc->num_encoding_fn_xyz) + (desc->use_viewdirs ? 3 + 6 * desc->num_encoding_fn_dir : 0);
  p.n_points = n_rows; p.S = 1; p.out = out;
  if (n_rows == 0) return 0;
  return dispatch_forward(*desc, precision, p, as_stream(stream));
}
