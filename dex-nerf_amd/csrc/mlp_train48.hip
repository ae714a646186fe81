// Backward-data chain of the fused network in the 48-points-per-wave geometry (the training mode with 8-bit saved tensors,
// DN_PREC_BF16_S8; reference: loss.backward() through FlexibleNeRFModel.forward, nerf/models.py:233-256,
// train_dexnerf_rgb.py:278).  The forward of this mode is mlp_forward48_kernel<..., SAVE = 2> (mlp_fused48.hip).
//
// Same formulation as mlp_backward_kernel (mlp_train.hip) - dX^T[K x points] = W^T[K x N] . dY^T[N x points] on the
// TRANSPOSED weight stream, the masked gradient tile of one stage being the B operand of the next - on
// v_mfma_f32_16x16x32_bf16: every A fragment read from LDS feeds three MFMAs (three 16-point groups per wave, 384 points per
// pass of the stream instead of 256), the explicit LDS read pipeline and the asymmetric weight fetch of the inference
// kernel (mlp_stage48.h).  The ReLU mask words come from the 48-point forward lane for lane: accumulator register r of tile nt
// of a backward stage is the same (feature, point) as in the forward stage whose output it differentiates, so no bit
// transpose is needed (mlp_geo48.h).  Every masked dL/d(pre-activation) is stored once, as e5m2 of (gradient x scale), in the
// s8-48 unit layout the weight-gradient kernel contracts (mlp_train.hip).
#include "mlp_stage48.h"

namespace dn {

struct Bwd48Params {
  const char* packed;      // backward (transposed) piece stream of build_backward_layout48, no bias region
  int total_pieces;
  int D, use_viewdirs;
  const float* g_out;      // (P, 4) dL/d[r, g, b, sigma] of the raw radiance field
  const char* masks;       // mask words of the training forward
  int mask_stages;
  long long n_points;
  int n_tiles;             // 384-point workgroup tiles
  char* grads;             // saved gradients, s8-48 units
  int grad_units;
  int gslot_dirout, gslot_feat, gslot_trunk0, gslot_layer1, gslot_out;
  float scale;             // the power of two the gradients are multiplied by before they are rounded to e5m2 (auto_scale: unused)
  int auto_scale;          // pick the scale from the largest |upstream gradient|: the maximum of the n_partials words at `partials`
  const unsigned* partials;   // (absmax_kernel's 24 in the record, or the compositing backward's one per workgroup: composite.hip)
  int n_partials;
  unsigned* block;         // the 256-byte record behind the saved gradients (kS8Block*): statistics + the scale that was used
  // (the saturation bounds +-57344 / scale are formed once at the top of the kernel, as TWO values - a bound negated at its
  // uses has CodeGenPrepare sink a copy of the free fneg to each of the thousands of uses in the unrolled tile pass: ten minutes)
};

constexpr int kBwd48WaveLds = 8 * kPieceBytes;   // per wave: 2 output-gradient slots (1 KiB) + 3 mask-word slots (2 KiB)

// masked gradient tile -> elements of the next stage's B piece + (every fourth tile) one saved unit
//   mw: this group's mask dwords (lo: tiles 0-7, hi: tiles 8-15), bit layout of mlp_geo48.h
template <int NT, class BO>
__device__ __forceinline__ void emit_grad48(const f32x4& acc, unsigned mw_lo, unsigned mw_hi, BO& bo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w = __builtin_bit_cast(u32x4, bo[NT / 2]);
  const unsigned word = (NT >> 3) ? mw_hi : mw_lo;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    // (no clamp: the saved e5m2 byte is formed straight from the 16-bit pair by a conversion that saturates - MODE.FP16_OVFL, set at
    // the top of the kernel; rounds 2-3 spent a v_med3_f32 per element here)
    const f32x2 f = {acc[2 * d], acc[2 * d + 1]};
    const unsigned pair = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
    // mask bits of registers 2d / 2d+1 sit at position p and p + 16: (bits & 0x00010001) = a 0 / 1 factor per 16-bit half, applied
    // with ONE packed 16-bit multiply (x * 1 = x, x * 0 = 0 on the bit pattern; as a multiply by 0xFFFF and an AND it was two)
    const unsigned bits = (word >> ((NT & 7) * 2 + d)) & 0x00010001u;
    unsigned masked;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(masked) : "v"(pair), "v"(bits));
    w[(NT & 1) * 2 + d] = masked;
  }
  bo[NT / 2] = __builtin_bit_cast(bf16x8, w);
}

// W: hidden width; DC > 0: depth fixed at compile time (with VIEWC the view-direction branch) - the tile pass is straight-line
// code, one settle at its end (mlp_fused48.hip); DC = 0: run-time depth / branch, one settle per stage.
// PTC: point groups per wave - 3 (48 points, 384 per workgroup tile), or 2 for the small launches the forward ran on 256-point
// tiles (mlp_fused48_kernel.h SAVE = 3; mlp_geo48.h g48_train_groups): the mask words are per wave tile, so the two must agree.
template <int W, int DC = 0, int VIEWC = 0, int PTC = 3>
__global__ __launch_bounds__(kG48Waves * 64, 2) void mlp_backward48_kernel(Bwd48Params p) {
  static_assert(PTC == 3 || (PTC == 2 && DC > 0), "two point groups per wave: fixed-shape instances");
  constexpr int F = 1;
  constexpr bool FIXED = DC > 0;
  constexpr bool ST = !FIXED;
  constexpr int PH = kPhasePieces;   // (the two-phase barrier form waits with vmcnt(0): with this kernel's stores in the queue that would be for HBM)
  using BP8 = bf16x8;
  constexpr int PT = PTC;
  constexpr int NT = W / 16;
  constexpr int KH = W / 32;
  constexpr int KHU = KH / 2;
  constexpr int WAVES = kG48Waves;
  constexpr int PPW = 16 * PT;
  constexpr int PPG = kG48Waves * PPW;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // per-wave staging: the output-gradient rows of the NEXT tile and the mask words two stages ahead arrive by LDS-DMA; no
  // VGPR-destination global load exists in the tile loop (mlp_train.hip)
  char* wbuf = smem + kRingBytes + wave * kBwd48WaveLds;
  const unsigned wbuf_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)wbuf));
  const int n_masks = FIXED ? (DC - 1) + (VIEWC ? 2 : 0) : p.mask_stages;   // stage q < n_masks applies mask n_masks-1-q; the last stage none
  // The e5m2 scale of this launch: the caller's power of two, or (auto_scale) the one that puts the largest upstream gradient at
  // 2^12 - 16x under e5m2's largest value for what the chain may amplify, 2^28 above its smallest subnormal.  Every wave forms the
  // same number from the same word; one thread records it for the weight-gradient kernel and the statistics reader.
  float scale = p.scale;
  if (p.auto_scale) {
    float gmax = 0.0f;   // every wave reduces the partial maxima itself (a few KiB out of L2): lanes stride the words, then a wave maximum
    for (int i = lane; i < p.n_partials; i += 64) gmax = fmaxf(gmax, __uint_as_float(p.partials[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
    scale = (gmax > 0.0f && gmax < 3.0e38f) ? exp2f(fminf(fmaxf(12.0f - ceilf(log2f(gmax)), -100.0f), 100.0f)) : 65536.0f;
  }
  // (wave-uniform numbers pinned to scalar registers: formed by vector instructions, they would each hold a VGPR for the whole
  // kernel - this kernel has none to spare)
  auto uniform_f = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
  scale = uniform_f(scale);
  const float inv_scale = uniform_f(1.0f / scale);
  // the e5m2 conversions of the saved gradients saturate at +-57344 (MODE.FP16_OVFL, bit 23: an overflow becomes the largest finite
  // value instead of inf - scripts/micro/cvt_sat_probe.hip; nothing else this bf16 kernel runs reads the bit): the gradient that flows
  // down the chain is not clamped, only its saved byte is, and the statistics count those bytes as before
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  if (blockIdx.x == 0 && threadIdx.x == 0) p.block[kS8BlockScale] = __float_as_uint(scale);
  // (the statistics words of the record are zeroed here for the weight-gradient kernel - the next launch on the stream - which
  // counts saturated / floor-level stores while it reads the gradients anyway)
  if (blockIdx.x == 0 && threadIdx.x < 4 * kS8BlockReplicas) p.block[kS8BlockStats + threadIdx.x] = 0u;
  const int n_points = static_cast<int>(p.n_points);   // (launches of >= 2^31 - 1024 points are refused by the host side)

  auto issue_gout = [&](int tile, int slot) {
    int pt = tile * PPG + wave * PPW + lane;   // lanes 48..63 stage rows nobody reads (inside the slot's 1 KiB)
    if (pt >= n_points) pt = n_points - 1;
    dma16_lanes(p.g_out + static_cast<long long>(pt) * 4, wbuf_addr + slot * kPieceBytes);
  };
  auto issue_mask = [&](int tile, int q) {  // both mask words of stage q of `tile` -> slot q % 3
    const char* src = p.masks + ((static_cast<long long>(tile) * WAVES + wave) * p.mask_stages + (n_masks - 1 - q)) * (2 * kPieceBytes) + lane * 16;
    const unsigned dst = wbuf_addr + 2 * kPieceBytes + (q % 3) * (2 * kPieceBytes);
    dma16_lanes(src, dst);
    dma16_lanes(src + kPieceBytes, dst + kPieceBytes);
  };

  Pipe48<WAVES> pipe;
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
  if constexpr (kG48LeaderDma && kG48AsmReads) {   // phases 0 and 1 landed is all the first barrier period needs (mlp_stage48.h g48_prologue_wait)
    g48_prologue_wait(wave);
    __builtin_amdgcn_s_barrier();
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  pipe.slot_nxt = 0;
  pipe.rda_cur = pipe.ring_addr + lane * 16;
  pipe.slot_cur_base = pipe.ring_addr;       // phase 0 lives in slot 0: phase_begin() of phase 0 turns this into rda_cur
  static_for<kPrefetch>([&](auto e_c) { pipe.template prologue_read<decltype(e_c)::value>(); });

  int g_slot = 0;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
    // this wave's three point groups' saved-gradient bases (wave-uniform; every store adds a small offset)
    const char* grad_grp[PT];
    {
      const long long wt = static_cast<long long>(tile) * WAVES + wave;
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const long long G = wt * PT + t;
        grad_grp[t] = uniform_ptr(p.grads + ((G >> 1) * p.grad_units * 2 + (G & 1)) * kPieceBytes);
      }
    }
    // two 16-bit pairs -> four e5m2 bytes of (gradient x scale)
    auto to_e5m2 = [&](unsigned d0, unsigned d1) { return cvt_pairs_8bit<true>(d0, d1, inv_scale); };
    auto save_unit = [&](auto t_c, int slot, const BP8& lo, const BP8& hi) {
      constexpr int t = decltype(t_c)::value;
      const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
      store16_unit48(grad_grp[t], static_cast<unsigned>(slot) * (2 * kPieceBytes), pipe.lane16,
                     make_uint4(to_e5m2(a[0], a[1]), to_e5m2(a[2], a[3]), to_e5m2(b[0], b[1]), to_e5m2(b[2], b[3])));
    };

    // ---- the output gradient of this lane's points: lane group 0 carries it (custom pieces: k = 8 g + e) ----
    float gv[PT][4];
    {
      const int j = lane & 15;
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wbuf + g_slot * kPieceBytes + (t * 16 + j) * 16);
        const int pt = tile * PPG + wave * PPW + t * 16 + j;
        const bool live = (lane < 16) && (pt < n_points);   // padding points (clamped copies in the forward) contribute nothing
#pragma unroll
        for (int c = 0; c < 4; ++c) gv[t][c] = live ? v[c] : 0.0f;
      }
    }
    // Start of stage q: fetch this stage's mask words from their LDS slot, then stage what will be needed two stages on (same
    // tile, or the first two stages / the output gradient of the next tile) into the slot read one stage ago.  The LDS reads
    // are complete (lgkmcnt(0)) before a DMA may overwrite a slot.
    unsigned mw[PT][2];
    auto stage_begin = [&](int q) {
      {
        // (read unconditionally - the slot exists for every q - and select: no branch around an LDS read in the MFMA stream)
        const char* slot = wbuf + 2 * kPieceBytes + (q % 3) * (2 * kPieceBytes) + lane * 16;
        const uint4 w0 = *reinterpret_cast<const uint4*>(slot);
        const uint2 w1 = *reinterpret_cast<const uint2*>(slot + kPieceBytes);
        const unsigned all = (q < n_masks) ? 0u : ~0u;   // the last stage has no mask
        mw[0][0] = w0.x | all; mw[0][1] = w0.y | all; mw[1][0] = w0.z | all; mw[1][1] = w0.w | all;
        if constexpr (PT == 3) { mw[PT - 1][0] = w1.x | all; mw[PT - 1][1] = w1.y | all; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int r = q + 2;
      if (r < n_masks) issue_mask(tile, r);
      else if (r > n_masks && nxt < p.n_tiles) {   // r == n_masks + 1 / + 2: stages 0 / 1 of the next tile
        const int r2 = r - (n_masks + 1);
        if (r2 < n_masks) issue_mask(nxt, r2);
        if (r2 == 0) issue_gout(nxt, g_slot ^ 1);
      }
    };
    auto emit_to = [&](auto nt_c, auto t_c, const f32x4& acc, auto& bout, int gslot) {
      constexpr int nt = decltype(nt_c)::value, t = decltype(t_c)::value;
      emit_grad48<nt>(acc, mw[t][0], mw[t][1], bout[t]);
      if constexpr (nt % 4 == 3) save_unit(t_c, gslot + nt / 4, bout[t][nt / 2 - 1], bout[t][nt / 2]);
    };

    BP8 ba[PT][KH], bb[PT][KH];
    BP8 none[PT][1];
    auto no_pe = [&](int, int) { return BP8{}; };
    int q = 0;  // stage counter of this tile
    // custom K pieces: element e of lane group 0 carries k = e
    auto custom_piece = [&](int t, int first, int count) {
      BP8 c{};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < count) c[e] = static_cast<__bf16>(gv[t][first + e]);
      return c;
    };
    // trunk, i = D-2 .. 0:  d x_i = layers_xyz[i][:, :W]^T d pre_i, masked by relu'(x_i) (x_0 = layer1's output: the last stage has
    // no mask word - stage_begin hands back all ones).  The gradient sets ping-pong between `ba` and `bb`; every trunk stage is a
    // whole number of phases, so they all start at the position the head stages leave (POS).
    auto trunk = [&](auto pos_c, auto depth_c) __attribute__((always_inline)) {
      constexpr int POS = decltype(pos_c)::value;
      constexpr int PAD = (kPhasePieces - POS % kPhasePieces) % kPhasePieces;
      static_assert((NT * KH) % kPhasePieces == 0, "trunk stages must preserve the phase offset");
      auto layer = [&](int i, const BP8 (&bin)[PT][KH], BP8 (&bout)[PT][KH], auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        stage_begin(q++);
        const int gslot = i > 0 ? p.gslot_trunk0 + (i - 1) * KHU : p.gslot_layer1;
        run_stage48<F, NT, KH, 0, POS, LAST, ST, (LAST ? PAD : 0), PH, 0, false>(pipe, bin, no_pe, 0u, 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
          emit_to(nt_c, t_c, acc, bout, gslot);
        });
      };
      constexpr int DD = decltype(depth_c)::value;
      if constexpr (DD > 0) {
        // fixed depth: straight-line code
        static_for<DD - 1>([&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          constexpr int i = DD - 2 - s;
          if constexpr (s % 2 == 0) layer(i, ba, bb, std::integral_constant<bool, i == 0>{});
          else layer(i, bb, ba, std::integral_constant<bool, i == 0>{});
        });
      } else {
        int i = p.D - 2;
        for (; i >= 2; i -= 2) {
          layer(i, ba, bb, std::false_type{});
          layer(i - 1, bb, ba, std::false_type{});
        }
        if (i == 1) {
          layer(1, ba, bb, std::false_type{});
          layer(0, bb, ba, std::true_type{});
        } else {
          layer(0, ba, bb, std::true_type{});
        }
      }
      if constexpr (PAD != 0) pipe.template skip<POS + NT * KH, PAD, PH, false>();   // (settles at its end)
      else pipe.template settle<false>();
    };
    auto with_viewdirs = [&](auto depth_c) __attribute__((always_inline)) {
      BP8 crgb[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) crgb[t] = custom_piece(t, 0, 3);
      static_for<PT>([&](auto t_c) { save_unit(t_c, p.gslot_out, crgb[decltype(t_c)::value], custom_piece(decltype(t_c)::value, 3, 1)); });   // for dW(fc_rgb), dW(fc_alpha)
      // ---- d g = fc_rgb^T d rgb, masked by relu'(layers_dir.0 out) ----
      stage_begin(q++);
      run_stage48<F, NT / 2, 0, 1, 0, false, ST, 0, PH, 0, false>(pipe, none, [&](int t, int) { return crgb[t]; }, 0u, 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
        emit_to(nt_c, t_c, acc, ba, p.gslot_dirout);
      });
      // ---- d feat = layers_dir.0[:, :W]^T d dirpre, masked by relu'(fc_feat out) ----
      constexpr int P1 = (NT / 2) % kPhasePieces;
      stage_begin(q++);
      // (the stage's K is the first KH / 2 pieces of `ba`)
      run_stage48<F, NT, KH / 2, 0, P1, false, ST, 0, PH, 0, false>(pipe, ba, no_pe, 0u, 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
        emit_to(nt_c, t_c, acc, bb, p.gslot_feat);
      });
      // ---- d h = fc_feat^T d featpre + fc_alpha^T d alpha, masked by relu'(layers_xyz[D-2] out) ----
      constexpr int P2 = (P1 + NT * (KH / 2)) % kPhasePieces;
      stage_begin(q++);
      // (the d sigma pieces are formed here, from the one live register per group: as arrays built at the top of the tile they
      // are twelve registers - nine of them zeros - carried through two stages)
      BP8 calpha[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) calpha[t] = custom_piece(t, 3, 1);
      run_stage48<F, NT, KH, 1, P2, false, ST, 0, PH, 0, false>(pipe, bb, [&](int t, int) { return calpha[t]; }, 0u, 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
        emit_to(nt_c, t_c, acc, ba, p.gslot_trunk0 + (p.D - 2) * KHU);
      });
      constexpr int PV = (P2 + NT * (KH + 1)) % kPhasePieces;
      trunk(std::integral_constant<int, PV>{}, depth_c);
    };
    auto without_viewdirs = [&](auto depth_c) __attribute__((always_inline)) {
      // ---- d h = fc_out^T d out, masked by relu'(layers_xyz[D-2] out) ----
      BP8 cout[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) cout[t] = custom_piece(t, 0, 4);
      static_for<PT>([&](auto t_c) { save_unit(t_c, p.gslot_out, cout[decltype(t_c)::value], BP8{}); });   // for dW(fc_out)
      stage_begin(q++);
      run_stage48<F, NT, 0, 1, 0, false, ST, 0, PH, 0, false>(pipe, none, [&](int t, int) { return cout[t]; }, 0u, 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
        emit_to(nt_c, t_c, acc, ba, p.gslot_trunk0 + (p.D - 2) * KHU);
      });
      trunk(std::integral_constant<int, NT % kPhasePieces>{}, depth_c);
    };
    if constexpr (FIXED) {
      if constexpr (VIEWC) with_viewdirs(std::integral_constant<int, DC>{});
      else without_viewdirs(std::integral_constant<int, DC>{});
    } else {
      if (p.use_viewdirs) with_viewdirs(std::integral_constant<int, 0>{});
      else without_viewdirs(std::integral_constant<int, 0>{});
    }
    g_slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// largest finite |x| of the upstream gradient: workgroup b (of kS8BlockPartialCount, 1024 threads) writes ITS maximum to
// block[kS8BlockPartials + b] - every word is written, nothing needs zeroing first.  x = n4 rows of four floats (16-byte aligned): one
// row per thread and step, four steps in flight (one wave per workgroup reading scalars took 0.5 ms for the 786,432 rows of a D8/W256
// step - a sixth of the whole training step).
__global__ __launch_bounds__(1024) void absmax_kernel(const float4* __restrict__ x, long long n4, unsigned* __restrict__ block) {
  __shared__ float part[16];
  float m = 0.0f;
  auto fold = [&](const float4& v) {
    const float a = fabsf(v.x), b = fabsf(v.y), c = fabsf(v.z), d = fabsf(v.w);
    m = (a < 3.0e38f) ? fmaxf(m, a) : m;   // (a non-finite upstream gradient does not set the scale)
    m = (b < 3.0e38f) ? fmaxf(m, b) : m;
    m = (c < 3.0e38f) ? fmaxf(m, c) : m;
    m = (d < 3.0e38f) ? fmaxf(m, d) : m;
  };
  const long long stride = static_cast<long long>(gridDim.x) * 1024;
  long long i = static_cast<long long>(blockIdx.x) * 1024 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
    fold(v0); fold(v1); fold(v2); fold(v3);
  }
  for (; i < n4; i += stride) fold(x[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = part[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r = fmaxf(r, part[w]);
    block[kS8BlockPartials + blockIdx.x] = __float_as_uint(r);
  }
}

// ---- pack: nn.Linear tensors -> the transposed 16 x 32 A pieces of build_backward_layout48 ------------------------------
__device__ __forceinline__ void pack48_backward_body(const NetLayout& L, const PackPtrs& ptrs, char* __restrict__ out);

__global__ void pack48_backward_kernel(NetLayout L, PackPtrs ptrs, char* __restrict__ out) { pack48_backward_body(L, ptrs, out); }

// two networks of one architecture in one launch (blockIdx.y picks the net)
__global__ void pack48_backward_pair_kernel(NetLayout L, PackPtrs ptrs_a, PackPtrs ptrs_b, char* __restrict__ out_a, char* __restrict__ out_b) {
  if (blockIdx.y == 0) pack48_backward_body(L, ptrs_a, out_a);
  else pack48_backward_body(L, ptrs_b, out_b);
}

__device__ __forceinline__ void pack48_backward_body(const NetLayout& L, const PackPtrs& ptrs, char* __restrict__ out) {
  __bf16* wout = reinterpret_cast<__bf16*>(out);
  const long long n_elems = static_cast<long long>(L.total_pieces) * 64 * 8;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < n_elems;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int e = static_cast<int>(idx % 8);
    const int lane = static_cast<int>((idx / 8) % 64);
    const int piece = static_cast<int>(idx / 512);
    const int i = lane & 15, g = lane >> 4;
    float v = 0.0f;
    int s = 0;
    while (s + 1 < L.n_stages && L.st[s + 1].piece0 <= piece) ++s;
    const StageDesc& st = L.st[s];
    const int rel = piece - st.piece0;
    if (rel < st.n_tiles * st.pieces_per_tile) {
      // A[row][k] = W[k][row]: k runs over the forward layer's outputs in accumulator order (hidden pieces), then over one
      // optional custom piece (lane group g, element e -> k = 8 g + e)
      const int ts = rel / st.pieces_per_tile;
      const int k = rel % st.pieces_per_tile;
      const int kh = st.hidden_in / 32;
      const int row = ts * 16 + i;
      if (row < st.n_real) {
        if (k < kh) {
          v = ptrs.w[st.src][static_cast<long long>(g48_hidden_col(k, g, e)) * st.ld + row];
        } else {
          const int kc = 8 * g + e;
          if (kc < st.custom_k) v = ptrs.w[st.src2 >= 0 ? st.src2 : st.src][static_cast<long long>(kc) * st.ld + row];
        }
      }
    }
    wout[idx] = static_cast<__bf16>(v);
  }
}

// ---- s8-48 units -> plain (P, width) fp32 rows (tests and probes: dn_mlp_unpack with DN_PREC_BF16_S8) --------------------
// which 0: e4m3 activations, 1: e5m2 gradients (x inv_scale).  kind 0: hidden vector, 1 / 2: xyz / view-direction panel,
// 3: the custom unit (piece 0 elements 0-3 -> columns 0-3, piece 1 elements 0-3 -> columns 4-7: [d rgb . | d alpha . . .]).
__global__ void unpack48_kernel(const char* __restrict__ buf, int units_per_group, int slot0, int n_units, int kind, int L,
                                long long n_points, int which, const unsigned* __restrict__ block, float* __restrict__ out, int ld_out, int col0) {
  const float inv_scale = (which == 1) ? 1.0f / __uint_as_float(block[kS8BlockScale]) : 1.0f;   // the scale the backward recorded
  const long long groups = (n_points + 15) / 16;
  const long long total = groups * n_units * 64 * 16;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(idx % 16);
    const int lane = static_cast<int>((idx / 16) % 64);
    const int u = static_cast<int>((idx / 1024) % n_units);
    const long long G = idx / (1024LL * n_units);
    const int g = lane >> 4, j = (lane & 15) ^ ((g & 1) << 3);   // (rows of the odd lane groups are swizzled: store16_unit48)
    const long long pt = G * 16 + j;
    if (pt >= n_points) continue;
    int col;
    if (kind == 0) col = g48_hidden_col(2 * u + (b >> 3), g, b & 7);
    else if (kind == 1) col = g48_pe_col(1, g, b, L);
    else if (kind == 2) col = g < 2 ? g48_pe_col(2, g + 2 * (b >> 3), b & 7, L) : -1;
    else col = (g == 0 && (b & 7) < 4) ? (b >> 3) * 4 + (b & 7) : -1;
    if (col < 0) continue;
    const char* src = buf + (((G >> 1) * units_per_group + slot0 + u) * 2 + (G & 1)) * kPieceBytes + lane * 16;
    const int word = reinterpret_cast<const int*>(src)[b >> 2];
    float v;
    if (which == 0) v = (b & 3) == 0 ? __builtin_amdgcn_cvt_f32_fp8(word, 0) : (b & 3) == 1 ? __builtin_amdgcn_cvt_f32_fp8(word, 1)
                        : (b & 3) == 2 ? __builtin_amdgcn_cvt_f32_fp8(word, 2) : __builtin_amdgcn_cvt_f32_fp8(word, 3);
    else v = ((b & 3) == 0 ? __builtin_amdgcn_cvt_f32_bf8(word, 0) : (b & 3) == 1 ? __builtin_amdgcn_cvt_f32_bf8(word, 1)
              : (b & 3) == 2 ? __builtin_amdgcn_cvt_f32_bf8(word, 2) : __builtin_amdgcn_cvt_f32_bf8(word, 3)) * inv_scale;
    out[pt * ld_out + col0 + col] = v;
  }
}

int unpack48_entry(const dn_mlp_desc* desc, int which, const void* native, int64_t n_points, int slot, int width, int kind, float* out,
                   int ld_out, int col0, hipStream_t stream) {
  TrainLayout48 t;
  build_train_layout48(*desc, &t);
  const int per_group = which == 0 ? t.act_units : t.grad_units;
  const int n_units = kind == 0 ? width / 64 : 1;
  DN_REQUIRE((kind != 0 || width % 64 == 0) && slot >= 0 && slot + n_units <= per_group, "dn_mlp_unpack (8-bit layout): slot range outside the group's units");
  const int L = kind == 1 ? desc->num_encoding_fn_xyz : desc->num_encoding_fn_dir;
  hipLaunchKernelGGL(unpack48_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const char*>(native), per_group, slot, n_units, kind, L,
                     static_cast<long long>(n_points), which,
                     reinterpret_cast<const unsigned*>(static_cast<const char*>(native) + g48_padded_records(n_points) * 2 * t.grad_units * kPieceBytes),
                     out, ld_out, col0);
  return check_launch("dn_mlp_unpack");
}

int launch_pack48_backward(const dn_mlp_desc& d, const PackPtrs& ptrs, char* packed, hipStream_t stream) {
  NetLayout L;
  build_backward_layout48(d, &L);
  hipLaunchKernelGGL(pack48_backward_kernel, dim3(pack48_blocks(L)), dim3(256), 0, stream, L, ptrs, packed);
  return check_launch("mlp_pack48_backward");
}

int launch_pack48_backward_pair(const dn_mlp_desc& d, const PackPtrs& a, const PackPtrs& b, char* packed_a, char* packed_b, hipStream_t stream) {
  NetLayout L;
  build_backward_layout48(d, &L);
  hipLaunchKernelGGL(pack48_backward_pair_kernel, dim3(pack48_blocks(L), 2), dim3(256), 0, stream, L, a, b, packed_a, packed_b);
  return check_launch("mlp_pack48_backward_pair");
}

int launch_backward48(const dn_mlp_desc& d, Bwd48Params p, hipStream_t stream) {
  NetLayout L;
  build_backward_layout48(d, &L);
  p.total_pieces = L.total_pieces;
  p.n_tiles = static_cast<int>((p.n_points + kG48PointsPerWg - 1) / kG48PointsPerWg);
  const size_t lds = static_cast<size_t>(kRingBytes) + kG48Waves * kBwd48WaveLds;
  const int cus = device_cus();
  int grid = p.n_tiles < cus ? p.n_tiles : cus;
  auto launch = [&](auto kern) -> int {
    if (int rc = ensure_big_lds(reinterpret_cast<const void*>(kern))) return rc;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(kG48Waves * 64), lds, stream, p);
    return check_launch("mlp_backward48");
  };
  // a small launch whose forward ran on 256-point tiles (two point groups per wave: mlp_geo48.h - the same question, the same answer)
  if (g48_two_group_shape(d) && g48_train_groups(p.n_points, cus) == 2) {
    p.n_tiles = static_cast<int>((p.n_points + 255) / 256);
    grid = p.n_tiles < cus ? p.n_tiles : cus;
#ifndef DN_T48_ONLY128
    if (d.hidden_size == 256) return launch(mlp_backward48_kernel<256, 8, 1, 2>);
#endif
    return launch(mlp_backward48_kernel<128, 4, 1, 2>);
  }
  const bool paper = d.hidden_size == 256 && d.num_layers == 8 && d.use_viewdirs;
  const bool shipped = d.hidden_size == 128 && d.num_layers == 4 && d.use_viewdirs;
  const bool fixed_ok = std::getenv("DEXNERF_G48_RUNTIME_SHAPE") == nullptr;
#ifndef DN_T48_ONLY128
  if (paper && fixed_ok) return launch(mlp_backward48_kernel<256, 8, 1>);
#endif
  if (shipped && fixed_ok) return launch(mlp_backward48_kernel<128, 4, 1>);
#ifdef DN_T48_FIXED_ONLY   // developer hook: skip the run-time-shape instances (compile time)
  set_error("mlp_backward48: built with the fixed-shape instances only");
  return DN_E_UNSUPPORTED;
#else
  return d.hidden_size == 256 ? launch(mlp_backward48_kernel<256>) : launch(mlp_backward48_kernel<128>);
#endif
}

}  // namespace dn

using namespace dn;

// The C entry points (dn_mlp_backward_data, dn_mlp_pack_backward, dn_mlp_train_sizes, ...) live in mlp_train.hip and dispatch here
// for DN_PREC_BF16_S8.
int dn::backward48_entry(const dn_mlp_desc* desc, const void* packed_bwd, const float* g_out, const void* masks, int64_t n_points,
                         void* grads, float grad_scale, hipStream_t stream, const unsigned* partials, int n_partials) {
  TrainLayout48 t;
  build_train_layout48(*desc, &t);
  Bwd48Params p{};
  p.packed = static_cast<const char*>(packed_bwd);
  p.D = desc->num_layers;
  p.use_viewdirs = desc->use_viewdirs;
  p.g_out = g_out;
  p.masks = static_cast<const char*>(masks);
  p.mask_stages = t.mask_stages;
  p.n_points = n_points;
  p.grads = static_cast<char*>(grads);
  p.grad_units = t.grad_units;
  p.gslot_dirout = t.gslot_dirout; p.gslot_feat = t.gslot_feat; p.gslot_trunk0 = t.gslot_trunk0; p.gslot_layer1 = t.gslot_layer1;
  p.gslot_out = t.gslot_out;
  // the 256-byte record behind the units (mlp_geo48.h): the largest |upstream gradient| first when the scale is to follow it
  p.block = reinterpret_cast<unsigned*>(static_cast<char*>(grads) + g48_padded_records(n_points) * 2 * t.grad_units * kPieceBytes);
  p.auto_scale = grad_scale == 0.0f ? 1 : 0;
  p.scale = p.auto_scale ? 65536.0f : grad_scale;
  p.partials = partials; p.n_partials = n_partials;
  if (p.auto_scale && partials == nullptr) {
    hipLaunchKernelGGL(absmax_kernel, dim3(kS8BlockPartialCount), dim3(1024), 0, stream, reinterpret_cast<const float4*>(g_out),
                       static_cast<long long>(n_points), p.block);
    if (int rc = check_launch("s8 absmax")) return rc;
    p.partials = p.block + kS8BlockPartials; p.n_partials = kS8BlockPartialCount;
  }
  return launch_backward48(*desc, p, stream);
}
