// The 48-points-per-wave forward kernel (template): included by the translation units that instantiate it - mlp_fused48.hip (the
// W = 256 instances and the host side) and mlp_fused48_w128.hip (the W = 128 instances) - so the two halves of the instance list
// compile in parallel.  Design notes: the comment at the top of mlp_fused48.hip.
#pragma once
#include "mlp_stage48.h"

namespace dn {


// one encoding slot: table entry = (frequency, phase in revolutions, identity weight, sine weight); the coordinate is a
// compile-time element of the lane's rotated point (g48_pe_col).  No per-slot decode and no selects (per-lane compares
// land in SGPR pairs; dozens of them per block spilled scalars into VGPR lanes and activation pieces to scratch).
// One hardware sine per slot: cos(2 pi r) = sin(2 pi (r + 1/4)) (revolutions, as mlp_device.h encode_pieces in bf16 mode).
// The table's frequency is in REVOLUTIONS per unit (f / 2 pi, folded in by the pack kernel: for the reference's default power-of-two
// frequencies (x f) (1 / 2 pi) and x (f / 2 pi) round identically - a scaling by 2^k is exact - so the encodings are the ones of
// rounds 1-3 bit for bit; two multiplies became one).  Only slots 0-2 of a lane group can be identity columns (g48_pe_col: rank 0);
// the others are plain sines (padding: frequency 0, phase 0 -> sin 0 = 0) and skip the two-weight mix: 4 vector instructions per slot
// instead of 7, and the top-of-tile block is 48 slots per lane (profiles/r04_headline_schedule.md: 4 % of the pass).
template <bool MAY_BE_IDENTITY = true>
__device__ __forceinline__ float pe_value(float xc, f32x4 entry) {
  const float rev = __builtin_amdgcn_fractf(xc * entry[0]) + entry[1];
  const float s = __builtin_amdgcn_sinf(rev);
  if constexpr (MAY_BE_IDENTITY) return entry[2] * xc + entry[3] * s;   // weights are 0 / 1: exact
  else return s;
}

// (x, y, z) rotated so that element k is coordinate (k + g) % 3 of the point.  Bitwise selects: written with ?: hipcc
// turns the rotation into a dynamically indexed stack array (scratch loads behind vmcnt waits).
__device__ __forceinline__ void rotate3(const float (&x)[3], int g, float (&xr)[3]) {
  const unsigned r = static_cast<unsigned>(g) % 3u;
  const unsigned m1 = 0u - (r & 1u), m2 = 0u - (r >> 1);
  const unsigned b0 = __float_as_uint(x[0]), b1 = __float_as_uint(x[1]), b2 = __float_as_uint(x[2]);
  xr[0] = __uint_as_float(b0 ^ ((b0 ^ b1) & m1) ^ ((b0 ^ b2) & m2));
  xr[1] = __uint_as_float(b1 ^ ((b1 ^ b2) & m1) ^ ((b1 ^ b0) & m2));
  xr[2] = __uint_as_float(b2 ^ ((b2 ^ b0) & m1) ^ ((b2 ^ b1) & m2));
}

// DC / MASKC / VIEWC > 0: depth, skip mask and view-direction branch fixed at compile time - the whole tile pass is then
// straight-line code: no control-flow merge for the compiler to place copies of in-flight fragments at (Pipe::settle), the
// ring bookkeeping stays in SGPRs, and the epilogue of a stage's last tile overlaps the next stage's first MFMAs like any
// other tile's (measured: the 12 per-stage settles + merges of the run-time form cost 5 % of the launch).  DC = 0: everything
// run-time (p.D, p.skip_mask, p.use_viewdirs), one settle per stage.
// SAVE = 2: the training forward of DN_PREC_BF16_S8 - the same chain also streams every stage's output (and both encodings) as
// e4m3 units in the s8-48 layout and the ReLU mask words (mlp_geo48.h) to HBM: non-temporal scalar-base 16-byte stores
// (mlp_device.h store16_uniform), a unit every fourth output tile per point group.
// OVLP = 1 (the as-shipped 4 x 128 instance, rays + depths as inputs): the xyz encoding of tile t + 1 is computed inside tile t -
// one slot per output tile of the first two trunk stages, in the shadow of their MFMAs - instead of in a block at the top of
// tile t + 1 during which the matrix pipes idle (18 % of this instance's pass: profiles/r02_config_sweep.md).  This instance has
// the registers for it (the W = 256 one does not: 13 carried VGPRs + the encoding table held in 38): the rotated coordinates of
// the next tile's points are read from the input rows after layer1, the pieces are parked in the (by then dead) LDS stash.
// COMP = 1 (fixed-shape render instances, samples per ray dividing the 384-point tile): the kernel composites the rays of a tile
// itself - the raw (rgb, sigma) rows are staged in LDS instead of being written to HBM (16 B per point), and after a workgroup
// barrier one wave per finished ray runs the body of composite_fwd_kernel on them (composite_body.h: the same code, the same bits).
template <int W, int F, int DC = 0, unsigned MASKC = 0, int VIEWC = 0, int SAVE = 0, int OVLP = 0, int COMP = 0>
__global__ __launch_bounds__(kG48Waves * 64, 2) void mlp_forward48_kernel(FwdParams p, G48Params q) {
  // SAVE = 3: the same training forward on TWO point groups per wave (32 points, 256 per workgroup tile) - for launches so small that
  // 384-point tiles leave compute units idle or make a short last round (mlp_geo48.h g48_train_groups: a 1024-ray step of the as-shipped
  // nets is 171 + 342 tiles on 256 units).  Same per-point arithmetic, same saved units (the layout is by 16-point group), mask words
  // per wave tile as before with the third group's words absent; the backward instance of the same width reads them back.
  static_assert(SAVE == 0 || ((SAVE == 2 || SAVE == 3) && F == 1), "saved tensors: the 8-bit layout, bf16 arithmetic");
  static_assert(SAVE != 3 || (OVLP == 0 && COMP == 0 && DC > 0), "two point groups per wave: the fixed-shape training forward");
  static_assert(COMP == 0 || (SAVE == 0 && OVLP == 0 && DC > 0), "in-kernel compositing: a render instance whose xyz stash is free at the end of a pass");
  constexpr bool OVL = OVLP == 1;
  static_assert(!OVL || (DC >= 3 && MASKC == 0u && VIEWC != 0 && SAVE == 0 && W == 128),
                "overlapped encoding: fixed shape, no skip layer (the stash is dead after layer1), two trunk stages of 24 output tiles");
  // OVLP = 2 (the paper network's render instance on rays + depths): tile t + 1's xyz encoding rides in the MFMA gaps of tile t's
  // view-direction stage, one vector instruction at a time (explicit schedule: run_stage48x's hook) - the registers of the dead trunk
  // set are free there, the stash is dead (the view-direction pieces are in registers), and the encoding table comes from LDS one
  // entry per slot, read a block ahead.  Only a workgroup's first tile is encoded at the top.
  constexpr bool OVX = OVLP == 2;
  static_assert(!OVX || (W == 256 && DC > 0 && VIEWC != 0 && SAVE == 0 && COMP == 0), "in-stage encoding: the explicit-schedule W = 256 render instance");
  constexpr int VSETS = OVL ? 3 : 2;           // view-direction row sets: the inputs of tile t + 2 arrive during tile t
  constexpr int IN_ROWS = 7 + 3 * VSETS;
  constexpr bool FIXED = DC > 0;
  // explicit-schedule tile pass (run_stage48x): the paper network's render instances
#ifdef DN_G48_NO_XS
  constexpr bool XS = false;
#else
  constexpr bool XS = FIXED && SAVE == 0 && (OVLP == 0 || OVLP == 2) && COMP == 0 && VIEWC != 0;
#endif
  constexpr bool ST = !FIXED;   // settle at stage ends
#if defined(DN_PIPE_ASM_READS) && defined(DN_PIPE_LEADER_DMA) && !defined(DN_G48_BARRIER_EVERY_PHASE)
  // barrier period in pieces: every second phase where a phase's parity is a compile-time position - the fixed-shape W = 256
  // instance (every stage boundary of D8 / skip 4 falls on an even phase, 74 phases per pass) - every phase elsewhere.
  // (The two-phase form waits with vmcnt(0): with the training forward's stores in the queue that wait would be for HBM.)
  constexpr int PH = (FIXED && W == 256 && SAVE == 0) ? 2 * kPhasePieces : kPhasePieces;
#else
  constexpr int PH = kPhasePieces;
#endif
  using BP8 = typename Prec<F>::BPiece;
  using Elem = typename Prec<F>::Elem;
  constexpr int PT = SAVE == 3 ? 2 : 3;   // point groups per wave
  constexpr int NT = W / 16;
  constexpr int KH = W / 32;
  constexpr int KXP = kG48XyzPieces, KDP = kG48DirPieces;
  constexpr int WAVES = kG48Waves;
  constexpr int PPW = 16 * PT;             // = kG48PointsPerWave but for SAVE = 3
  constexpr int PPG = kG48Waves * PPW;

  // training forward: the 8-bit conversions of the saved units saturate (MODE.FP16_OVFL, bit 23: an e4m3 overflow becomes 448 instead of
  // NaN - scripts/micro/cvt_sat_probe.hip), so the stage outputs need no clamp; nothing else this bf16 instance runs reads the bit
  if constexpr (SAVE != 0) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* bias_lds = smem + kRingBytes;
  const char* tab_lds = bias_lds + q.bias_bytes;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Lane-derived LDS addresses are NOT carried across the trunk (every carried VGPR is a spill at this register budget, and
  // a spill reload waits with vmcnt(0), draining the weight pipeline): each use site rebuilds them from an opaque copy
  // of the thread index, which the optimiser cannot merge with the other sites.
  auto fresh_lane = [&]() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t & 63; };
  auto pex_of = [&](int ln) { return smem + kRingBytes + q.bias_bytes + kG48TableBytes + wave * (PT * KXP * kPieceBytes) + ln * 16; };
  // per-wave input rows of 48 floats: 0-2 origin / point, 3-5 direction, 6 depth, then two sets of 3 view-direction rows
  // (the next tile is staged at the top of this one, when registers are free; the view direction of THIS tile is only
  // consumed near its end, so those rows alternate between two sets)
  float* inbuf = reinterpret_cast<float*>(smem + kRingBytes + q.bias_bytes + kG48TableBytes + WAVES * PT * KXP * kPieceBytes) +
                 wave * (IN_ROWS * PPW);   // wave-uniform
  static_assert(OVL || IN_ROWS == kG48InRows, "g48_lds_bytes sizes the input rows");

  // inputs of a tile by 4-byte LDS-DMA: lane l < 48 stages point l of this wave (mlp_fused.hip issue_inputs)
  // 32-bit point indices throughout (the dispatcher sends launches of >= 2^31 - 1024 points to the 32-point kernel): the
  // 64-bit forms cost a loop-invariant VGPR pair (spilled) and a 64-bit division per lane per tile
  const int n_points = static_cast<int>(p.n_points);
  auto issue_inputs = [&](int tile, int vset) {
    const int lane = fresh_lane();
    int pt = tile * PPG + wave * PPW + lane;
    if (pt >= n_points) pt = n_points - 1;
    auto dma = [&](const float* src, int row) {
      if (lane < PPW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(inbuf + row * PPW), 4, 0, 0);
    };
    if (p.mode == 0) {
      const float* r = p.rays + static_cast<long long>(pt / p.S) * p.ray_stride;
#pragma unroll
      for (int c = 0; c < 6; ++c) dma(r + c, c);
      dma(p.z + pt, 6);
      if (p.use_viewdirs) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dma(r + 8 + c, 7 + 3 * vset + c);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) dma(p.pts + static_cast<long long>(pt) * 3 + c, c);
      if (p.use_viewdirs) {
        const float* v = p.viewdirs + static_cast<long long>(pt / p.S) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dma(v + c, 7 + 3 * vset + c);
      }
    }
  };

  {  // bias rows + encoding tables -> LDS once per workgroup
    const f32x4* gsrc = reinterpret_cast<const f32x4*>(q.base);
    f32x4* l = reinterpret_cast<f32x4*>(bias_lds);
    for (int i = threadIdx.x; i < (q.bias_bytes + kG48TableBytes) / 16; i += WAVES * 64) l[i] = gsrc[i];
  }
  issue_inputs(blockIdx.x, 0);
  // OVL: the same DMAs without control flow (mid-pass a branch is a merge point the compiler may park copies of in-flight weight
  // fragments at): the lane mask is set inside the asm statement, the tile index is clamped by the caller
  const unsigned inbuf_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)inbuf));
  auto issue_inputs_flat = [&](int tile, int set) {
    const int ln = fresh_lane();
    int pt = tile * PPG + wave * PPW + ln;
    pt = pt < n_points ? pt : n_points - 1;
    const float* r = p.rays + static_cast<long long>(pt / p.S) * p.ray_stride;
#pragma unroll
    for (int c = 0; c < 6; ++c) dma4_lanes48(r + c, inbuf_addr + c * (PPW * 4));
    dma4_lanes48(p.z + pt, inbuf_addr + 6 * (PPW * 4));
#pragma unroll
    for (int c = 0; c < 3; ++c) dma4_lanes48(r + 8 + c, inbuf_addr + (7 + c) * (PPW * 4) + set * (3 * PPW * 4));
  };

  Pipe48<WAVES> pipe;
  pipe.ring = ring;
  pipe.ring_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring));
  pipe.lane16 = lane * 16;
  pipe.wsrc = q.base + q.bias_bytes + kG48TableBytes;
  pipe.total_bytes = static_cast<unsigned>(q.total_pieces) * kPieceBytes;
  pipe.q_issue = 0;
  pipe.slot_wr = 0;
  pipe.wave = wave;
#pragma unroll
  for (int ph = 0; ph < (PH == kPhasePieces ? kRingPhases - 1 : kRingPhases - 2); ++ph) pipe.issue_phase();
  if constexpr (PH == kPhasePieces && kG48LeaderDma && kG48AsmReads) {
    // The first barrier period needs what every later one needs: phases 0 and 1 landed (Pipe48::phase_begin).  A fetching wave leaves
    // its eight youngest loads - its shares of phases 2 and 3 - in flight (everything older - the inputs - has then landed); waves 4-7
    // have nothing but their input DMAs outstanding and wait for all of them.  (A launch of one or two tiles per workgroup - a training
    // step of the small nets - paid the flight time of two phases here.)
    g48_prologue_wait(wave);
    __builtin_amdgcn_s_barrier();
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
#ifdef DN_EXP_HALF   // timing experiment only: one wave per SIMD does the work (what a wave sustains ALONE); 1: waves 0-3, 2: waves 4-7 (+ the fetching by 0-3 is lost: combine with DN_EXP_NODMA)
  if ((DN_EXP_HALF == 1) ? wave >= 4 : wave < 4) return;
#endif
  pipe.slot_nxt = 0;
  // this lane group's 4 rows of bias tile 0 (LDS byte address; the stages add tile offsets)
  // (rebuilt at every use from an opaque copy of the thread index - see fresh_lane - instead of being carried in a VGPR)
  auto bias_at = [&](int tile) { return pipe.ring_addr + kRingBytes + ((fresh_lane() >> 4) << 4) + tile * 64; };
#ifdef DN_PIPE_ASM_READS
  pipe.rda_cur = pipe.ring_addr + lane * 16;
  pipe.slot_cur_base = pipe.ring_addr;       // phase 0 lives in slot 0: phase_begin() of phase 0 turns this into rda_cur
  static_for<kPrefetch - 1>([&](auto e_c) { pipe.template prologue_read<decltype(e_c)::value>(); });
  pipe.template bias_prefetch<0>(bias_at(0));    // same order as in steady state: ..., bias, one more A read
  pipe.template prologue_read<kPrefetch - 1>();
#else
  pipe.rd_cur = ring + lane * 16;
  pipe.rd_nxt = ring + lane * 16;
#pragma unroll
  for (int e = 0; e < kPrefetch; ++e) pipe.af[e] = *reinterpret_cast<const f32x4*>(pipe.rd_nxt + e * kPieceBytes);
#endif

#if defined(DN_G48_PRIO) && DN_G48_PRIO == 1   // static priority for the younger half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  unsigned trk = 0;   // fp16 instances: running maximum of the stage inputs' 16-bit patterns (run_stage48, TRK)
  int vset = 0;  // which view-direction rows hold this tile's directions (wave-uniform, next set every tile)
  // OVL: this lane group's 16 xyz table entries in registers for the whole kernel (a slot mid-stage must not read LDS: the
  // compiler would wait with lgkmcnt(0) and drain the weight-fragment reads in flight).  Slots >= 3 are pure sines - or padding,
  // frequency 0 and phase 0: sin(0) = 0 - only slots 0-2 can be identity columns (lane group 0) and keep their two weights.
  float tfreq[OVL ? 16 : 1], tphase[OVL ? 16 : 1], tw_id[3], tw_sin[3];
  float xr_n[PT][3];   // OVL: the next tile's three points of this lane, rotated for its lane group
  BP8 encp;            // OVL: the piece being assembled
  if constexpr (OVL) {
    const f32x4* tabx = reinterpret_cast<const f32x4*>(tab_lds) + (lane >> 4) * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4 e = tabx[u];
      tfreq[u] = e[0]; tphase[u] = e[1];
      if (u < 3) { tw_id[u] = e[2]; tw_sin[u] = e[3]; }
    }
  }
  // OVX: which lanes hold an identity column in slots 0 - 2 (lane group 0, by the table's identity weight): three lane masks
  unsigned long long idm[3] = {0ull, 0ull, 0ull};
  if constexpr (OVX) {
    const f32x4* tabx = reinterpret_cast<const f32x4*>(tab_lds) + (lane >> 4) * 16;
#pragma unroll
    for (int u = 0; u < 3; ++u) idm[u] = __ballot(tabx[u][2] != 0.0f);
  }
  const int n_tiles = static_cast<int>(p.n_tiles);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, vset = (VSETS == 2 ? vset ^ 1 : (vset == 2 ? 0 : vset + 1))) {
    // training forward: this wave's three point groups' saved-unit bases and its mask words' (s8-48 layout, mlp_geo48.h) - wave-
    // uniform, kept in scalar registers for the tile; every store adds a small offset (store16_uniform_at)
    const char* act_grp[PT] = {};
    const char* mask_base = nullptr;
    if constexpr (SAVE != 0) {
      const long long wt = static_cast<long long>(tile) * WAVES + wave;
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const long long G = wt * PT + t;   // the group's index along the point sequence
        act_grp[t] = uniform_ptr(p.act + ((G >> 1) * p.act_pieces * 2 + (G & 1)) * kPieceBytes);
      }
      mask_base = uniform_ptr(p.masks + wt * p.mask_words * (2 * kPieceBytes));
    }
    // ---- xyz encoding of this lane's three points, its 16 columns each, into the per-wave LDS stash ----
    // (OVL: only a workgroup's first tile is encoded here; every later one was encoded during the tile before it)
    const bool first_tile = tile == static_cast<int>(blockIdx.x);
#ifdef DN_EXP_NOTOP   // timing experiment only: no top-of-tile block (the stash keeps whatever it held; inputs are not staged)
    if (false) {
#else
    if ((!OVL && !OVX) || first_tile) {
#endif
      const int ln = fresh_lane();
      const int j = ln & 15;
      const f32x4* tabx = reinterpret_cast<const f32x4*>(tab_lds) + (ln >> 4) * 16;
      char* pex = pex_of(ln);
#ifdef DN_PIPE_LEADER_DMA
      // waves 4-7 issue no weight DMAs, so no counted wait of theirs ever pushes this tile's input DMAs (issued one tile
      // ago) through: they wait for them here - by now their VMEM queue holds nothing else but the last output stores
      if (wave >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      float in[PT][7];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < 7; ++c) in[t][c] = inbuf[c * PPW + t * 16 + j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int nxt = tile + gridDim.x;
        if (nxt < n_tiles) issue_inputs(nxt, vset ^ 1);
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)  // plain mul then add (train_utils.py:136)
          x[c] = (p.mode == 0) ? in[t][c] + in[t][3 + c] * in[t][6] : in[t][c];
        float xr[3];
        rotate3(x, ln >> 4, xr);
        BP8 pk[KXP];
#pragma unroll
        for (int k = 0; k < KXP; ++k) {
          BP8 piece;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            piece[e] = static_cast<Elem>((k * 8 + e < 3) ? pe_value<true>(xr[(k * 8 + e) % 3], tabx[k * 8 + e])
                                                         : pe_value<false>(xr[(k * 8 + e) % 3], tabx[k * 8 + e]));
          *reinterpret_cast<BP8*>(pex + (t * KXP + k) * kPieceBytes) = piece;
          pk[k] = piece;
        }
        if constexpr (SAVE != 0) {
          static_assert(KXP == 2, "the xyz panel is one saved unit");
          // (the helper lambdas of the tile body are defined further down: the same store, spelled out)
          typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
          auto cv = [](unsigned d0, unsigned d1) { return cvt_pairs_8bit<false>(d0, d1, 1.0f); };
          const u32x4_ a = __builtin_bit_cast(u32x4_, pk[0]), b = __builtin_bit_cast(u32x4_, pk[1]);
          store16_unit48(act_grp[t], static_cast<unsigned>(p.slot_xyz) * (2 * kPieceBytes), static_cast<unsigned>(ln) * 16u,
                             make_uint4(cv(a[0], a[1]), cv(a[2], a[3]), cv(b[0], b[1]), cv(b[2], b[3])));
        }
      }
    }
    auto pe_xyz = [&](int t, int k) { return *reinterpret_cast<const BP8*>(pex_of(pipe.lane16 >> 4) + (t * KXP + k) * kPieceBytes); };
    auto no_pe = [&](int, int) { return BP8{}; };
    // ---- training forward: saved units and mask words (s8-48 layout, mlp_geo48.h) ----
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned maskw[PT][2];
    auto mask_clear = [&]() {
#pragma unroll
      for (int t = 0; t < PT; ++t) { maskw[t][0] = 0u; maskw[t][1] = 0u; }
    };
    // two 16-bit pairs -> four e4m3 bytes
    auto to_e4m3 = [](unsigned d0, unsigned d1) { return cvt_pairs_8bit<false>(d0, d1, 1.0f); };
    auto save_unit = [&](auto t_c, int slot, const BP8& lo, const BP8& hi) {
      if constexpr (SAVE != 0) {
        constexpr int t = decltype(t_c)::value;
        const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
        store16_unit48(act_grp[t], static_cast<unsigned>(slot) * (2 * kPieceBytes), pipe.lane16,
                           make_uint4(to_e4m3(a[0], a[1]), to_e4m3(a[2], a[3]), to_e4m3(b[0], b[1]), to_e4m3(b[2], b[3])));
      }
    };
    // after emit48 of output tile nt of group t into bo: ReLU mask bits off the packed outputs, and every fourth tile one unit
    auto mask_tail = [&](auto nt_c, auto t_c, const auto& bo) {
#ifdef DN_EXP_TF_NOMASK   // timing experiment only: no ReLU mask bits
      return;
#endif
      if constexpr (SAVE != 0) {
        constexpr int nt = decltype(nt_c)::value, t = decltype(t_c)::value;
        const u32x4 w = __builtin_bit_cast(u32x4, bo[nt / 2]);
        const unsigned ones = 0x00010001u;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          // (through a named scalar: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index - hipcc 7.2)
          const unsigned pair = w[(nt & 1) * 2 + d];
          // non-zero <=> the unit is active: min(x, 1) on both 16-bit halves.  As an instruction: written with the vector builtins
          // hipcc lowers it to two compares, two selects and a v_perm_b32
          unsigned m;
          asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(pair), "s"(ones));
          // (opaque: as plain ORs the optimiser may reassociate the stage's chain and keep every m alive to its end)
          unsigned mw = maskw[t][nt >> 3];   // (asm operands do not capture: name a local)
          asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(mw) : "v"(m), "n"((nt & 7) * 2 + d));
          maskw[t][nt >> 3] = mw;
        }
      }
    };
    auto unit_tail = [&](auto nt_c, auto t_c, const auto& bo, int slot0) {
#ifdef DN_EXP_TF_NOUNIT   // timing experiment only: no saved units (no 8-bit conversion, no store)
      return;
#endif
      if constexpr (SAVE != 0) {
        constexpr int nt = decltype(nt_c)::value;
        if constexpr (nt % 4 == 3) save_unit(t_c, slot0 + nt / 4, bo[nt / 2 - 1], bo[nt / 2]);
      }
    };
    auto mask_store = [&](int stage) {
      if constexpr (SAVE != 0) {
        store16_uniform_at(mask_base, static_cast<unsigned>(stage) * (2 * kPieceBytes), pipe.lane16, make_uint4(maskw[0][0], maskw[0][1], maskw[1][0], maskw[1][1]));
        if constexpr (PT == 3)
          store16_uniform_at(mask_base, static_cast<unsigned>(stage) * (2 * kPieceBytes) + kPieceBytes, pipe.lane16, make_uint4(maskw[PT - 1][0], maskw[PT - 1][1], 0u, 0u));
      }
    };
    constexpr int KHU = KH / 2;   // units of a hidden vector

    BP8 ba[PT][KH], bb[PT][KH];
    int bias_tile = 0;
    float out4[PT][4];
    if constexpr (XS) {
      // ================= explicit schedule (mlp_stage48.h run_stage48x): same stages, same pieces, same arithmetic =================
      static_assert(DC >= 2 && KDP == 1, "a trunk behind layer1; the view-direction panel is one piece");
      constexpr int PX = PH;   // barrier period: 32 pieces for W = 256 (every stage starts on one), 16 otherwise
      f32x4 pacc[PT];
      auto nothing = [&](auto) {};
      // ---- layer1: xyz encoding -> W, no activation (models.py:238) ----
      {
        BP8 pe[PT][KXP];
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int k = 0; k < KXP; ++k) pe[t][k] = pe_xyz(t, k);
        run_stage48x<F, PX, NT, KXP, 0, 0, false, 0, 6, 0, 0>(pipe, pe, no_pe, bias_at(0), 0u, pacc,
            [&](auto nt_c, auto s_c) { hidden_op48<F, false, decltype(nt_c)::value, decltype(s_c)::value>(pacc, ba); }, nothing);
      }
      bias_tile += NT;
      // ---- trunk (models.py:239-246): each stage's first blocks finish the stage before it ----
      static_for<DC - 1>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        constexpr int P0 = xs_trunk_pos(i, NT, KH, KXP, MASKC) % PX;
        auto& bin = (i % 2 == 0) ? ba : bb;
        auto& bout = (i % 2 == 0) ? bb : ba;
        auto ops = [&](auto nt_c, auto s_c) { hidden_op48<F, true, decltype(nt_c)::value, decltype(s_c)::value>(pacc, bout); };
        auto pend = [&](auto s_c) { hidden_op48<F, (i > 0), NT - 1, decltype(s_c)::value>(pacc, bin); };
        constexpr int PN = i > 0 ? 12 : 6;
        constexpr int BY = (NT - 1) / 2;
        if constexpr ((MASKC >> i) & 1u) {
          BP8 pe[PT][KXP];   // the skip layer's second K panel, in registers for the stage (see layer1)
#pragma unroll
          for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int k = 0; k < KXP; ++k) pe[t][k] = pe_xyz(t, k);
          run_stage48x<F, PX, NT, KH, KXP, P0, false, 0, 12, PN, BY, (i == 0 ? 2 : 1)>(pipe, bin, [&](int t, int k) { return pe[t][k]; }, bias_at(bias_tile), 0u, pacc, ops, pend, &trk);
        } else {
          run_stage48x<F, PX, NT, KH, 0, P0, false, 0, 12, PN, BY, (i == 0 ? 2 : 1)>(pipe, bin, no_pe, bias_at(bias_tile), 0u, pacc, ops, pend, &trk);
        }
        bias_tile += NT;
      });
      auto& hx = ((DC - 1) % 2 == 0) ? ba : bb;
      auto& hy = ((DC - 1) % 2 == 0) ? bb : ba;
      // ---- fc_alpha (its own 16-row tile, row 0, streamed first) + fc_feat with ReLU (models.py:248-249) ----
      constexpr int POS_A = xs_trunk_pos(DC - 1, NT, KH, KXP, MASKC) % PX;
      run_stage48x<F, PX, 1, KH, 0, POS_A, false, 0, 3, (DC > 1 ? 12 : 6), (NT - 1) / 2>(pipe, hx, no_pe, bias_at(bias_tile), 0u, pacc,
          [&](auto, auto s_c) { constexpr int t = decltype(s_c)::value; pick_op48(out4[t][3], pacc[t][0]); },   // row 0: lane group 0, register 0
          [&](auto s_c) { hidden_op48<F, (DC > 1), NT - 1, decltype(s_c)::value>(pacc, hx); });
      constexpr int POS_F = (POS_A + KH) % PX;
      run_stage48x<F, PX, NT, KH, 0, POS_F, false, 0, 12, 3, KH, 1>(pipe, hx, no_pe, bias_at(bias_tile + 1), 0u, pacc,
          [&](auto nt_c, auto s_c) { hidden_op48<F, true, decltype(nt_c)::value, decltype(s_c)::value>(pacc, hy); },
          [&](auto s_c) { constexpr int t = decltype(s_c)::value; pick_op48(out4[t][3], pacc[t][0]); }, &trk);
      bias_tile += NT + 1;
      // ---- view-direction encoding (one 32-deep piece per point group), into the xyz stash (dead once the trunk is done) ----
      __builtin_amdgcn_sched_barrier(0);
      {
        const int ln = fresh_lane();
        const int j = ln & 15;
        const f32x4* tabd = reinterpret_cast<const f32x4*>(tab_lds + 1024) + (ln >> 4) * 8;
        char* pex = pex_of(ln);
        static_for<PT>([&](auto t_c) {
          constexpr int t = decltype(t_c)::value;
          float v[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = inbuf[(7 + 3 * vset + c) * PPW + t * 16 + j];
          float vr[3];
          rotate3(v, ln >> 4, vr);
          BP8 piece;
#pragma unroll
          for (int e = 0; e < 8; ++e) piece[e] = static_cast<Elem>((e < 3) ? pe_value<true>(vr[e % 3], tabd[e]) : pe_value<false>(vr[e % 3], tabd[e]));
          *reinterpret_cast<BP8*>(pex + t * kPieceBytes) = piece;
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- layers_dir[0] on cat(feat, view) -> W/2, ReLU (models.py:250-252) ----
      constexpr int POS_D = (POS_F + NT * KH) % PX;
      constexpr int KT_D = KH + KDP;
      BP8 bg[PT][KH / 2];
      BP8 ped[PT];
      // OVX state: the two values of a dword pair, two table entries in flight (frequency in revolutions, phase), LDS addresses
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      float ev[2] = {0.0f, 0.0f};
      f32x2 tbl[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
      unsigned tab_addr_v = 0u, stash_addr_v = 0u;
      float in_n[PT][7];
      if constexpr (OVX) {
        static_assert(kEncOps <= enc_begin((NT / 2) * KT_D, KT_D), "the view-direction stage has a gap for every encoding op");
        // the next tile's inputs: their DMAs were issued a whole tile ago (the first tile's: at its top).  Waves 4-7 issue no weight
        // DMAs, so no counted wait of theirs pushes input DMAs through: they wait here (their queue holds nothing else but the last
        // output stores); waves 0-3 have passed dozens of barrier-period waits (vmcnt(0)) since.  Branch inside the statement.
        {
          const unsigned must_wait = __builtin_amdgcn_readfirstlane(wave >= 4 ? 1u : 0u);
          asm volatile("s_cmp_eq_u32 %0, 0\n\t"
                       "s_cbranch_scc1 .Ldn_ovx_nowait%=\n\t"
                       "s_waitcnt vmcnt(0)\n"
                       ".Ldn_ovx_nowait%=:" ::"s"(must_wait) : "scc", "memory");
        }
        const int ln = fresh_lane();
        const int j = ln & 15;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int c = 0; c < 7; ++c) in_n[t][c] = inbuf[c * PPW + t * 16 + j];
        tab_addr_v = pipe.ring_addr + kRingBytes + static_cast<unsigned>(q.bias_bytes) + static_cast<unsigned>(ln >> 4) * 256u;
        stash_addr_v = pipe.ring_addr + kRingBytes + static_cast<unsigned>(q.bias_bytes) + kG48TableBytes + wave * (PT * KXP * kPieceBytes) + static_cast<unsigned>(ln) * 16u;
        asm volatile("ds_read_b64 %0, %1" : "=v"(tbl[0]) : "v"(tab_addr_v));    // slot 0's table entry (the others ride in the stage)
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) ped[t] = *reinterpret_cast<const BP8*>(pex_of(pipe.lane16 >> 4) + t * kPieceBytes);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ped[0]), "+v"(ped[1]), "+v"(ped[2]), "+v"(pipe.af[0]), "+v"(pipe.af[1]), "+v"(pipe.bias_nxt), "+v"(tbl[0]));
      if constexpr (OVX) {
        // rows 0 - 6 and this tile's view-direction set are free again: the inputs of the tile after next (clamped: a tile that
        // does not exist re-reads this one's, nobody looks at the result)
        const int t2 = tile + 2 * static_cast<int>(gridDim.x);
        issue_inputs_flat(t2 < n_tiles ? t2 : tile, vset);
        const int ln = fresh_lane();
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          float x[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) x[c] = in_n[t][c] + in_n[t][3 + c] * in_n[t][6];   // plain mul then add (train_utils.py:136)
          rotate3(x, ln >> 4, xr_n[t]);
        }
      }
      // one gap of one block of the stage: its share of the encoding queue (OVX; nothing otherwise)
      auto enc_hook = [&](auto b_c, auto w_c) {
        if constexpr (OVX) {
          constexpr int b = decltype(b_c)::value, w = decltype(w_c)::value;
          constexpr int q0 = enc_begin(b, KT_D), cap = enc_cap(b, KT_D);
          auto one = [&](auto q_c) {
            constexpr int qq = decltype(q_c)::value;
            if constexpr (qq < kEncOps) {
              // (asm operands do not make a lambda capture: everything the statements name is bound to a local first)
              constexpr int mt = enc_table_slot_at(qq);
              constexpr int m = enc_slot_of(qq), u = m % 16, tq = m / 16, e = u % 8, k2 = u / 8;
              constexpr int kind = enc_kind(u, qq - enc_first(m));
              const unsigned ta = tab_addr_v, sa = stash_addr_v;
              if constexpr (mt >= 0) {   // the table entry of a later slot: an LDS read older than this block's A-fragment read
                f32x2& dst = tbl[mt & 1];
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(ta), "n"((mt % 16) * 16));
              }
              const float xc = xr_n[tq][u % 3];
              float& val = ev[e & 1];
              const f32x2 entry = tbl[m & 1];
              if constexpr (kind == kEncMul) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(val) : "v"(xc), "v"(entry[0]));
              else if constexpr (kind == kEncFract) asm volatile("v_fract_f32 %0, %0" : "+v"(val));
              else if constexpr (kind == kEncAdd) asm volatile("v_add_f32 %0, %0, %1" : "+v"(val) : "v"(entry[1]));
              else if constexpr (kind == kEncSin) asm volatile("v_sin_f32 %0, %0" : "+v"(val));
              else if constexpr (kind == kEncSel) {
                const unsigned long long im = idm[u < 3 ? u : 0];
                asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(val) : "v"(xc), "s"(im));
              } else if constexpr (kind == kEncCvt) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const float lo = ev[0], hi = ev[1];
                unsigned dw;
                if constexpr (F == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw) : "v"(lo), "v"(hi));
                else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(dw) : "v"(lo), "v"(hi));
                u32x4 pw = __builtin_bit_cast(u32x4, encp);
                pw[e / 2] = dw;
                encp = __builtin_bit_cast(BP8, pw);
              } else {
                const BP8 piece = encp;
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(sa), "v"(piece), "n"((tq * KXP + k2) * kPieceBytes));
              }
            }
          };
          if constexpr (w < 2) one(std::integral_constant<int, q0 + w>{});
          else if constexpr (w == 2) static_for<cap - 2>([&](auto j_c) { one(std::integral_constant<int, q0 + 2 + decltype(j_c)::value>{}); });
        }
      };
      run_stage48x<F, PX, NT / 2, KH, KDP, POS_D, false, 0, 12, 12, (NT - 1) / 2, 1>(pipe, hy, [&](int t, int) { return ped[t]; }, bias_at(bias_tile), 0u, pacc,
          [&](auto nt_c, auto s_c) { hidden_op48<F, true, decltype(nt_c)::value, decltype(s_c)::value>(pacc, bg); },
          [&](auto s_c) { hidden_op48<F, true, NT - 1, decltype(s_c)::value>(pacc, hy); }, &trk, enc_hook);
      bias_tile += NT / 2;
      // ---- fc_rgb (models.py:253) ----
      constexpr int POS_R = (POS_D + (NT / 2) * (KH + KDP)) % PX;
      constexpr int END = POS_R + KH / 2;
      static_assert(END <= PX && (END - 1) / kPhasePieces == POS_R / kPhasePieces, "the tail stays inside one phase");
      constexpr int PAD_R = (kPhasePieces - END % kPhasePieces) % kPhasePieces;
      static_assert((END + PAD_R) % PX == 0, "a tile pass is a whole number of barrier periods");
      run_stage48x<F, PX, 1, KH / 2, 0, POS_R, true, PAD_R, 9, 12, (NT / 2 - 1) / 2, 1>(pipe, bg, no_pe, bias_at(bias_tile), bias_at(0), pacc,
          [&](auto, auto s_c) { constexpr int sv = decltype(s_c)::value; pick_op48(out4[sv / 3][sv % 3], pacc[sv / 3][sv % 3]); },
          [&](auto s_c) { hidden_op48<F, true, NT / 2 - 1, decltype(s_c)::value>(pacc, bg); }, &trk);
      if constexpr (PAD_R != 0) pipe.template skip_xs<PX, END, PAD_R>();   // (settles at its end)
      else pipe.settle();
      // the last tile's rows: nothing rides behind this stage, so its accumulators are read here - behind the wait states a 4-pass
      // MFMA's result needs (7; the compiler sees no MFMA and pads nothing)
      asm volatile("s_nop 7\n\ts_nop 1" ::: "memory");
      static_for<9>([&](auto s_c) { constexpr int sv = decltype(s_c)::value; pick_op48(out4[sv / 3][sv % 3], pacc[sv / 3][sv % 3]); });
    } else {
    // ---- layer1: xyz encoding -> W, no activation (models.py:238) ----
    // (the encoding pieces go into registers once per stage - bb is still free here: read through the pe_xyz lambda they
    // are re-read from LDS for every tile, because the DMA asm's memory clobber forbids the compiler to keep them, each time
    // behind a compiler-placed lgkmcnt(0): layer1 took 14 k cycles per pass for 3 k cycles of MFMA work)
    {
      BP8 pe[PT][KXP];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int k = 0; k < KXP; ++k) pe[t][k] = pe_xyz(t, k);
      run_stage48<F, NT, KXP, 0, 0, false, ST, 0, PH>(pipe, pe, no_pe, bias_at(0), 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
        emit48<F, false, decltype(nt_c)::value>(acc, ba[decltype(t_c)::value]);
        unit_tail(nt_c, t_c, ba[decltype(t_c)::value], p.slot_layer1);
      });
    }
    bias_tile += NT;
#if defined(DN_STAMP) && DN_STAMP == 4
    pipe.template stage_end<0>();
#endif
    if constexpr (OVL) {
      // the next tile's inputs - their DMAs were issued a whole tile ago (the first tile's: at its top) - become this lane's three
      // rotated points, then the rows are handed to the DMAs of the tile after next.  Waves 4-7 issue no weight DMAs, so no counted
      // wait of theirs pushes input DMAs through: they wait here (as does everyone on a workgroup's first tile).  The branch
      // lives inside the asm statement: the compiler sees straight-line code.
      {
        const unsigned must_wait = __builtin_amdgcn_readfirstlane((wave >= 4 || first_tile) ? 1u : 0u);
        asm volatile("s_cmp_eq_u32 %0, 0\n\t"
                     "s_cbranch_scc1 .Ldn_ovl_nowait%=\n\t"
                     "s_waitcnt vmcnt(0)\n"
                     ".Ldn_ovl_nowait%=:" ::"s"(must_wait) : "scc", "memory");
      }
      const int ln = fresh_lane();
      const int j = ln & 15;
      float in[PT][7];
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int c = 0; c < 7; ++c) in[t][c] = inbuf[c * PPW + t * 16 + j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int t2 = tile + 2 * static_cast<int>(gridDim.x);
        issue_inputs_flat(t2 < n_tiles ? t2 : tile, vset == 0 ? 2 : vset - 1);   // rows 0-6 are free again; view set (vset + 2) % 3
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = in[t][c] + in[t][3 + c] * in[t][6];   // plain mul then add (train_utils.py:136)
        rotate3(x, ln >> 4, xr_n[t]);
      }
    }
    // one encoding slot of the next tile (OVL): M = 0 .. 47 in the order the first two trunk stages emit their output tiles
    auto enc_step = [&](auto m_c) {
      if constexpr (OVL) {
        constexpr int m = decltype(m_c)::value;
        constexpr int tq = m / 16, k = (m % 16) / 8, e = m % 8, u = k * 8 + e;
        static_assert(KXP == 2 && PT == 3, "48 slots per lane");
        const float xc = xr_n[tq][u % 3];
        const float rev = __builtin_amdgcn_fractf(xc * tfreq[u]) + tphase[u];       // (pe_value, op for op)
        const float sv = __builtin_amdgcn_sinf(rev);
        float val = sv;
        if constexpr (u < 3) val = tw_id[u] * xc + tw_sin[u] * sv;
        encp[e] = static_cast<Elem>(val);
        if constexpr (e == 7) *reinterpret_cast<BP8*>(pex_of(fresh_lane()) + (tq * KXP + k) * kPieceBytes) = encp;
      }
    };
    // ---- heads on the trunk output hx (hy: the other, by then free, activation set) ----
    auto heads = [&](const BP8 (&hx)[PT][KH], BP8 (&hy)[PT][KH], auto view_c) __attribute__((always_inline)) {
      if constexpr (decltype(view_c)::value) {
        // ---- fc_alpha (its own 16-row tile, row 0, streamed first) + fc_feat with ReLU (models.py:248-249) ----
        run_stage48<F, 1, KH, 0, 0, false, ST, 0, PH>(pipe, hx, no_pe, bias_at(bias_tile), 0u, [&](auto, auto t_c, const f32x4& acc) {
          out4[decltype(t_c)::value][3] = acc[0];  // row 0 lives in lane group 0, register 0
        });
        mask_clear();
        run_stage48<F, NT, KH, 0, KH % PH, false, ST, 0, PH, 1>(pipe, hx, no_pe, bias_at(bias_tile + 1), 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
          emit48<F, true, decltype(nt_c)::value>(acc, hy[decltype(t_c)::value]);
          mask_tail(nt_c, t_c, hy[decltype(t_c)::value]);
          unit_tail(nt_c, t_c, hy[decltype(t_c)::value], p.slot_feat);
        }, &trk);
        mask_store(p.D - 1);
        bias_tile += NT + 1;
        // ---- view-direction encoding (one 32-deep piece per point group) ----
        // fenced on both sides: interleaved into the fc_feat MFMAs its temporaries push finished activation pieces to scratch
        __builtin_amdgcn_sched_barrier(0);
        // (one point group at a time, into the xyz stash - dead once the trunk is done - so that neither the block's
        // temporaries nor the pieces themselves compete with the 96 registers of fc_feat's output)
        BP8 ped_now[PT];
        {
          const int ln = fresh_lane();
          const int j = ln & 15;
          const f32x4* tabd = reinterpret_cast<const f32x4*>(tab_lds + 1024) + (ln >> 4) * 8;
          char* pex = pex_of(ln);
          static_for<PT>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            float v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = inbuf[(7 + 3 * vset + c) * PPW + t * 16 + j];
            float vr[3];
            rotate3(v, ln >> 4, vr);
            BP8 piece;
#pragma unroll
            for (int e = 0; e < 8; ++e) piece[e] = static_cast<Elem>((e < 3) ? pe_value<true>(vr[e % 3], tabd[e]) : pe_value<false>(vr[e % 3], tabd[e]));
            if constexpr (OVL) ped_now[t] = piece;   // (the stash holds the NEXT tile's xyz pieces by now: same lane, no need to park)
            else *reinterpret_cast<BP8*>(pex + t * kPieceBytes) = piece;
            if constexpr (SAVE != 0) {
              // one piece = 8 bytes per lane: lanes of groups 0 / 1 store [their own 8 bytes | those of groups 2 / 3] (the upper
              // half of the unit's rows is then a copy nobody reads) - one 32-feature fragment for the weight-gradient kernel
              const u32x4 a = __builtin_bit_cast(u32x4, piece);
              const unsigned w0 = to_e4m3(a[0], a[1]), w1 = to_e4m3(a[2], a[3]);
              const auto s0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);   // [1]: lanes < 32 receive lane + 32's word
              const auto s1 = __builtin_amdgcn_permlane32_swap(w1, w1, false, false);
              store16_unit48(act_grp[t], static_cast<unsigned>(p.slot_dir) * (2 * kPieceBytes), static_cast<unsigned>(ln) * 16u, make_uint4(w0, w1, s0[1], s1[1]));
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- layers_dir[0] on cat(feat, view) -> W/2, ReLU (models.py:250-252) ----
        constexpr int POS_D = ((NT + 1) * KH) % PH;
        BP8 bg[PT][KH / 2];
        BP8 ped[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          if constexpr (OVL) ped[t] = ped_now[t];
          else ped[t] = *reinterpret_cast<const BP8*>(pex_of(pipe.lane16 >> 4) + t * kPieceBytes);
        }
        auto pe_dir = [&](int t, int) { return ped[t]; };
        mask_clear();
        run_stage48<F, NT / 2, KH, KDP, POS_D, false, ST, 0, PH, 1>(pipe, hy, pe_dir, bias_at(bias_tile), 0u, [&](auto nt_c, auto t_c, const f32x4& acc) {
          emit48<F, true, decltype(nt_c)::value>(acc, bg[decltype(t_c)::value]);
          mask_tail(nt_c, t_c, bg[decltype(t_c)::value]);
          unit_tail(nt_c, t_c, bg[decltype(t_c)::value], p.slot_dirout);
        }, &trk);
        mask_store(p.D);
        bias_tile += NT / 2;
        // ---- fc_rgb (models.py:253) ----
        constexpr int POS_R = (POS_D + (NT / 2) * (KH + KDP)) % PH;
        constexpr int END = POS_R + KH / 2;
        static_assert(END <= PH && (END - 1) / kPhasePieces == POS_R / kPhasePieces, "the tail stays inside one phase");
        constexpr int PAD_R = (kPhasePieces - END % kPhasePieces) % kPhasePieces;
        static_assert((END + PAD_R) % PH == 0, "a tile pass is a whole number of barrier periods");
        run_stage48<F, 1, KH / 2, 0, POS_R, true, ST, PAD_R, PH, 1>(pipe, bg, no_pe, bias_at(bias_tile), bias_at(0), [&](auto, auto t_c, const f32x4& acc) {
          constexpr int t = decltype(t_c)::value;
          out4[t][0] = acc[0]; out4[t][1] = acc[1]; out4[t][2] = acc[2];
        }, &trk);
        if constexpr (PAD_R != 0) pipe.template skip<END, PAD_R, PH>();   // (settles at its end)
        else pipe.settle();
      } else {
        // ---- fc_out (models.py:256) ----
        static_assert(PH == kPhasePieces, "no-viewdirs nets run the every-phase barrier");
        run_stage48<F, 1, KH, 0, 0, true, ST, (kPhasePieces - KH % kPhasePieces) % kPhasePieces, kPhasePieces, 1>(pipe, hx, no_pe, bias_at(bias_tile), bias_at(0), [&](auto, auto t_c, const f32x4& acc) {
          constexpr int t = decltype(t_c)::value;
          out4[t][0] = acc[0]; out4[t][1] = acc[1]; out4[t][2] = acc[2]; out4[t][3] = acc[3];
        }, &trk);
        if constexpr (KH % kPhasePieces != 0) pipe.template skip<KH % kPhasePieces, kPhasePieces - KH % kPhasePieces>();
        else pipe.settle();
      }
    };
    // ---- trunk (models.py:239-246): the activations ping-pong between two register sets ----
    if constexpr (FIXED) {
      static_assert((NT * KXP) % PH == 0 && (NT * KH) % PH == 0 && (NT * (KH + KXP)) % PH == 0,
                    "layer1 and every trunk layer span whole barrier periods: each stage starts at position 0 of one");
      static_for<DC - 1>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        auto& bin = (i % 2 == 0) ? ba : bb;
        auto& bout = (i % 2 == 0) ? bb : ba;
        auto emit = [&](auto nt_c, auto t_c, const f32x4& acc) {
          emit48<F, true, decltype(nt_c)::value>(acc, bout[decltype(t_c)::value]);
          mask_tail(nt_c, t_c, bout[decltype(t_c)::value]);
          unit_tail(nt_c, t_c, bout[decltype(t_c)::value], p.slot_trunk0 + i * KHU);
          if constexpr (OVL && i < 2) {
            static_assert(!OVL || NT * PT == 24, "two trunk stages = the 48 encoding slots of a lane");
            enc_step(std::integral_constant<int, i * 24 + decltype(nt_c)::value * PT + decltype(t_c)::value>{});
          }
        };
        mask_clear();
        if constexpr ((MASKC >> i) & 1u) {
#ifdef DN_G48_SKIP_PE_FROM_LDS
          run_stage48<F, NT, KH, KXP, 0, false, false, 0, PH>(pipe, bin, pe_xyz, bias_at(bias_tile), 0u, emit);
#else
          BP8 pe[PT][KXP];   // the skip layer's second K panel, in registers for the stage (see layer1)
#pragma unroll
          for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int k = 0; k < KXP; ++k) pe[t][k] = pe_xyz(t, k);
          run_stage48<F, NT, KH, KXP, 0, false, false, 0, PH, (i == 0 ? 2 : 1)>(pipe, bin, [&](int t, int k) { return pe[t][k]; }, bias_at(bias_tile), 0u, emit, &trk);
#endif
        } else {
          run_stage48<F, NT, KH, 0, 0, false, false, 0, PH, (i == 0 ? 2 : 1)>(pipe, bin, no_pe, bias_at(bias_tile), 0u, emit, &trk);
        }
        mask_store(i);
        bias_tile += NT;
      });
#if defined(DN_STAMP) && DN_STAMP == 4
      pipe.template stage_end<1>();
#endif
      if constexpr ((DC - 1) % 2 == 0) heads(ba, bb, std::integral_constant<bool, VIEWC != 0>{});
      else heads(bb, ba, std::integral_constant<bool, VIEWC != 0>{});
    } else {
      // two layers per iteration of a run-time loop
      auto trunk_layer = [&](int i, const BP8 (&bin)[PT][KH], BP8 (&bout)[PT][KH]) __attribute__((always_inline)) {
        auto emit = [&](auto nt_c, auto t_c, const f32x4& acc) {
          emit48<F, true, decltype(nt_c)::value>(acc, bout[decltype(t_c)::value]);
          mask_tail(nt_c, t_c, bout[decltype(t_c)::value]);
          unit_tail(nt_c, t_c, bout[decltype(t_c)::value], p.slot_trunk0 + i * KHU);
        };
        mask_clear();
        // (the range tracker costs the run-time-shape W = 256 instance 315 spilled registers: it tracks in the heads only, and
        // dn_fp16_range_guard() says so)
        constexpr int TRK_RT = W == 128 ? 2 : 0;
        if ((p.skip_mask >> i) & 1u) run_stage48<F, NT, KH, KXP, 0, false, true, 0, kPhasePieces, TRK_RT>(pipe, bin, pe_xyz, bias_at(bias_tile), 0u, emit, &trk);
        else run_stage48<F, NT, KH, 0, 0, false, true, 0, kPhasePieces, TRK_RT>(pipe, bin, no_pe, bias_at(bias_tile), 0u, emit, &trk);
        mask_store(i);
        bias_tile += NT;
      };
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
      if (p.use_viewdirs) heads(ba, bb, std::true_type{});
      else heads(ba, bb, std::false_type{});
    }
    }   // !XS
#ifdef DN_STAMP
    pipe.pass_end();
#endif
    const int lo = fresh_lane();
    if constexpr (COMP != 0) {
      // the tile's 384 raw rows go into the xyz stash (dead from the trunk to the next tile's top; one 6 KiB image, point order)
      f32x4* stage = reinterpret_cast<f32x4*>(smem + kRingBytes + q.bias_bytes + kG48TableBytes);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        if (lo < 16) {
          f32x4 o;
          o[0] = out4[t][0]; o[1] = out4[t][1]; o[2] = out4[t][2]; o[3] = out4[t][3];
          stage[wave * PPW + t * 16 + lo] = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (not __syncthreads: its vmcnt(0) would drain the weight ring)
      __builtin_amdgcn_s_barrier();
      const int S = p.S;
      const int rays_per_tile = PPG / S;
      for (int r = wave; r < rays_per_tile; r += WAVES) {   // wave-uniform
        const long long ray = static_cast<long long>(tile) * rays_per_tile + r;
        if (ray < q.comp.n_rays) {
          const f32x4* rows = stage + r * S;
          composite_ray([&](int sc) { const f32x4 r = rows[sc]; return make_float4(r[0], r[1], r[2], r[3]); }, p.z + ray * S,
                        p.rays + ray * p.ray_stride + 3, ray, lo, static_cast<const float*>(nullptr), 0.0f, q.comp.white, q.comp.th,
                        q.comp.n_thres, q.comp.n_rays, S, q.comp.rgb, q.comp.disp, q.comp.acc, q.comp.weights, q.comp.depth, q.comp.dex,
                        q.comp.nonfinite, RngRef{nullptr, 0u});
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stash is the next tile's encodings' again
      __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const int pt = tile * PPG + wave * PPW + t * 16 + (lo & 15);
        if (pt < n_points && lo < 16) {
          f32x4 o;
          o[0] = out4[t][0]; o[1] = out4[t][1]; o[2] = out4[t][2]; o[3] = out4[t][3];
          __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p.out + static_cast<long long>(pt) * 4));
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (F == 2) {
    const bool out_of_range = (trk & 0xFFFFu) >= 0x7C00u || (trk >> 16) >= 0x7C00u;
    if (p.range_flag != nullptr && __ballot(out_of_range) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(p.range_flag, 1u);
  }
#ifdef DN_STAMP
  if ((threadIdx.x & 63) == 0) {
    unsigned* d = q.dbg + (blockIdx.x * WAVES + wave) * 16;
    d[0] = pipe.st_vm; d[1] = pipe.st_bar; d[2] = pipe.st_dma; d[3] = pipe.st_seg; d[4] = pipe.st_n;
    d[5] = pipe.st_sub[0]; d[6] = pipe.st_sub[1]; d[7] = pipe.st_sub[2]; d[3] = pipe.st_sub[3];
    d[8] = pipe.st_top; d[9] = pipe.st_tail; d[10] = pipe.st_cls[0]; d[11] = pipe.st_cls[1];
  }
#endif
}

}  // namespace dn
