// Shared by the 48-points-per-wave kernels (mlp_fused48.hip: inference + training forward; mlp_train48.hip: backward-data chain):
// the Pipe configuration of this geometry and the per-stage MFMA loop.  Included by those two translation units only.
#pragma once
#if defined(DN_PREFETCH) && !defined(DN_ABLATION_BUILD)
#error "DN_PREFETCH is set by this file (the 48-point kernel's FIFO depth); a command-line value is an ablation hook (scripts/build_exp.sh)"
#endif
#ifdef DN_G48_PREFETCH   // (ablation hook, refused by mlp_device.h outside an ablation build)
#define DN_PREFETCH DN_G48_PREFETCH
#else
#define DN_PREFETCH 2
#endif
#define DN_PREFETCH_SET_BY_KERNEL_SOURCE 1
#define DN_PIPE_SCALAR_STATE 1   // ring bookkeeping in SGPRs: frees the VGPRs that were spilling (0.5 % on the launch)
#ifndef DN_G48_COMPILER_READS    // (ablation hook: the r01 pipeline with compiler-issued reads and waits)
#define DN_PIPE_ASM_READS 1      // A-fragment / bias LDS reads and their counted waits as opaque asm (mlp_device.h Pipe)
#endif
#ifndef DN_G48_SYMMETRIC_DMA     // (ablation hook: every wave fetches two pieces per phase, as in round 1)
// Asymmetric roles.  The two waves of a SIMD do not share the matrix pipe fairly: the older one (waves 0-3) wins the
// arbitration, runs a phase ahead and then sits at the phase barrier (s_memtime stamps, profiles/r02_fine_net_stalls.md:
// ~640 cycles per phase against ~160 for waves 4-7), while the younger one - the critical path - also paid ~300 cycles per
// phase of its own LDS-DMA issue.  So the waves with the slack fetch the whole weight stream (four pieces each) and
// waves 4-7 issue MFMAs only.
#define DN_PIPE_LEADER_DMA 1
#endif
#include <type_traits>
#include "mlp_geo48.h"

namespace dn {

// the weight pipeline of this geometry (mlp_device.h PipeT): explicit LDS reads, waves 0-3 fetch, scalar ring state, FIFO of 2 - each
// switchable by the ablation hooks above
#ifdef DN_PIPE_ASM_READS
constexpr bool kG48AsmReads = true;
#else
constexpr bool kG48AsmReads = false;
#endif
#ifdef DN_PIPE_LEADER_DMA
constexpr bool kG48LeaderDma = true;
#else
constexpr bool kG48LeaderDma = false;
#endif
using PipeGeo48 = PipeGeometry<kG48AsmReads, kG48LeaderDma, true, kPrefetch>;
template <int WAVES>
using Pipe48 = PipeT<WAVES, PipeGeo48>;

// End of a kernel's prologue where the barrier period is one phase (PipeT::phase_begin): kRingPhases - 1 phases were requested, the
// first period needs phases 0 and 1.  Fetching waves (0-3: four loads per phase) keep their eight youngest loads in flight - VMEM ops
// retire in order, so everything they issued earlier (input rows, mask words) has landed too; waves 4-7 issued no weight loads and
// wait for all they have.  LDS stores of the prologue (bias rows, tables) are waited for as well.  Branch inside the statement.
__device__ __forceinline__ void g48_prologue_wait(unsigned wave) {
  const unsigned fetcher = __builtin_amdgcn_readfirstlane(wave < 4 ? 1u : 0u);
  asm volatile("s_cmp_eq_u32 %0, 0\n\t"
               "s_cbranch_scc1 .Ldn_prol_all%=\n\t"
               "s_waitcnt vmcnt(8) lgkmcnt(0)\n\t"
               "s_branch .Ldn_prol_done%=\n"
               ".Ldn_prol_all%=:\n\t"
               "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
               ".Ldn_prol_done%=:" ::"s"(fetcher) : "scc", "memory");
}

// F = 1: bf16, 2: fp16 (Prec<F> of mlp_device.h): same MFMA rate and layouts
template <int F>
__device__ __forceinline__ f32x4 mfma48(typename Prec<F>::BPiece a, typename Prec<F>::BPiece b, f32x4 c) {
  if constexpr (F == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// how many of the pieces [q0, q1) of the stream are ever fetched: all, except - behind the LAST stage of a pass, which ends at
// position END - the PAD padding pieces nobody consumes
template <bool LAST, int END, int PAD>
constexpr int g48_issued(int q0, int q1) {
  int n = 0;
  for (int q = q0; q < q1; ++q) n += (LAST && q >= END && q < END + PAD) ? 0 : 1;
  return n;
}

// One GEMM stage: NT_OUT 16-row output tiles, KH hidden pieces + KP encoding pieces per tile, three point groups.
// bias_addr: LDS byte address of this lane group's 16 bytes of the stage's bias tile 0; the bias tiles of a tile pass are
// contiguous in stream order, so "the next tile's bias" is the next 64 bytes, except after the last stage of the pass
// (LAST): there it is next_addr (tile 0 of layer1).  Read pipeline (DN_PIPE_ASM_READS): step k of a tile = take A(p),
// three MFMAs, read A(p+2); the next tile's bias read goes out right after the A read of step KT-2, i.e. between A(next
// tile, 0) and A(next tile, 1) - so with a FIFO of P pieces the wait counts are P - 1 everywhere and P at k = KT-1; the
// bias take at k = 0 waits with 1 (one A read was issued after the bias read), which also lands every older A read.
// TRK (fp16 range guard, FwdParams::range_flag): 1 / 2 = fold the bit patterns of this stage's hidden INPUT pieces `bh` into the
// running unsigned 16-bit maximum `*trk` (2: the input carries no ReLU, clear the sign bits first) - a few v_pk_max_u16 per output
// tile, on registers that are live for the whole stage anyway (tracking the freshly converted outputs instead kept them live
// behind a serial chain: 776 spilled registers).  A final pattern >= 0x7C00 is an infinity or a NaN.
// BIAS = false (the backward-data chain): no bias rows - the accumulators start at zero and no bias read rides in the pipeline.
template <int F, int NT_OUT, int KH, int KP, int POS0, bool LAST = false, bool SETTLE = true, int PAD = 0, int PH = kPhasePieces, int TRK = 0,
          bool BIAS = true, class PipeT, class BH, class BP, class Emit>
__device__ __forceinline__ void run_stage48(PipeT& pipe, const BH& bh, BP&& bp, unsigned bias_addr, unsigned next_addr, Emit&& emit,
                                            unsigned* trk = nullptr) {
  constexpr int PT = static_cast<int>(std::extent<BH, 0>::value), KT = KH + KP;   // point groups per wave: 3 (2: the small-launch training instances)
  static_assert(PT == 2 || PT == 3, "bh is [point groups][pieces]");
  static_assert(KT >= 2 || !BIAS, "the bias prefetch distance assumes at least two pieces per tile");
  static_for<NT_OUT>([&](auto nt_c) {
    constexpr int nt = decltype(nt_c)::value;
    f32x4 acc[PT];
    static_for<KT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int pos = POS0 + nt * KT + k;
#if defined(DN_STAMP) && DN_STAMP == 1
      if constexpr (pos % 4 == 0 && pos % kPhasePieces != 0) pipe.template substamp<(pos % kPhasePieces) / 4>();
#endif
      pipe.template at_position<PH, pos>();   // phase boundary (barrier + weight DMA) / mid-phase DMA, if this is one
#ifdef DN_PIPE_ASM_READS
      if constexpr (k == 0) {
        f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (BIAS) b = pipe.template bias_take<1>();   // issued before A(pos + 1): one younger read may stay in flight
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = b;
      }
      // younger reads of ours than A(pos): the other FIFO entries - those that were issued at all (see PAD below) - plus
      // the next tile's bias at k = KT - 1
      constexpr int newer = g48_issued<LAST, POS0 + NT_OUT * KT, PAD>(pos + 1, pos + kPrefetch) + ((BIAS && k == KT - 1) ? 1 : 0);
      const auto a = __builtin_bit_cast(typename Prec<F>::BPiece, pipe.template take<pos, newer>());
#else
      if constexpr (k == 0) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(pipe.ring + (bias_addr - pipe.ring_addr) + nt * 64);
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = b;
      }
      const auto a = __builtin_bit_cast(typename Prec<F>::BPiece, pipe.af[pos % kPrefetch]);
#endif
#ifdef DN_G48_EPI_PIN
      // the previous tile's epilogue (12 conversions / ReLUs, same scheduling region: it follows the sched_barrier below) goes ONE
      // vector instruction per MFMA gap - an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, one 4-5-cycle instruction
      // fits the rest - after DN_G48_EPI_PIN leading MFMAs (distance to the accumulators' last writes: no hazard nops)
      if constexpr (k == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, DN_G48_EPI_PIN, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
#endif
      static_for<PT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (k < KH) acc[t] = mfma48<F>(a, bh[t][k], acc[t]);
        else acc[t] = mfma48<F>(a, bp(t, k - KH), acc[t]);
      });
#ifdef DN_PIPE_ASM_READS
      // PAD padding pieces follow the last stage of a pass (Pipe::skip): a read of one of THOSE would never be consumed, and
      // a fragment nobody consumes is a dead value to the compiler - it reuses the registers while the read is in flight
      if constexpr (g48_issued<LAST, POS0 + NT_OUT * KT, PAD>(pos + kPrefetch, pos + kPrefetch + 1) == 1) pipe.template prefetch<pos>();
#else
      pipe.template prefetch<pos>();
#endif
#ifdef DN_PIPE_ASM_READS
      if constexpr (BIAS && k == KT - 2) {
        if constexpr (nt + 1 < NT_OUT) pipe.template bias_prefetch<(nt + 1) * 64>(bias_addr);
        else if constexpr (LAST) pipe.template bias_prefetch<0>(next_addr);
        else pipe.template bias_prefetch<NT_OUT * 64>(bias_addr);
      }
#else
      __builtin_amdgcn_sched_group_barrier(0x008, PT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<PT>([&](auto t_c) { emit(nt_c, t_c, acc[decltype(t_c)::value]); });
    if constexpr (F == 2 && TRK != 0) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int TOT = PT * KH * 4;   // input dwords of this wave
      constexpr int Q0 = nt * TOT / NT_OUT, Q1 = (nt + 1) * TOT / NT_OUT;
      auto dword = [&](int q) { return __builtin_bit_cast(u32x4, bh[q / (KH * 4)][(q / 4) % KH])[q % 4]; };
      if constexpr (TRK == 2) {
        // inputs that carry sign bits (layer1's output): clear them, unsigned 16-bit maximum of the patterns
        static_for<Q1 - Q0>([&](auto q_c) {
          const unsigned v = dword(Q0 + decltype(q_c)::value);
          // opaque: written as plain max operations the optimiser reassociates the whole kernel's chain into one expression
          // evaluated at the end of the tile loop - every stage's pieces stay live until then (hundreds of spilled registers)
          unsigned t = *trk;
          unsigned tmp;
          asm volatile("v_and_b32 %1, 0x7fff7fff, %2\n\tv_pk_max_u16 %0, %0, %1" : "+v"(t), "=&v"(tmp) : "v"(v));
          *trk = t;
        });
      } else {
        // ReLU outputs: non-negative patterns, +inf, NaN.  v_pk_maximum3_f16 (IEEE maximum: a NaN operand gives NaN) folds TWO
        // input dwords per instruction and orders non-negative fp16 like their bit patterns - half the instructions of the
        // unsigned form (one per dword: ~1,000 per pass beside 3,516 MFMAs, ~3 % of the fp16 instance)
        static_for<(Q1 - Q0 + 1) / 2>([&](auto q_c) {
          constexpr int qa = Q0 + 2 * decltype(q_c)::value, qb = (qa + 1 < Q1) ? qa + 1 : qa;
          const unsigned va = dword(qa), vb = dword(qb);
          unsigned t = *trk;
          asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(t) : "v"(va), "v"(vb));
          *trk = t;
        });
      }
    }
  });
#ifdef DN_PIPE_ASM_READS
  // run-time network shape: no read stays in flight across a stage boundary (control flow merges there: see Pipe::settle);
  // the fixed-shape instances are straight-line code from the top of a tile pass to its end and settle once, there
  if constexpr (SETTLE) pipe.template settle<BIAS>();
#endif
}

// ===== explicit-schedule stage (the fixed-shape W = 256 render instances) ==========================================================
// Measured on the round-3 kernel (profiles/r04_headline_schedule.md): ONE wave of a SIMD running the tile pass alone keeps the matrix
// pipe busy 0.59 of the time (0.79 with no memory operation at all), two waves together 0.77 - whenever the SIMD partner is away (at
// the phase barrier, issuing weight DMAs, in its top-of-tile block) the pipe gets single-wave efficiency, and the compiler's placement
// of a tile's epilogue is what a wave alone cannot hide: twelve conversions / ReLUs packed three to an MFMA gap behind `s_nop 3`
// hazard padding, an `s_nop 0` in front of every piece's first MFMA.  Here every MFMA, conversion and ReLU of the tile loop is its
// own `asm volatile` statement - volatile statements keep their source order, so the source order IS the instruction order:
//   piece k of tile nt:   wait A(pos) | MFMA g0 | op | MFMA g1 | op | MFMA g2 | (ops) | read A(pos + 2)
// and the epilogue of tile nt - 1 (or of the previous stage's last tile: `pend`) rides one vector instruction per MFMA gap (an
// MFMA holds the SIMD's vector issue for 8 of its 16 cycles; a 4-5-cycle instruction fits the rest) in the first blocks of tile nt,
// reading the finished accumulators (`pacc`) while tile nt accumulates into fresh registers.  No hazard padding is needed by
// construction (and none is inserted: the compiler sees no MFMA): an accumulator is read >= 4 instructions and >= 48 cycles after the
// MFMA that finished it (a 4-pass MFMA's result is due after 7 wait states), an MFMA re-reads its accumulator after two others,
// a wait and a read.  Group order of the ops: g0, g1, g2 - group 2, whose last MFMA is the youngest, is touched last.
template <int F>
__device__ __forceinline__ void mfma48_first(f32x4& d, const f32x4& a, const typename Prec<F>::BPiece& b, const f32x4& c) {
  if constexpr (F == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
// a block's FIRST MFMA carries the counted wait for its A fragment in the same statement (as two statements hipcc puts an `s_nop 0`
// between them: it takes the wait statement for an unknown writer of the fragment registers); the fragment is an in-out operand, so the
// compiler cannot touch it between its read statement and this one (tests/test_asm_hazards.py replays the counter on the result)
#ifdef DN_EXP_NOWAIT
#define DN_XS_WAIT "; no wait %[n]\n\t"
#else
#define DN_XS_WAIT "s_waitcnt lgkmcnt(%[n])\n\t"
#endif
template <int F, int NEWER>
__device__ __forceinline__ void mfma48_first_w(f32x4& d, f32x4& a, const typename Prec<F>::BPiece& b, const f32x4& c) {
  if constexpr (F == 1) asm volatile(DN_XS_WAIT "v_mfma_f32_16x16x32_bf16 %[d], %[a], %[b], %[c]" : [d] "=&v"(d), [a] "+v"(a) : [b] "v"(b), [c] "v"(c), [n] "n"(NEWER));
  else asm volatile(DN_XS_WAIT "v_mfma_f32_16x16x32_f16 %[d], %[a], %[b], %[c]" : [d] "=&v"(d), [a] "+v"(a) : [b] "v"(b), [c] "v"(c), [n] "n"(NEWER));
}
template <int F, int NEWER>
__device__ __forceinline__ void mfma48_acc_w(f32x4& d, f32x4& a, const typename Prec<F>::BPiece& b) {
  if constexpr (F == 1) asm volatile(DN_XS_WAIT "v_mfma_f32_16x16x32_bf16 %[d], %[a], %[b], %[d]" : [d] "+v"(d), [a] "+v"(a) : [b] "v"(b), [n] "n"(NEWER));
  else asm volatile(DN_XS_WAIT "v_mfma_f32_16x16x32_f16 %[d], %[a], %[b], %[d]" : [d] "+v"(d), [a] "+v"(a) : [b] "v"(b), [n] "n"(NEWER));
}
template <int F>
__device__ __forceinline__ void mfma48_acc(f32x4& d, const f32x4& a, const typename Prec<F>::BPiece& b) {
  if constexpr (F == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// op S of a hidden tile's epilogue (emit48, one instruction at a time): S = 0 .. 5 convert the pair (2d, 2d + 1) of group S / 2,
// S = 6 .. 11 (RELU) clamp the packed pair at zero as signed 16-bit integers.  NT: the tile's index in its stage (see emit48).
template <int F, bool RELU, int NT, int S, class BSet>
__device__ __forceinline__ void hidden_op48(const f32x4 (&pacc)[3], BSet& bo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  static_assert(S >= 0 && S < (RELU ? 12 : 6), "op index");
  constexpr int t = (S % 6) / 2, d = S % 2, e = (NT & 1) * 2 + d;
  u32x4 w = __builtin_bit_cast(u32x4, bo[t][NT / 2]);
  if constexpr (S < 6) {
    const float x = pacc[t][2 * d], y = pacc[t][2 * d + 1];
    unsigned o;
    if constexpr (F == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(y));
    w[e] = o;
  } else {
    unsigned o = w[e];
    asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(o));
    w[e] = o;
  }
  bo[t][NT / 2] = __builtin_bit_cast(typename Prec<F>::BPiece, w);
}
// an accumulator element into a plain register (the heads' output rows), as a pinned statement like every other reader of `pacc`
__device__ __forceinline__ void pick_op48(float& dst, float src) { asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "v"(src)); }

// How the NP pending ops of the previous tile are laid over the blocks of a tile of KT pieces: the first E blocks, PER to a block;
// DEADLINE = the first block whose MFMAs read what the ops produce (the last tile of a stage feeds piece (NT_OUT - 1) / 2 of the next).
constexpr int xs_blocks(int np, int kt, int deadline) {
  int e = kt < 6 ? kt : 6;
  if (deadline < e) e = deadline;
  if (e < 1) e = 1;
  (void)np;
  return e;
}
constexpr int xs_per(int np, int e) { return (np + e - 1) / e; }
// pieces of a fixed-shape tile pass in front of trunk stage i (layer1 + the trunk stages before it)
constexpr int xs_trunk_pos(int i, int nt, int kh, int kxp, unsigned mask) {
  int p = nt * kxp;
  for (int j = 0; j < i; ++j) p += nt * (kh + (((mask >> j) & 1u) ? kxp : 0));
  return p;
}

// ---- in-stage xyz encoding (mlp_fused48.hip, OVLP = 2): the work of a lane's 48 slots (three point groups x 16 slots, pe_value op for
// op) as a flat queue of single instructions, dealt to the blocks of the view-direction stage behind the pending ops.
// Slot u of a group: MUL (x f'), FRACT, ADD (phase), SIN, [SEL: slots 0-2, identity columns of lane group 0], [CVT: odd elements, two
// values -> one dword of the piece], [WRITE: element 7, the finished piece -> LDS stash].
constexpr int enc_slot_len(int u) { return 4 + (u < 3 ? 1 : 0) + ((u % 8) % 2) + ((u % 8) == 7 ? 1 : 0); }
constexpr int enc_first(int m) {   // queue position of slot m's first op (m = 16 group + u)
  int q = 0;
  for (int i = 0; i < m; ++i) q += enc_slot_len(i % 16);
  return q;
}
constexpr int kEncSlots = 48;
constexpr int kEncOps = enc_first(kEncSlots);
constexpr int enc_slot_of(int q) {
  int m = 0;
  while (m < kEncSlots && q >= enc_slot_len(m % 16)) { q -= enc_slot_len(m % 16); ++m; }
  return m;
}
enum { kEncMul = 0, kEncFract = 1, kEncAdd = 2, kEncSin = 3, kEncSel = 4, kEncCvt = 5, kEncWrite = 6 };
constexpr int enc_kind(int u, int idx) {
  if (idx < 4) return idx;
  int j = 4;
  if (u < 3) { if (idx == j) return kEncSel; ++j; }
  if ((u % 8) % 2) { if (idx == j) return kEncCvt; ++j; }
  return kEncWrite;
}
// a block of a KT-piece tile takes 3 ops (one per MFMA gap, beside the two pending ops of the tile's first six blocks) or 4 (op-free blocks)
constexpr int enc_cap(int b, int kt) { return (b % kt) >= 6 ? 4 : 3; }
constexpr int enc_begin(int b, int kt) {
  int q = 0;
  for (int i = 0; i < b; ++i) q += enc_cap(i, kt);
  return q;
}
// the table entry of slot m is read kEncLead queue positions ahead of its MUL: more than a block's capacity, so at least one counted
// wait of the A-fragment pipeline lies in between (it retires every older LDS read); slot 0's is read before the stage
constexpr int kEncLead = 5;
constexpr int enc_table_slot_at(int q) {   // the slot whose table read rides at queue position q, or -1
  for (int m = 1; m < kEncSlots; ++m)
    if (enc_first(m) - kEncLead == q) return m;
  return -1;
}

// One GEMM stage, explicit schedule.  NOPS: ops per tile of THIS stage, run by ops(nt_c, s_c) on `pacc` during the following tile;
// PX: the barrier period in pieces (Pipe48::xs_period_begin).  PEND_N / PEND_BY: the ops the caller still owes for the previous stage's last tile (pend(s_c)) and the block of this stage's
// first tile by which they must be done.  On return `pacc` holds the last tile's accumulators and the caller owes ITS NOPS ops.
// TRK: as run_stage48 (fp16 range tracker on the stage's input pieces), two dwords per instruction, in the op-free blocks.
// Hook: other work riding in the stage's blocks - hook(block_c, where_c) is called behind the ops of MFMA gap `where` = 0, 1, 2 of
// block `block` = nt * KT + k, and with where = 3 just BEFORE the block's A-fragment read: an LDS operation issued there is older
// than that read, so the next block's counted wait (at most ONE younger read in flight) has retired it - no wait of its own.
struct XsNoHook {
  template <class B, class W>
  __device__ __forceinline__ void operator()(B, W) const {}
};
template <int F, int PX, int NT_OUT, int KH, int KP, int POS0, bool LAST, int PAD, int NOPS, int PEND_N, int PEND_BY, int TRK = 0, class PipeT, class BH,
          class BP, class Ops, class Pend, class Hook = XsNoHook>
__device__ __forceinline__ void run_stage48x(PipeT& pipe, const BH& bh, BP&& bp, unsigned bias_addr, unsigned next_addr, f32x4 (&pacc)[3],
                                             Ops&& ops, Pend&& pend, unsigned* trk = nullptr, Hook&& hook = Hook{}) {
  constexpr int PT = 3, KT = KH + KP;
  static_assert(KT >= 2 && KH >= 1, "the bias prefetch distance assumes at least two pieces per tile; a tile's first piece is a hidden piece");
  static_for<NT_OUT>([&](auto nt_c) {
    constexpr int nt = decltype(nt_c)::value;
    constexpr int NP = nt == 0 ? PEND_N : NOPS;
    constexpr int E = xs_blocks(NP, KT, nt == 0 ? PEND_BY : KT);
    constexpr int PER = xs_per(NP, E);
    static_assert(PER * E >= NP, "every pending op has a slot");
    // fp16 tracker: this tile's share of the stage's input dwords, two per instruction, after the pending ops' blocks
    constexpr int TOT = (F == 2 && TRK != 0) ? PT * KH * 4 : 0;
    constexpr int Q0 = nt * TOT / NT_OUT, Q1 = (nt + 1) * TOT / NT_OUT;
    constexpr int NTRK = (TRK == 2) ? (Q1 - Q0) : (Q1 - Q0 + 1) / 2;
    constexpr int TB = KT - E > 0 ? KT - E : 1;                       // blocks left for them (else: all in the last block)
    constexpr int TPER = (NTRK + TB - 1) / TB;
    auto run_op = [&](auto s_c) {
      constexpr int sidx = decltype(s_c)::value;
      if constexpr (sidx < NP) {
        if constexpr (nt == 0) pend(s_c);
        else ops(std::integral_constant<int, nt - 1>{}, s_c);
      }
    };
    auto run_trk = [&](auto q_c) {
      constexpr int qi = decltype(q_c)::value;
      if constexpr (qi < NTRK) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        auto dword = [&](int q) { return __builtin_bit_cast(u32x4, bh[q / (KH * 4)][(q / 4) % KH])[q % 4]; };
        unsigned tv = *trk;
        if constexpr (TRK == 2) {
          const unsigned v = dword(Q0 + qi);
          unsigned tmp;
          asm volatile("v_and_b32 %1, 0x7fff7fff, %2\n\tv_pk_max_u16 %0, %0, %1" : "+v"(tv), "=&v"(tmp) : "v"(v));
        } else {
          constexpr int qa = Q0 + 2 * qi, qb = (qa + 1 < Q1) ? qa + 1 : qa;
          const unsigned va = dword(qa), vb = dword(qb);
          asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(tv) : "v"(va), "v"(vb));
        }
        *trk = tv;
      }
    };
    f32x4 acc[PT];
    static_for<KT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int pos = POS0 + nt * KT + k;
      pipe.template at_position_xs<PX, pos>();
      f32x4 b = {0.0f, 0.0f, 0.0f, 0.0f};
      if constexpr (k == 0) b = pipe.template bias_take<1>();
      constexpr int newer = g48_issued<LAST, POS0 + NT_OUT * KT, PAD>(pos + 1, pos + kPrefetch) + ((k == KT - 1) ? 1 : 0);
      f32x4& a = pipe.af[pos % kPrefetch];
      static_for<PT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (t == 0) {   // (with the wait for A(pos): at most `newer` of our younger reads stay in flight)
          if constexpr (k == 0) mfma48_first_w<F, newer>(acc[t], a, bh[t][0], b);
          else if constexpr (k < KH) mfma48_acc_w<F, newer>(acc[t], a, bh[t][k]);
          else mfma48_acc_w<F, newer>(acc[t], a, bp(t, k - KH));
        } else if constexpr (k == 0) mfma48_first<F>(acc[t], a, bh[t][0], b);
        else if constexpr (k < KH) mfma48_acc<F>(acc[t], a, bh[t][k]);
        else mfma48_acc<F>(acc[t], a, bp(t, k - KH));
        // the gap behind this MFMA: one op (behind the third MFMA: whatever is left of this block's share)
        if constexpr (k < E) {
          if constexpr (t < 2) run_op(std::integral_constant<int, k * PER + t>{});
          else static_for<(PER > 2 ? PER - 2 : 0)>([&](auto j_c) { run_op(std::integral_constant<int, k * PER + 2 + decltype(j_c)::value>{}); });
        } else if constexpr (NTRK > 0) {
          constexpr int kb = (KT - E > 0) ? k - E : 0;
          if constexpr (t < 2) run_trk(std::integral_constant<int, kb * TPER + t>{});
          else static_for<(TPER > 2 ? TPER - 2 : 0)>([&](auto j_c) { run_trk(std::integral_constant<int, kb * TPER + 2 + decltype(j_c)::value>{}); });
        }
        hook(std::integral_constant<int, nt * KT + k>{}, t_c);
      });
      if constexpr (KT - E <= 0 && NTRK > 0 && k == KT - 1)   // no op-free block: the tracker's ops behind the tile's last MFMA
        static_for<NTRK>([&](auto q_c) { run_trk(q_c); });
      pipe.template xs_after_piece<PX, pos>();
      hook(std::integral_constant<int, nt * KT + k>{}, std::integral_constant<int, 3>{});
      if constexpr (g48_issued<LAST, POS0 + NT_OUT * KT, PAD>(pos + kPrefetch, pos + kPrefetch + 1) == 1) pipe.template prefetch<pos>();
      if constexpr (k == KT - 2) {
        if constexpr (nt + 1 < NT_OUT) pipe.template bias_prefetch<(nt + 1) * 64>(bias_addr);
        else if constexpr (LAST) pipe.template bias_prefetch<0>(next_addr);
        else pipe.template bias_prefetch<NT_OUT * 64>(bias_addr);
      }
    });
#pragma unroll
    for (int t = 0; t < PT; ++t) pacc[t] = acc[t];
  });
}

// rows 4g..4g+3 of output tile NT -> elements (NT & 1) * 4 .. + 3 of B piece NT / 2 (g48_hidden_col)
// (The training forward with 8-bit saved tensors runs the same two instructions per dword as the inference kernels: its saved e4m3
// bytes are formed from the 16-bit pairs by v_cvt_scalef32_pk_fp8_bf16, which SATURATES at 448 when MODE.FP16_OVFL is set - measured,
// scripts/micro/cvt_sat_probe.hip; in the default mode an overflow converts to NaN, and rounds 2-3 clamped every stage output with a
// v_med3_f32 per element first: a third of this epilogue's instructions.  mlp_fused48_kernel.h sets the bit for SAVE instances.)
template <int F, bool RELU, int NT, class BO>
__device__ __forceinline__ void emit48(const f32x4& acc, BO& bo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef typename Prec<F>::Elem e16x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w = __builtin_bit_cast(u32x4, bo[NT / 2]);
#ifdef DN_EXP_NOEPI   // timing experiment only: raw accumulator bits as the piece's dwords - no convert, no ReLU (1: two moves per tile, 2: one)
  w[(NT & 1) * 2] = __builtin_bit_cast(unsigned, acc[0]);
  if (DN_EXP_NOEPI < 2) w[(NT & 1) * 2 + 1] = __builtin_bit_cast(unsigned, acc[2]);
  bo[NT / 2] = __builtin_bit_cast(typename Prec<F>::BPiece, w);
  return;
#endif
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    f32x2 f = {acc[2 * d], acc[2 * d + 1]};
    e16x2 v = __builtin_convertvector(f, e16x2);  // one packed convert
    if constexpr (RELU) {  // a negative bf16 / fp16 is a negative int16 (mlp_device.h make_piece)
      s16x2 bits = __builtin_bit_cast(s16x2, v);
      const s16x2 zero = {0, 0};
      bits = __builtin_elementwise_max(bits, zero);
      v = __builtin_bit_cast(e16x2, bits);
    }
    w[(NT & 1) * 2 + d] = __builtin_bit_cast(unsigned, v);
  }
  bo[NT / 2] = __builtin_bit_cast(typename Prec<F>::BPiece, w);
}

}  // namespace dn

