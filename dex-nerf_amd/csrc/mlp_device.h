// Device-side building blocks shared by the fused inference kernel (mlp_fused.hip) and the training kernels
// (mlp_train.hip): MFMA piece types, the LDS-DMA weight pipeline, the per-stage MFMA loop and the in-register
// positional encoding.  See mlp_layout.h for the stream layout and DESIGN.md for the design.
#pragma once
#include <cstdlib>
#include <utility>

#include "mlp_layout.h"

#include "dn_ablation.h"

namespace dn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr int kRingPhases = 5;
constexpr int kSlotBytes = kPhasePieces * kPieceBytes;  // 16 KiB
constexpr int kRingBytes = kRingPhases * kSlotBytes;    // 80 KiB
#ifndef DN_PREFETCH
#define DN_PREFETCH 4
#endif
constexpr int kPrefetch = DN_PREFETCH;                  // A-fragment pieces read ahead of the MFMA that uses them
constexpr int kInRows = 10;                             // per-wave input staging rows

// arithmetic mode (the template parameter is still called BF16: 0 = fp32, 1 = bf16, 2 = fp16; non-zero = 16-bit MFMA)
template <int BF16> struct Prec;
template <> struct Prec<1> {
  using BPiece = bf16x8;
  using Elem = __bf16;
  static constexpr int EPP = 8;    // k-values (elements) per lane per piece
  static constexpr int PPT = 2;    // pieces per 32-row hidden tile
};
template <> struct Prec<2> {   // fp16: same MFMA rate and layouts as bf16, 10-bit mantissa (render-only mode)
  using BPiece = f16x8;
  using Elem = _Float16;
  static constexpr int EPP = 8;
  static constexpr int PPT = 2;
};
template <> struct Prec<0> {
  using BPiece = f32x4;
  using Elem = float;
  static constexpr int EPP = 4;
  static constexpr int PPT = 4;
};
// Geometry: PT point-tiles (of 32 points) per wave.  bf16 default: PT=1 -> 8 waves x 32 points, two waves per
// SIMD.  bf16 PT=2 -> 4 waves x 64 points, one wave per SIMD (every A fragment read from LDS feeds two MFMAs;
// halves the LDS read traffic but a lone wave per SIMD hides no stall - measured 4 % slower).  fp32: PT=1,
// 4 waves x 32 points, one wave per SIMD.
template <int BF16, int PT> constexpr int waves_of() { return (BF16 && PT == 1) ? 8 : 4; }

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
  // training forward (SAVE kernels): saved activations [tile32][act_pieces][64 lanes][16 B], ReLU bit masks
  // [tile32][mask_words][64 lanes][16 B]; slots as in TrainLayout
  char* act;
  char* masks;
  int act_pieces, mask_words;   // (SAVE == 2, 8-bit saved tensors: act_pieces counts 1 KiB units = pairs of pieces)
  int slot_xyz, slot_dir, slot_layer1, slot_trunk0, slot_feat, slot_dirout;
  int save8;   // training forward: store the saved pieces at 8 bits (DN_PREC_BF16_S8)
  // fp16 inference (48-point kernel): device word that receives +1 per wave whose hidden activations left fp16's range (an inf
  // or NaN among the converted stage outputs), or NULL.  The raw output alone does not show it: the matrix pipe's NaN has its
  // sign bit set, so the integer-max ReLU turns it into 0 and the rest of the network computes finite garbage.
  unsigned* range_flag;
};

// ---- weight pipeline: LDS ring fed by LDS-DMA -------------------------------------------------------
// Ring of 5 x 16 KiB phases.  At the barrier that opens phase p every wave has waited for its own DMAs of
// phases <= p+1, so after the barrier phases p AND p+1 are fully landed: the A-fragment read stream (a FIFO of
// kPrefetch pieces per wave) runs continuously across phase boundaries.  Phases p+2, p+3 stay in flight
// (counted vmcnt, never 0 in the loop); phase p+4 is issued into the slot phase p-1 just vacated.
// Extra VMEM ops (input DMAs, the output store) are younger or older than the DMAs a wait must cover and,
// because VMEM ops retire in order, can only make a counted wait stricter, never weaker.
// One class, two geometries.  The geometry type says which form of the pipeline a kernel family runs:
//   ASM_READS     the A-fragment / bias LDS reads and their COUNTED waits are opaque asm statements (below)
//   LEADER_DMA    waves 0-3 fetch the whole weight stream, four pieces each per phase; waves 4-7 issue MFMAs only (mlp_stage48.h)
//   SCALAR_STATE  the wave-uniform ring bookkeeping is pinned to SGPRs
//   PREFETCH      depth of the A-fragment FIFO (= the translation unit's kPrefetch)
// PipeGeo32 (here): the 32-points-per-wave kernels of mlp_fused.hip / mlp_train.hip; PipeGeo48 (mlp_stage48.h): the 48-point kernels.
// Members only one form uses cost nothing (a Pipe lives in registers; unused fields are never materialised); member functions of a
// class template are instantiated only where they are called.
template <bool ASM_READS_, bool LEADER_DMA_, bool SCALAR_STATE_, int PREFETCH_>
struct PipeGeometry {
  static constexpr bool ASM_READS = ASM_READS_, LEADER_DMA = LEADER_DMA_, SCALAR_STATE = SCALAR_STATE_;
  static constexpr int PREFETCH = PREFETCH_;
};
template <int WAVES, class G>
struct PipeT {
  static_assert(G::PREFETCH == kPrefetch, "the FIFO depth of a geometry is its translation unit's kPrefetch");
  static constexpr int PER_WAVE = kPhasePieces / WAVES;
  char* ring;           // LDS
  unsigned ring_addr;   // its 32-bit LDS byte address (for M0)
  const char* wsrc;     // global pieces (wave-uniform pointer)
  unsigned total_bytes; // stream length in bytes
  unsigned q_issue;     // byte offset of the next phase to DMA (wave-uniform)
  unsigned slot_wr, slot_nxt;
  unsigned wave;
  unsigned pend_src, pend_dst;  // this wave's DMA of the phase being issued (wave-uniform byte offsets)
  const char* rd_cur;   // LDS read pointers (+ lane*16) of the current and the next phase
  const char* rd_nxt;
  unsigned lane16;
  f32x4 af[kPrefetch];  // A-fragment FIFO: af[pos % kPrefetch] holds piece `pos` when it is consumed
  // ---- ASM_READS: explicit LDS read pipeline (mlp_fused48.hip).  hipcc's own waitcnt insertion turns a depth-2 software pipeline of
  // ds_read_b128 into "issue the read for piece p+2, then s_waitcnt lgkmcnt(0)": every other piece, and every tile's bias
  // read, exposed a full LDS round trip in front of the MFMAs (r02 PMC: waves parked in s_waitcnt 39 % of their cycles).
  // Here the reads and their COUNTED waits are opaque asm statements: a consumer waits with lgkmcnt(N), N = the number of
  // our own reads issued after the one it needs (LDS returns in order; any compiler-issued LDS access in between only
  // makes the wait stricter, never weaker).  The value flows read-asm -> wait-asm ("+v") -> MFMA, so the compiler cannot
  // use a fragment before its wait; tests/test_asm_hazards.py checks in the disassembly that nothing touches a fragment
  // register between its ds_read and its wait.
  unsigned rda_cur;            // 32-bit LDS byte address (+ lane * 16) of the current phase slot; the next slot's is formed
                               // where it is needed (the last kPrefetch pieces of a phase) from the scalar slot base
#ifdef DN_STAMP   // diagnostic build only: where a wave's cycles go at the phase boundaries (s_memtime, accumulated in SGPRs)
  unsigned st_vm = 0, st_bar = 0, st_dma = 0, st_seg = 0, st_n = 0, st_prev = 0;
  unsigned st_sub[4] = {0, 0, 0, 0}, st_last = 0;   // MFMA time of the four quarters of a phase (4 pieces each)
  unsigned st_top = 0, st_top_pending = 0;          // end of a pass (tail pieces, output store, next tile's encodings) up to the next phase
  __device__ __forceinline__ void pass_end() {       // after the last full quarter of a pass
    const unsigned t = stamp();
    if (st_n) st_tail += t - st_last;     // light mode: st_tail = the whole pass minus its top
    st_last = t;
    st_top_pending = 1;
  }
  unsigned st_tail = 0;
  unsigned st_cls[3] = {0, 0, 0};                    // DN_STAMP == 4: time per stage class (layer1 / trunk / heads)
  template <int CLS>
  __device__ __forceinline__ void stage_end() {
    const unsigned t = stamp();
    st_cls[CLS] += t - st_last;
    st_last = t;
  }
  template <int I>
  __device__ __forceinline__ void substamp() {       // at piece 4 * I of a phase, I = 1..3
    const unsigned t = stamp();
    st_sub[I - 1] += t - st_last;
    st_last = t;
  }
  __device__ __forceinline__ unsigned stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return __builtin_amdgcn_readfirstlane(static_cast<unsigned>(t));
  }
#endif
  unsigned slot_cur_base;      // (scalar) LDS byte address of the slot of the NEXT phase (becomes rda_cur at phase_begin)
  f32x4 bias_nxt;              // bias rows of the NEXT 16-row tile, read two pieces ahead of its first MFMA
#ifdef DN_EXP_REGSTAGE
  f32x4 stage[PER_WAVE];
  unsigned stage_dst;
  bool have_stage = false;
#endif

  // DMA of one phase = PER_WAVE consecutive pieces per wave, as ONE opaque asm statement: SGPR-base form of
  // global_load_lds (32-bit lane offset), M0 saved/restored inside the statement, and a wave-uniform skip
  // branch *inside* the asm so the compiler sees straight-line code (a C++ branch here splits every phase into
  // basic blocks and costs dozens of spilled registers).  The instruction offset advances both the global and
  // the LDS address.  hipcc does not count these in its own waitcnt bookkeeping - the counted waits are ours.
  // Hazards the recognizer cannot see inside the statement: the SGPR base comes from v_readfirstlane (VALU writes SGPR ->
  // VMEM reads it: 5 wait states = the five scalar instructions ahead of the load, s_nop 1 for margin) and M0 is
  // written one instruction + nop before the LDS-DMA reads it.
  __device__ __forceinline__ void dma_phase(unsigned src_off, unsigned dst_off, unsigned go) {
#ifndef DN_EXP_NODMA
    // every "s" operand must be provably wave-uniform: readfirstlane them (they are uniform by construction)
    const unsigned long long src_bits = reinterpret_cast<unsigned long long>(wsrc + src_off);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(src_bits >> 32));
    const char* src = reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
    const unsigned lds = __builtin_amdgcn_readfirstlane(ring_addr + dst_off);
    go = __builtin_amdgcn_readfirstlane(go);
    unsigned keep;
    if constexpr (PER_WAVE == 2) {
      asm volatile(
          "s_cmp_lg_u32 %[go], 0\n\t"
          "s_cbranch_scc0 .Ldn_dma_skip%=\n\t"
          "s_mov_b32 %[keep], m0\n\t"
          "s_mov_b32 m0, %[lds]\n\t"
          "s_nop 1\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase]\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase] offset:1024\n\t"
          "s_mov_b32 m0, %[keep]\n"
          ".Ldn_dma_skip%=:"
          : [keep] "=&s"(keep)
          : [go] "s"(go), [lds] "s"(lds), [voff] "v"(lane16), [sbase] "s"(src)
          : "memory", "scc");
    } else {
      asm volatile(
          "s_cmp_lg_u32 %[go], 0\n\t"
          "s_cbranch_scc0 .Ldn_dma_skip%=\n\t"
          "s_mov_b32 %[keep], m0\n\t"
          "s_mov_b32 m0, %[lds]\n\t"
          "s_nop 1\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase]\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase] offset:1024\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase] offset:2048\n\t"
          "global_load_lds_dwordx4 %[voff], %[sbase] offset:3072\n\t"
          "s_mov_b32 m0, %[keep]\n"
          ".Ldn_dma_skip%=:"
          : [keep] "=&s"(keep)
          : [go] "s"(go), [lds] "s"(lds), [voff] "v"(lane16), [sbase] "s"(src)
          : "memory", "scc");
    }
#endif
  }

  __device__ __forceinline__ void advance_issue() {
#ifdef DN_EXP_ROTATE  // experiment: which wave fetches which pieces of a phase rotates with the workgroup's index in its XCD
    const unsigned who = (wave + (blockIdx.x >> 3)) % WAVES;
#else
    const unsigned who = wave;
#endif
    if constexpr (G::LEADER_DMA) {
      // waves 0-3 fetch the whole phase, four pieces each (two right after the barrier, two at mid-phase); waves 4-7 none
      pend_src = q_issue + who * (2 * PER_WAVE * kPieceBytes);
      pend_dst = slot_wr * kSlotBytes + who * (2 * PER_WAVE * kPieceBytes);
    } else {
      pend_src = q_issue + who * (PER_WAVE * kPieceBytes);
      pend_dst = slot_wr * kSlotBytes + who * (PER_WAVE * kPieceBytes);
    }
    q_issue += kSlotBytes;
    if (q_issue >= total_bytes) q_issue = 0;
    slot_wr = (slot_wr + 1 == kRingPhases) ? 0 : slot_wr + 1;
    if constexpr (G::SCALAR_STATE) {  // pin the (wave-uniform) ring state to SGPRs: hipcc otherwise keeps it in ~8 VGPRs (mlp_fused48.hip: 8 -> 0 spills)
      q_issue = __builtin_amdgcn_readfirstlane(q_issue);
      slot_wr = __builtin_amdgcn_readfirstlane(slot_wr);
      pend_src = __builtin_amdgcn_readfirstlane(pend_src);
      pend_dst = __builtin_amdgcn_readfirstlane(pend_dst);
    }
  }

  __device__ __forceinline__ void issue_phase() {  // prologue only
    advance_issue();
    if constexpr (G::LEADER_DMA) {
      dma_phase(pend_src, pend_dst, wave < 4 ? 1u : 0u);
      dma_phase(pend_src + PER_WAVE * kPieceBytes, pend_dst + PER_WAVE * kPieceBytes, wave < 4 ? 1u : 0u);
    } else {
      dma_phase(pend_src, pend_dst, 1u);
    }
  }

  // Called at every 16-piece boundary of the (compile-time laid out) consumption sequence.
  // With two waves per SIMD (bf16 build) the DMA issue is split: waves 0-3 issue right after the barrier, waves
  // 4-7 eight pieces later (mid_phase), so a wave's DMA-issue time (an LDS-DMA costs ~60-180 issue cycles) is
  // covered by its SIMD partner's MFMAs instead of both stalling the matrix pipe together.
  __device__ __forceinline__ void phase_begin() {
    // (kRingPhases-3) younger phases may stay outstanding; lgkmcnt(0): this wave's LDS reads of the previous
    // phase are complete before its slot is recycled (and the FIFO entries for this phase have arrived).
    static_assert(kRingPhases == 5, "the counted waits below assume two younger phases in flight");
#ifdef DN_EXP_SHALLOW  // ablation: only one younger phase in flight
    if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#elif defined(DN_EXP_LOOSEWAIT)  // timing experiment only (UNSAFE: lets 8 more VMEM ops stay outstanding)
#if DN_EXP_LOOSEWAIT >= 2
    if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
#else
    if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
#endif
#else
#ifdef DN_STAMP   // (diagnostic builds of the 48-point kernels)
    unsigned st0 = 0, st1 = 0, st2 = 0;
#endif
    if constexpr (G::ASM_READS) {
      // No LDS wait here: every read of the slot being recycled (phase p-1) was waited for by the take() in front of its
      // MFMAs, which precede this point in program order; the reads still in flight belong to phases p and p+1.
      static_assert(!G::ASM_READS || PER_WAVE == 2, "the asm-read pipeline is the 8-wave geometry");
#if defined(DN_STAMP) && DN_STAMP == 2   // light mode: only the top-of-tile time (two stamps per pass)
      if (st_top_pending) { st_top += stamp() - st_last; st_top_pending = 0; ++st_n; }
#elif defined(DN_STAMP) && DN_STAMP == 4   // stage mode: the top-of-tile ends at the first phase boundary of a pass
      if (st_top_pending) { const unsigned t = stamp(); st_top += t - st_last; st_last = t; st_top_pending = 0; ++st_n; }
#elif defined(DN_STAMP) && DN_STAMP == 3   // barrier mode: arrival / release of every phase barrier (two stamps per phase)
      st0 = stamp();
      if (st_top_pending) { st_top += st0 - st_last; st_top_pending = 0; }
      else if (st_n) st_seg += st0 - st_prev;
#elif defined(DN_STAMP)
      st0 = stamp();
      if (st_top_pending) { st_top += st0 - st_last; st_top_pending = 0; }
      else if (st_n) { st_seg += st0 - st_prev; st_sub[3] += st0 - st_last; }
#endif
      // a fetching wave of the LEADER_DMA form has four DMAs per phase, otherwise every wave two: two younger phases stay in flight
      if constexpr (G::LEADER_DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#if defined(DN_STAMP) && DN_STAMP == 1
      st1 = stamp();
      st_vm += st1 - st0;
#endif
    } else {
      if constexpr (PER_WAVE == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    }
#endif
#ifdef DN_EXP_REGSTAGE
    // experiment: weights through registers (global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA
    if (have_stage) {
#pragma unroll
      for (int e = 0; e < PER_WAVE; ++e)
        *reinterpret_cast<f32x4*>(ring + stage_dst + e * kPieceBytes + lane16) = stage[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    advance_issue();
#pragma unroll
    for (int e = 0; e < PER_WAVE; ++e)
      stage[e] = *reinterpret_cast<const f32x4*>(wsrc + pend_src + e * kPieceBytes + lane16);
    stage_dst = pend_dst;
    have_stage = true;
#else
#ifndef DN_EXP_NOBARRIER   // timing experiment only (UNSAFE: no cross-wave ordering of ring slots)
    __builtin_amdgcn_s_barrier();
#endif
#if defined(DN_STAMP) && DN_STAMP == 1
    st2 = stamp();
    st_bar += st2 - st1;
#elif defined(DN_STAMP) && DN_STAMP == 3
    st_prev = stamp();
    st_bar += st_prev - st0;
    st_last = st_prev;
    ++st_n;
#endif
#if defined(DN_G48_PRIO) && DN_G48_PRIO == 2   // first half of a phase: the younger waves (4-7) lead, second half: the older ones
    if (wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
    advance_issue();
    dma_phase(pend_src, pend_dst, (WAVES == 4 || wave < 4) ? 1u : 0u);
#if defined(DN_STAMP) && DN_STAMP == 1
    st_prev = stamp();
    st_dma += st_prev - st2;
    st_last = st_prev;
    ++st_n;
#endif
#endif
    slot_nxt = (slot_nxt + 1 == kRingPhases) ? 0 : slot_nxt + 1;
    if constexpr (G::ASM_READS) {
      rda_cur = slot_cur_base + lane16;
      slot_cur_base = __builtin_amdgcn_readfirstlane(ring_addr + slot_nxt * kSlotBytes);   // now the NEXT phase's slot
    } else {
      rd_cur = rd_nxt;
      rd_nxt = ring + slot_nxt * kSlotBytes + lane16;
    }
  }

  __device__ __forceinline__ void mid_phase() {
#if defined(DN_G48_PRIO) && DN_G48_PRIO == 2
    if (wave >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
#endif
#ifndef DN_EXP_REGSTAGE
#if defined(DN_STAMP) && DN_STAMP == 1
    const unsigned m0 = stamp();
#endif
    if constexpr (WAVES == 8 && G::LEADER_DMA) dma_phase(pend_src + PER_WAVE * kPieceBytes, pend_dst + PER_WAVE * kPieceBytes, wave < 4 ? 1u : 0u);
    else if constexpr (WAVES == 8) dma_phase(pend_src, pend_dst, wave >= 4 ? 1u : 0u);
#if defined(DN_STAMP) && DN_STAMP == 1
    const unsigned m1 = stamp();
    st_dma += m1 - m0;
    st_prev += m1 - m0;   // keep the DMA issue out of the MFMA-segment figure
    st_last = m1;
#endif
#endif
  }

  // (the forms below - two-phase barrier period, spread fetch - are the ASM_READS + LEADER_DMA geometry's: nothing else calls them)
  // ---- one workgroup barrier per TWO phases (PH = 32: the fixed-shape W = 256 instance, where the parity of a phase is a
  // compile-time position).  The barrier is the one cost of the ring that cannot be overlapped: the SIMD partners do not share
  // the matrix pipe fairly, the older wave parks ~630 cycles at every barrier and the younger one then runs alone.
  // EVEN phase p: every DMA of ours has landed (vmcnt(0): phases <= p+2), barrier, then this wave (0-3) fetches phase p+3 at
  // once and phase p+4 at mid-phase - into the slots of phases p-2 and p-1, which every wave has left.  ODD phase: nothing but
  // the slot rotation.  The ring holds p, p+1, p+2 landed and p+3, p+4 in flight: 5 slots; flight time 1.5-2 phases (~1.5 us).
  template <bool EVEN>
  __device__ __forceinline__ void phase_begin2() {
#if defined(DN_STAMP) && DN_STAMP == 2
    if (st_top_pending) { st_top += stamp() - st_last; st_top_pending = 0; ++st_n; }
#endif
    if constexpr (EVEN) {
#ifndef DN_EXP_NOBARRIER   // timing experiment only (UNSAFE: no cross-wave ordering of ring slots)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#endif
      advance_issue();
      dma_phase(pend_src, pend_dst, wave < 4 ? 1u : 0u);
      dma_phase(pend_src + PER_WAVE * kPieceBytes, pend_dst + PER_WAVE * kPieceBytes, wave < 4 ? 1u : 0u);
    }
    slot_nxt = (slot_nxt + 1 == kRingPhases) ? 0 : slot_nxt + 1;
    rda_cur = slot_cur_base + lane16;
    slot_cur_base = __builtin_amdgcn_readfirstlane(ring_addr + slot_nxt * kSlotBytes);
  }
  template <bool EVEN>
  __device__ __forceinline__ void mid_phase2() {
    if constexpr (EVEN) {
      advance_issue();
      dma_phase(pend_src, pend_dst, wave < 4 ? 1u : 0u);
      dma_phase(pend_src + PER_WAVE * kPieceBytes, pend_dst + PER_WAVE * kPieceBytes, wave < 4 ? 1u : 0u);
    }
  }
  // ---- the same two-phase period with the fetch SPREAD over the period's first half (explicit-schedule instances, run_stage48x).
  // What a weight DMA costs its wave is the queue in front of the CU's one address path: behind the barrier the four fetching waves
  // used to issue four 1 KiB loads each at once (and four more at mid-phase) - ~80 cycles per load for the issuing wave, measured
  // with waves 0-3 running alone (profiles/r04_headline_schedule.md).  Here ONE wave issues at a time, two loads per piece
  // position: wave w (0-3) at positions 4w .. 4w + 3 of the 32-piece period - pieces 4w .. 4w + 3 of phase p + 3, then of phase
  // p + 4.  The last load leaves at position 15; the barrier that needs it landed (vmcnt(0), even phase p + 2) is 16 pieces later.
  // Scalar state per period: this wave's global source and LDS destination of its four pieces of either phase.
  unsigned long long xs_src[2];
  unsigned xs_dst[2];
  // PERIOD = 32 (one barrier per two phases: the W = 256 instance, see phase_begin2) or 16 (one per phase: phase_begin's ring
  // accounting - two younger phases of four loads per fetching wave stay in flight - with the fetch of phase p + 4 spread over
  // positions 0 .. 7 of phase p, wave w at 2w and 2w + 1)
  template <int PERIOD>
  __device__ __forceinline__ void xs_period_begin() {
    static_assert(PERIOD == kPhasePieces || PERIOD == 2 * kPhasePieces, "barrier period: one or two phases");
    if constexpr (PERIOD == 2 * kPhasePieces) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifndef DN_EXP_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
#pragma unroll
    for (int h = 0; h < PERIOD / kPhasePieces; ++h) {
      advance_issue();   // (pend_src / pend_dst: this wave's 4 KiB of the phase)
      const unsigned long long bits = reinterpret_cast<unsigned long long>(wsrc + pend_src);
      const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
      const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
      xs_src[h] = (static_cast<unsigned long long>(hi) << 32) | lo;
      xs_dst[h] = __builtin_amdgcn_readfirstlane(ring_addr + pend_dst);
    }
  }
  // fetch step Q of the period (0 .. PERIOD / 2 - 1): ONE wave issues two loads (1 KiB each); everyone else skips (branch inside the statement)
  template <int PERIOD, int Q>
  __device__ __forceinline__ void xs_dma_step() {
#ifndef DN_EXP_NODMA
    constexpr int STEPS = PERIOD / 8;          // steps per fetching wave: 4 (two phases x two pairs) or 2
    constexpr int H = (Q % STEPS) / 2;         // which of the period's phases
    constexpr int OFF = (Q % 2) * 2048;        // which pair of this wave's four pieces of that phase
    unsigned keep;
    asm volatile(
        "s_cmp_lg_u32 %[wave], %[who]\n\t"
        "s_cbranch_scc1 .Ldn_xs_skip%=\n\t"
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_nop 1\n\t"
        "global_load_lds_dwordx4 %[voff], %[sbase] offset:%[o0]\n\t"
        "global_load_lds_dwordx4 %[voff], %[sbase] offset:%[o1]\n\t"
        "s_mov_b32 m0, %[keep]\n"
        ".Ldn_xs_skip%=:"
        : [keep] "=&s"(keep)
        : [wave] "s"(wave), [who] "n"(Q / STEPS), [lds] "s"(xs_dst[H]), [voff] "v"(lane16), [sbase] "s"(xs_src[H]), [o0] "n"(OFF), [o1] "n"(OFF + 1024)
        : "memory", "scc");
#endif
  }
  template <int PERIOD, int POS>
  __device__ __forceinline__ void at_position_xs() {
    if constexpr (POS % PERIOD == 0) xs_period_begin<PERIOD>();
    if constexpr (POS % kPhasePieces == 0) {
      slot_nxt = (slot_nxt + 1 == kRingPhases) ? 0 : slot_nxt + 1;
      rda_cur = slot_cur_base + lane16;
      slot_cur_base = __builtin_amdgcn_readfirstlane(ring_addr + slot_nxt * kSlotBytes);
    }
  }
  // the piece position POS has been consumed: its fetch step, if it has one
  template <int PERIOD, int POS>
  __device__ __forceinline__ void xs_after_piece() {
    if constexpr ((POS % PERIOD) < PERIOD / 2) xs_dma_step<PERIOD, POS % PERIOD>();
  }
  // phase boundary / mid-phase hooks at position POS of a stream whose barrier period is PH pieces (16, or 32: see above)
  template <int PH, int POS>
  __device__ __forceinline__ void at_position() {
    static_assert(PH == kPhasePieces || (G::ASM_READS && G::LEADER_DMA), "the two-phase barrier period is the 48-point geometry's");
    if constexpr (POS % kPhasePieces == 0) {
      if constexpr (PH == 2 * kPhasePieces) phase_begin2<(POS % PH) == 0>();
      else phase_begin();
    }
    if constexpr (POS % kPhasePieces == kPhasePieces / 2) {
      if constexpr (PH == 2 * kPhasePieces) mid_phase2<(POS % PH) == kPhasePieces / 2>();
      else mid_phase();
    }
  }

  // Padding pieces (the stream is padded to whole phases): advance the FIFO over N pieces starting at position
  // POS without issuing MFMAs, so the next tile pass starts again at position 0 of a fresh phase.
  template <int POS, int N, int PH = kPhasePieces, bool BIAS = true>
  __device__ __forceinline__ void skip() {
    static_for<N>([&](auto i_c) {
      constexpr int pos = POS + decltype(i_c)::value;
      static_assert(pos % kPhasePieces != 0 || decltype(i_c)::value == 0, "padding never crosses a phase");
      if constexpr (pos % kPhasePieces == kPhasePieces / 2) at_position<PH, pos>();
      // ASM_READS: only the last kPrefetch skipped positions fetch pieces that will be consumed (the first pieces of the next pass);
      // with fewer padding pieces than FIFO entries the stage before has already fetched the rest (run_stage48, PAD)
      if constexpr (!G::ASM_READS || decltype(i_c)::value >= N - kPrefetch) prefetch<pos>();
    });
    settle<BIAS>();
  }

  // the same for the explicit-schedule pass: the fetch steps of the skipped positions still happen
  template <int PERIOD, int POS, int N>
  __device__ __forceinline__ void skip_xs() {
    static_for<N>([&](auto i_c) {
      constexpr int pos = POS + decltype(i_c)::value;
      static_assert(pos % kPhasePieces != 0 || decltype(i_c)::value == 0, "padding never crosses a phase");
      static_assert(pos % PERIOD != 0, "padding never opens a barrier period");
      xs_after_piece<PERIOD, pos>();
      if constexpr (decltype(i_c)::value >= N - kPrefetch) prefetch<pos>();
    });
    settle<true>();
  }

  // after consuming piece POS (position within the 16-piece phase), read piece POS + kPrefetch into its FIFO slot
  template <int POS>
  __device__ __forceinline__ void prefetch() {
    constexpr int q = (POS % kPhasePieces) + kPrefetch;
    if constexpr (G::ASM_READS) {
      const unsigned base = (q < kPhasePieces) ? rda_cur : slot_cur_base + lane16;
#ifdef DN_EXP_NOREAD   // timing experiment only: no A-fragment traffic (the FIFO keeps whatever it held)
      asm volatile("; no read %1" : "+v"(af[POS % kPrefetch]) : "v"(base));
#else
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[POS % kPrefetch]) : "v"(base), "n"((q % kPhasePieces) * kPieceBytes));
#endif
    } else {
      const char* base = (q < kPhasePieces) ? rd_cur : rd_nxt;
#ifndef DN_EXP_NOREAD
      af[POS % kPrefetch] = *reinterpret_cast<const f32x4*>(base + (q % kPhasePieces) * kPieceBytes);
#endif
    }
  }

  // ---- ASM_READS: the explicit reads and their counted waits ----
  // prologue: FIFO entry E (piece E of the first phase) from the slot rda_nxt points at
  template <int E>
  __device__ __forceinline__ void prologue_read() {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[E]) : "v"(slot_cur_base + lane16), "n"(E * kPieceBytes));
  }
  // piece POS is about to be consumed: wait until at most NEWER of our younger reads are outstanding
  template <int POS, int NEWER>
  __device__ __forceinline__ f32x4 take() {
#ifdef DN_EXP_NOWAIT   // timing experiment only (UNSAFE: fragments are consumed before they have arrived)
    asm volatile("; no wait %1" : "+v"(af[POS % kPrefetch]) : "n"(NEWER));
#else
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[POS % kPrefetch]) : "n"(NEWER));
#endif
    return af[POS % kPrefetch];
  }
  // bias rows of the next tile: addr = LDS byte address of this lane group's 16 bytes of a bias tile, OFF = byte offset
  template <int OFF>
  __device__ __forceinline__ void bias_prefetch(unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias_nxt) : "v"(addr), "n"(OFF));
  }
  template <int NEWER>
  __device__ __forceinline__ f32x4 bias_take() {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bias_nxt) : "n"(NEWER));
    return bias_nxt;
  }
  // Land everything in flight.  To the compiler a fragment is an ordinary value from its read-asm on, so wherever control
  // flow merges (the trunk's run-time layer loop, the skip / no-skip branch, the tile loop) it may copy the FIFO registers
  // (phi copies) - before the data has arrived, if a read were still in flight there.  Called at the end of every stage
  // and after the padding pieces: one exposed LDS round trip per stage (12 per 1184 pieces).
  template <bool BIAS = true>   // BIAS = false: a stream without bias rows (the backward chain) - bias_nxt is not a live register
  __device__ __forceinline__ void settle() {
    if constexpr (!G::ASM_READS) return;   // (compiler-issued reads: the compiler places its own waits)
#ifdef DN_EXP_NOSETTLE   // timing experiment only (UNSAFE: phi copies may read fragments in flight)
    return;
#endif
    static_assert(kPrefetch >= 2 && kPrefetch <= 4, "settle() names every FIFO entry");
    if constexpr (!BIAS) {
      static_assert(kPrefetch == 2, "the bias-less form is the 48-point pipeline's");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
    } else
    if constexpr (kPrefetch == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(bias_nxt));
    else if constexpr (kPrefetch == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(bias_nxt));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[kPrefetch - 1]), "+v"(bias_nxt));
  }
};
// the 32-points-per-wave kernels (mlp_fused.hip, mlp_train.hip): compiler-issued reads, every wave fetches, FIFO of kPrefetch = 4
#ifdef DN_PREFETCH_SET_BY_KERNEL_SOURCE   // (a 48-point translation unit: kPrefetch is that geometry's; this alias is not instantiated there)
using PipeGeo32 = PipeGeometry<false, false, false, 4>;
#else
using PipeGeo32 = PipeGeometry<false, false, false, kPrefetch>;
#endif
template <int WAVES>
using Pipe = PipeT<WAVES, PipeGeo32>;

template <int BF16>
__device__ __forceinline__ f32x16 mma_piece(f32x16 acc, f32x4 a_raw, typename Prec<BF16>::BPiece b) {
  if constexpr (BF16 == 1) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, a_raw);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else if constexpr (BF16 == 2) {
    const f16x8 a = __builtin_bit_cast(f16x8, a_raw);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_raw[3], b[3], acc, 0, 0, 0);
    return acc;
  }
}

// One GEMM stage: NT_OUT output tiles, KH hidden pieces + KP positional-encoding pieces per tile, for the PT
// point-tiles this wave owns (each A fragment is read once and feeds PT MFMAs).
// Each tile's K-reduction completes on its own, so only PT 32x32 accumulator tiles are live at a time; `emit`
// consumes them (ReLU + convert into the next stage's B pieces, or pick the output rows) while the next tile's
// MFMAs are already being issued.  POS0 = piece position (mod 16) at which the stage starts; phase boundaries
// (counted vmcnt + barrier + next DMA) are compile-time positions in the unrolled sequence.
template <int BF16, int PT, int NT_OUT, int KH, int KP, int POS0, bool HAS_BIAS = true, class PipeT, class BH, class BP, class Emit>
__device__ __forceinline__ void run_stage(PipeT& pipe, const BH& bh /* [PT][KH] */, BP&& bp /* (t, k) -> PE piece */,
                                          const char* bias_lds /* this lane-half's 64 B of tile 0 */, Emit&& emit) {
  constexpr int KT = KH + KP;
  static_for<NT_OUT>([&](auto nt_c) {
    constexpr int nt = decltype(nt_c)::value;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0, b3 = b0;
    if constexpr (HAS_BIAS) {
      const f32x4* bptr = reinterpret_cast<const f32x4*>(bias_lds + nt * 128);
      b0 = bptr[0]; b1 = bptr[1]; b2 = bptr[2]; b3 = bptr[3];
    }
    f32x16 a[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      a[t][0] = b0[0]; a[t][1] = b0[1]; a[t][2] = b0[2]; a[t][3] = b0[3];
      a[t][4] = b1[0]; a[t][5] = b1[1]; a[t][6] = b1[2]; a[t][7] = b1[3];
      a[t][8] = b2[0]; a[t][9] = b2[1]; a[t][10] = b2[2]; a[t][11] = b2[3];
      a[t][12] = b3[0]; a[t][13] = b3[1]; a[t][14] = b3[2]; a[t][15] = b3[3];
    }
#ifdef DN_EXP_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    static_for<KT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int pos = POS0 + nt * KT + k;
      if constexpr (pos % kPhasePieces == 0) pipe.phase_begin();
      if constexpr (pos % kPhasePieces == kPhasePieces / 2) pipe.mid_phase();
      const f32x4 araw = pipe.af[pos % kPrefetch];
      static_for<PT>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (k < KH) a[t] = mma_piece<BF16>(a[t], araw, bh[t][k]);
        else a[t] = mma_piece<BF16>(a[t], araw, bp(t, k - KH));
      });
      pipe.template prefetch<pos>();
#ifndef DN_EXP_NOPIN
      // pin the interleave: the MFMAs of this piece, then the one LDS read that refills its FIFO slot
      __builtin_amdgcn_sched_group_barrier(0x008, (BF16 ? 1 : 4) * PT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
    });
#ifdef DN_EXP_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // region boundary BEFORE the epilogue: emit(nt)'s VALU work may overlap tile nt+1's MFMAs, but whole tiles
    // are not interleaved (that would keep several accumulator tiles live and spill)
    __builtin_amdgcn_sched_barrier(0);
    static_for<PT>([&](auto t_c) { emit(nt_c, t_c, a[decltype(t_c)::value]); });
  });
}

// accumulator tile -> the next stage's B pieces (in the register-resident chain), optional ReLU.
// bf16: convert first, then ReLU on the packed pairs as a signed-int16 max with 0 (a negative bf16 has its
// sign bit set, i.e. is a negative int16; -0.0 -> +0.0): 4 v_pk_max_i16 instead of 8 v_max_f32 per piece.
template <int BF16, bool RELU, int S>
__device__ __forceinline__ typename Prec<BF16>::BPiece make_piece(const f32x16& acc) {
  using P = Prec<BF16>;
  typename P::BPiece piece;
  if constexpr (BF16) {
#pragma unroll
    for (int e = 0; e < 8; ++e) piece[e] = static_cast<typename P::Elem>(acc[S * 8 + e]);
    if constexpr (RELU) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      s16x8 bits = __builtin_bit_cast(s16x8, piece);
      const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      bits = __builtin_elementwise_max(bits, zero);
      piece = __builtin_bit_cast(typename P::BPiece, bits);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) piece[e] = RELU ? fmaxf(acc[S * 4 + e], 0.0f) : acc[S * 4 + e];
  }
  return piece;
}

template <int BF16, bool RELU, int NT, class BO>
__device__ __forceinline__ void emit_pieces(const f32x16& acc, BO& bo) {
  using P = Prec<BF16>;
  static_for<P::PPT>([&](auto s_c) {
    constexpr int s = decltype(s_c)::value;
    bo[NT * P::PPT + s] = make_piece<BF16, RELU, s>(acc);
  });
}

// 16-byte store to (wave-uniform base) + lane*16 as ONE instruction with an SGPR base and a single lane-offset VGPR:
// no per-store 64-bit VGPR address (the training kernels issue ~150 of these per tile; with flat VGPR addressing
// hipcc spilled pieces in the head stages, and every spill reload waits with vmcnt(0) - draining the HBM stores and
// the weight pipeline).  `base` must be computed from scalar values only.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
  return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <class V>
__device__ __forceinline__ void store16_uniform(const char* base, unsigned lane16, const V& val) {
  static_assert(sizeof(V) == 16, "one dwordx4 per lane");
  const char* b = uniform_ptr(base);  // hipcc may have formed the (uniform) address in VGPRs: an "s" operand needs SGPRs
  const unsigned voff = lane16;
  const f32x4 data = __builtin_bit_cast(f32x4, val);
  // The hazard recognizer does not look inside an asm statement, so the wait states are spelled out:
  //   s_nop 4 before: the base may come straight from v_readfirstlane (VALU writes SGPR -> VMEM reads it: 5 wait states);
  //   s_nop 1 after: a store of more than 8 bytes still reads its data VGPRs when the next instruction issues - a VALU
  //   write to them there corrupts the stored dwords (seen: hipcc reuses the piece registers at once).
  // Non-temporal: the saved activations / gradients are a write-once stream; with the default policy their lines
  // crowd the 1.2 MB weight stream out of L2 (nt: forward 1.36 -> 1.26 ms, backward 1.21 -> 0.92 ms at 786 k points).
#if !defined(DN_STORE_POLICY_ID) || DN_STORE_POLICY_ID == 1
#define DN_STORE_POLICY " nt"
#elif DN_STORE_POLICY_ID == 0   // ablation hooks
#define DN_STORE_POLICY ""
#elif DN_STORE_POLICY_ID == 2
#define DN_STORE_POLICY " sc1"
#else
#define DN_STORE_POLICY " sc0 sc1"
#endif
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %[voff], %[data], %[sbase]" DN_STORE_POLICY "\n\ts_nop 1"
               : : [voff] "v"(voff), [data] "v"(data), [sbase] "s"(b) : "memory");
}

// The same store at (uniform base) + (uniform byte offset) + lane * 16, the offset added to the lane offset INSIDE the statement:
// a kernel with a hundred stores per tile at different offsets of a few base pointers (the 48-point training kernels) otherwise
// has every full 64-bit address formed ahead of time in scalar registers (83 spilled SGPRs), or in VGPRs.
template <class V>
__device__ __forceinline__ void store16_uniform_at(const char* base, unsigned byte_off, unsigned lane16, const V& val) {
  static_assert(sizeof(V) == 16, "one dwordx4 per lane");
  // `base` is the result of uniform_ptr() and `byte_off` scalar arithmetic on kernel arguments: both already live in SGPRs.
  // (Passing them through v_readfirstlane again made hipcc keep VGPR copies of the scalars they are formed from across the
  // whole tile loop - eight spilled registers in the backward kernel.)
  const char* b = base;
  const unsigned off = byte_off;
  const unsigned voff = lane16;
  const f32x4 data = __builtin_bit_cast(f32x4, val);
  unsigned tmp;
  asm volatile("v_add_u32 %[tmp], %[off], %[voff]\n\ts_nop 4\n\tglobal_store_dwordx4 %[tmp], %[data], %[sbase]" DN_STORE_POLICY "\n\ts_nop 1"
               : [tmp] "=&v"(tmp) : [voff] "v"(voff), [off] "s"(off), [data] "v"(data), [sbase] "s"(b) : "memory");
}

// The unit store of the s8-48 layout (mlp_geo48.h): as store16_uniform_at, with the lane's row inside the 1 KiB unit swizzled -
// lane (g = lane / 16, j = lane % 16) writes row g * 16 + (j ^ 8 (g & 1)): the odd lane groups keep their first eight points in
// the upper 128 bytes of their 256, so that the weight-gradient kernel's two 16-lane read groups (an even and an odd lane group,
// eight points each) fall on different halves of the 64 LDS banks (PMC on the unswizzled layout: one conflict cycle per LDS cycle).
template <class V>
__device__ __forceinline__ void store16_unit48(const char* base, unsigned byte_off, unsigned lane16, const V& val) {
  static_assert(sizeof(V) == 16, "one dwordx4 per lane");
  const char* b = base;
  const unsigned off = byte_off;
  const unsigned voff = lane16;
  const f32x4 data = __builtin_bit_cast(f32x4, val);
  unsigned tmp;
  // (wait states between a VALU write of the base SGPRs - an SGPR-spill restore by v_readlane_b32 can sit right in front of this
  // statement - and the store: the four address instructions + s_nop 0 = 5)
  asm volatile("v_and_b32 %[tmp], 0x100, %[voff]\n\tv_lshrrev_b32 %[tmp], 1, %[tmp]\n\tv_xor_b32 %[tmp], %[tmp], %[voff]\n\tv_add_u32 %[tmp], %[off], %[tmp]\n\t"
               "s_nop 0\n\tglobal_store_dwordx4 %[tmp], %[data], %[sbase]" DN_STORE_POLICY "\n\ts_nop 1"
               : [tmp] "=&v"(tmp) : [voff] "v"(voff), [off] "s"(off), [data] "v"(data), [sbase] "s"(b) : "memory");
}

// ---- 8-bit saved tensors (DN_PREC_BF16_S8): a bf16 B piece (8 values per lane) -> 8 bytes --------------------------
// GRAD = false: e4m3 (activations, O(1)), saturated at +-448 (OCP e4m3 has no infinity: an unclamped overflow converts to NaN,
// and one NaN activation poisons a whole layer's weight gradient); GRAD = true: e5m2 of value * scale, saturated at +-57344
// (e5m2 has infinities).  One v_med3_f32 per element either way.
// From the bf16 values the kernel itself used (exact in fp32), so the stored bytes equal convert_s8_kernel's on the bf16 buffers.
constexpr float kE4m3Max = 448.0f, kE5m2Max = 57344.0f;
template <bool GRAD>
__device__ __forceinline__ void piece_to_8bit(const bf16x8& v, float scale, unsigned& w0, unsigned& w1) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f[e] = static_cast<float>(v[e]);
    if constexpr (GRAD) f[e] = __builtin_amdgcn_fmed3f(f[e] * scale, -kE5m2Max, kE5m2Max);
    else f[e] = __builtin_amdgcn_fmed3f(f[e], -kE4m3Max, kE4m3Max);
  }
  int a = 0, b = 0;
  if constexpr (GRAD) {
    a = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], a, false); a = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], a, true);
    b = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], b, false); b = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], b, true);
  } else {
    a = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], a, false); a = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], a, true);
    b = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], b, false); b = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], b, true);
  }
  w0 = static_cast<unsigned>(a); w1 = static_cast<unsigned>(b);
}

// 16 B per lane, per-lane global address -> LDS (M0 base + lane*16), as an opaque instruction: the counted waits of
// the weight pipeline cover it (it is issued >= 3 phases before its data is read) and hipcc never waits on it.
__device__ __forceinline__ void dma16_lanes(const void* src_lane, unsigned lds_addr) {
  const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
  const void* src = src_lane;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\t"
      "s_nop 1\n\t"
      "global_load_lds_dwordx4 %[vaddr], off\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [vaddr] "v"(src)
      : "memory");
}

// Two 16-bit (bf16) pairs -> four 8-bit values of x / scale (e4m3, or e5m2 with BF8), the two converts as ONE asm statement with an
// early-clobber output.  Through the builtin the destination is read-modify-write (each convert keeps the other half), so the
// compiler materialises an initial value for every dword - a v_mov per four stored bytes, ~500 per tile pass of the training kernels.
template <bool BF8>
__device__ __forceinline__ unsigned cvt_pairs_8bit(unsigned d0, unsigned d1, float scale) {
  unsigned r;
  if constexpr (BF8)
    asm("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %3\n\tv_cvt_scalef32_pk_bf8_bf16 %0, %2, %3 op_sel:[0,0,1]" : "=&v"(r) : "v"(d0), "v"(d1), "v"(scale));
  else
    asm("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %3\n\tv_cvt_scalef32_pk_fp8_bf16 %0, %2, %3 op_sel:[0,0,1]" : "=&v"(r) : "v"(d0), "v"(d1), "v"(scale));
  return r;
}

// 4 B per lane for lanes 0-47 (a 48-point wave tile's input row): the lane mask is set and restored inside the statement, so the
// caller stays straight-line code (every lane is active where this is used)
__device__ __forceinline__ void dma4_lanes48(const void* src_lane, unsigned lds_addr) {
  const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
  const void* src = src_lane;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[lds]\n\t"
      "s_mov_b32 exec_hi, 0xffff\n\t"
      "s_nop 1\n\t"
      "global_load_lds_dword %[vaddr], off\n\t"
      "s_mov_b32 exec_hi, -1\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [lds] "s"(lds), [vaddr] "v"(src)
      : "memory");
}

// ---- positional encoding straight into B-piece layout -------------------------------------------------
// Slot u of this lane-half (mlp_layout.h pe_slot_col): u < 6*(L/2): sin/cos of this half's frequencies;
// then identity (half 0: x, y; half 1: z); rest zero padding.
template <int BF16, int L, int NPIECES, class BP>
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
      piece[e] = static_cast<typename P::Elem>(v);
    });
    bp[p] = piece;
  });
}

// Encoded-rows input (FlexibleNeRFModel.forward(x) call surface): gather this lane's slots from x.
template <int BF16, int L, int NPIECES, class BP>
__device__ __forceinline__ void gather_pieces(const float* row, int h, BP& bp) {
  using P = Prec<BF16>;
  static_for<NPIECES>([&](auto p_c) {
    constexpr int p = decltype(p_c)::value;
    typename P::BPiece piece;
#pragma unroll
    for (int e = 0; e < P::EPP; ++e) {
      const int col = pe_slot_col(L, h, p * P::EPP + e);
      const float v = (col >= 0) ? row[col] : 0.0f;
      piece[e] = static_cast<typename P::Elem>(v);
    }
    bp[p] = piece;
  });
}

}  // namespace dn
