// Developer microbenchmark: what the chip sustains on v_mfma_f32_16x16x32_{bf16,f16} with RANDOM operands when nothing else runs -
// the clock-limited ceiling the fused network kernel is measured against (MI355X_MICROARCH.md "DVFS give-back": the 2.5 PFLOP/s
// datasheet figure is 2.4 GHz x full issue; under load the chip holds less).
//   mode 0: operands in registers only (2 A fragments, 24 B pieces, 3 accumulators per wave: the 48-point kernel's operand shape)
//   mode 1: + every A fragment re-read from LDS (ds_read_b128, FIFO of 2, counted waits) as in the kernel
//   mode 2: mode 1 + the weight stream arriving by LDS-DMA into a 5 x 16 KiB ring (one barrier per two phases), random weights
// Each mode: 8 waves per workgroup (two per SIMD), one workgroup per CU; prints wall TFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime).   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/micro/mfma_ceiling.hip -o mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int F> struct P;
template <> struct P<1> { using B = bf16x8; using E = __bf16; };
template <> struct P<2> { using B = f16x8; using E = _Float16; };

template <int F>
__device__ __forceinline__ void mfma(f32x4& d, const f32x4& a, const typename P<F>::B& b) {
  if constexpr (F == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

constexpr int kPiece = 1024, kPhase = 16, kSlots = 5;

template <int F, int MODE>
__global__ __launch_bounds__(512, 2) void ceiling_kernel(const char* wstream, unsigned stream_bytes, int iters, float* out, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using B = typename P<F>::B;
  using E = typename P<F>::E;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // random-looking operands: B pieces (the "activations") from a hash of (lane, k), ~half of them zero like ReLU outputs
  B b[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned h = (lane * 2654435761u) ^ ((t * 8 + k) * 40503u + e * 9176u + blockIdx.x * 7919u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const float v = ((h & 0xffff) / 65536.0f - 0.5f) * 2.0f;
        b[t][k][e] = static_cast<E>(v > 0.0f ? v : 0.0f);
      }
  // LDS image: the whole 80 KiB ring filled with the first bytes of the (random) weight stream
  for (unsigned i = threadIdx.x; i < kSlots * kPhase * kPiece / 16; i += 512)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(wstream)[i];
  __syncthreads();
  const unsigned ring = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const unsigned lane16 = lane * 16;
  f32x4 af[2];
  af[0] = reinterpret_cast<const f32x4*>(smem)[lane];
  af[1] = reinterpret_cast<const f32x4*>(smem + kPiece)[lane];
  f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned src_off = kSlots * kPhase * kPiece;    // next phase to fetch (mode 2)
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)); }
  for (int it = 0; it < iters; ++it) {
    // one "period" = 2 phases = 32 pieces = 96 MFMAs per wave
#pragma unroll
    for (int pos = 0; pos < 32; ++pos) {
      const unsigned slot = ((it * 2 + pos / 16) % kSlots) * (kPhase * kPiece);
      if (MODE == 2 && pos == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(af[pos & 1]));
#pragma unroll
      for (int t = 0; t < 3; ++t) mfma<F>(acc[t], af[pos & 1], b[t][pos & 7]);
      if (MODE == 2 && pos < 16) {
        // wave pos / 4 fetches two pieces of the two phases that were consumed last period (slots are free behind the barrier)
        const unsigned h = (pos % 4) / 2;
        const unsigned dst_slot = ((it * 2 + 3 + h) % kSlots) * (kPhase * kPiece);
        const unsigned long long sbase = reinterpret_cast<unsigned long long>(wstream) + src_off + h * (kPhase * kPiece) + wave * 4096u;
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sbase)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sbase >> 32));
        const unsigned long long sb = (static_cast<unsigned long long>(hi) << 32) | lo;
        const unsigned lds = __builtin_amdgcn_readfirstlane(ring + dst_slot + wave * 4096u);
        unsigned keep;
        if ((pos % 2) == 0)
          asm volatile("s_cmp_lg_u32 %[wave], %[who]\n\ts_cbranch_scc1 .Lskip%=\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 1\n\t"
                       "global_load_lds_dwordx4 %[voff], %[sbase] offset:0\n\tglobal_load_lds_dwordx4 %[voff], %[sbase] offset:1024\n\ts_mov_b32 m0, %[keep]\n.Lskip%=:"
                       : [keep] "=&s"(keep) : [wave] "s"(wave), [who] "s"(pos / 4), [lds] "s"(lds), [voff] "v"(lane16), [sbase] "s"(sb) : "memory", "scc");
        else
          asm volatile("s_cmp_lg_u32 %[wave], %[who]\n\ts_cbranch_scc1 .Lskip%=\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 1\n\t"
                       "global_load_lds_dwordx4 %[voff], %[sbase] offset:2048\n\tglobal_load_lds_dwordx4 %[voff], %[sbase] offset:3072\n\ts_mov_b32 m0, %[keep]\n.Lskip%=:"
                       : [keep] "=&s"(keep) : [wave] "s"(wave), [who] "s"(pos / 4), [lds] "s"(lds), [voff] "v"(lane16), [sbase] "s"(sb) : "memory", "scc");
      }
      if (MODE >= 1) {
        const unsigned nslot = ((it * 2 + (pos + 2) / 16) % kSlots) * (kPhase * kPiece);
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[pos & 1]) : "v"(ring + nslot + ((pos + 2) % 16) * kPiece + lane16));
      }
      (void)slot;
    }
    if (MODE == 2) { src_off += 2 * kPhase * kPiece; if (src_off + 2 * kPhase * kPiece > stream_bytes) src_off = 0; }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
  if (threadIdx.x == 0) {
    unsigned long long t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    clk[blockIdx.x * 2] = t1 - t0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 3; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = s;   // keep the chain alive
}

// mode 3: what a 3-way bf16 split of an fp32 product would issue (x = hi + mid + lo, 8 mantissa bits each; the six partial products
// hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid reach fp32's 24 bits): per k-step THREE weight fragments from LDS and SIX MFMAs into one
// 16-point accumulator group - three activation sets of 32 registers per 16 points, so a wave holds ONE group where the bf16 kernel holds
// three - and per output tile the split of the four accumulators into hi / mid / lo pieces (~22 vector instructions).  Upper bound for
// an "fp32x" mode: no weight DMA (the 3x larger stream would arrive at ~2x today's byte rate), no encodings, no heads.
__global__ __launch_bounds__(512, 2) void split3_kernel(const char* wstream, int iters, float* out, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  bf16x8 b[3][8];   // hi / mid / lo pieces of one 16-point group, 256 features
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned h = (lane * 2654435761u) ^ ((t * 8 + k) * 40503u + e * 9176u + blockIdx.x * 7919u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const float v = ((h & 0xffff) / 65536.0f - 0.5f) * 2.0f;
        b[t][k][e] = static_cast<__bf16>(t == 0 ? (v > 0.0f ? v : 0.0f) : v * (t == 1 ? 0.004f : 0.00002f));
      }
  for (unsigned i = threadIdx.x; i < kSlots * kPhase * kPiece / 16; i += 512)
    reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(wstream)[i];
  __syncthreads();
  const unsigned ring = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const unsigned lane16 = lane * 16;
  f32x4 af[2][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) af[s][c] = reinterpret_cast<const f32x4*>(smem + (s * 3 + c) * kPiece)[lane];
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)); }
  f32x4 acc = {0, 0, 0, 0};
  float keep = 0.0f;
  unsigned rd = 6;   // next piece triple to read (pieces, modulo the ring)
  for (int it8 = 0; it8 < iters / 8; ++it8) {
    // eight output tiles per trip (tile j writes piece j of the three sets: compile-time register indices)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // one output tile: 8 k-steps x 6 MFMAs, then the split of its four values
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(af[k & 1][0]), "+v"(af[k & 1][1]), "+v"(af[k & 1][2]));
        mfma<1>(acc, af[k & 1][0], b[0][k]);   // hi hi
        mfma<1>(acc, af[k & 1][0], b[1][k]);   // hi mid
        mfma<1>(acc, af[k & 1][1], b[0][k]);   // mid hi
        mfma<1>(acc, af[k & 1][0], b[2][k]);   // hi lo
        mfma<1>(acc, af[k & 1][2], b[0][k]);   // lo hi
        mfma<1>(acc, af[k & 1][1], b[1][k]);   // mid mid
        const unsigned base = ring + rd * kPiece + lane16;
        rd = (rd + 3 >= kSlots * kPhase - 3) ? 0 : rd + 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[k & 1][c]) : "v"(base), "n"(0));
      }
      // split: hi = bf16(x), r = x - hi, mid = bf16(r), lo = bf16(r - mid)  (ReLU first), into piece j of the three sets
      asm volatile("s_nop 7" ::: "memory");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = acc[e] > 0.0f ? acc[e] : 0.0f;
        const __bf16 hi = static_cast<__bf16>(x);
        const float r = x - static_cast<float>(hi);
        const __bf16 mid = static_cast<__bf16>(r);
        const __bf16 lo = static_cast<__bf16>(r - static_cast<float>(mid));
        b[0][j][(j & 1) * 4 + e] = hi; b[1][j][(j & 1) * 4 + e] = mid; b[2][j][(j & 1) * 4 + e] = lo;
      }
      keep += acc[0];
      acc = f32x4{0, 0, 0, 0};
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]));
  if (threadIdx.x == 0) {
    unsigned long long t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    clk[blockIdx.x * 2] = t1 - t0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
  if (keep == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = keep;
}

static void run_split3(const char* w, float* out, unsigned long long* clk, int cus) {
  const int iters = 60000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(split3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kPhase * kPiece);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(split3_kernel, dim3(cus), dim3(512), kSlots * kPhase * kPiece, 0, w, iters, out, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep >= 3 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(cus * 2);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double ghz = 0; for (int i = 0; i < cus; ++i) ghz += double(h[2 * i]) / double(h[2 * i + 1]) * 0.1; ghz /= cus;
  const double mfma_flop = double(cus) * 8 * iters * 48.0 * (16.0 * 16 * 32 * 2);
  printf("%-34s %8.2f ms  %7.1f TFLOP/s of bf16 MFMAs = %6.1f TFLOP/s of fp32-grade products (/6) = %.2f x the 157.3 TFLOP/s fp32-MFMA peak   in-kernel clock %.3f GHz\n",
         "bf16 x 3 split (16 points / wave)", best, mfma_flop / best / 1e9, mfma_flop / best / 1e9 / 6.0, mfma_flop / best / 1e9 / 6.0 / 157.3, ghz);
}

template <int F, int MODE>
static void run(const char* name, const char* w, unsigned bytes, float* out, unsigned long long* clk, int cus) {
  const int iters = 6000;
  auto k = ceiling_kernel<F, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * kPhase * kPiece);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {      // the clock settles over the first repetitions: the LAST ones are what a long run sees
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), kSlots * kPhase * kPiece, 0, w, bytes, iters, out, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (rep >= 3 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(cus * 2);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double ghz = 0; for (int i = 0; i < cus; ++i) ghz += double(h[2 * i]) / double(h[2 * i + 1]) * 0.1; ghz /= cus;
  const double flop = double(cus) * 8 * iters * 96.0 * (16.0 * 16 * 32 * 2);
  const double cyc = double(h[0]);
  printf("%-34s %8.2f ms  %7.1f TFLOP/s  = %.3f of 2.5 PF   in-kernel clock %.3f GHz   matrix pipe busy %.3f\n", name, best, flop / best / 1e9,
         flop / best / 1e9 / 2500.0, ghz, double(iters) * 96 * 2 * 16 / cyc);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const unsigned bytes = 1184 * 1024;
  std::vector<unsigned short> hw(bytes / 2);
  unsigned s = 12345u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) / 16777216.0f - 0.5f) * 0.25f; unsigned u; std::memcpy(&u, &f, 4); v = static_cast<unsigned short>(u >> 16); }
  char* w; float* out; unsigned long long* clk;
  hipMalloc(&w, bytes); hipMalloc(&out, cus * 512 * 4); hipMalloc(&clk, cus * 16);
  hipMemcpy(w, hw.data(), bytes, hipMemcpyHostToDevice);
  // fp16 image of the same values
  std::vector<unsigned short> hh(bytes / 2);
  for (size_t i = 0; i < hh.size(); ++i) { unsigned u = static_cast<unsigned>(hw[i]) << 16; float f; std::memcpy(&f, &u, 4); _Float16 q = static_cast<_Float16>(f); std::memcpy(&hh[i], &q, 2); }
  char* w16; hipMalloc(&w16, bytes); hipMemcpy(w16, hh.data(), bytes, hipMemcpyHostToDevice);
  printf("%d CUs, 8 waves per CU, 96 MFMAs per wave per period, random operands\n", cus);
  run<1, 0>("bf16 registers only", w, bytes, out, clk, cus);
  run<1, 1>("bf16 + A fragments from LDS", w, bytes, out, clk, cus);
  run<1, 2>("bf16 + LDS + weight DMA ring", w, bytes, out, clk, cus);
  run<2, 0>("fp16 registers only", w16, bytes, out, clk, cus);
  run<2, 1>("fp16 + A fragments from LDS", w16, bytes, out, clk, cus);
  run<2, 2>("fp16 + LDS + weight DMA ring", w16, bytes, out, clk, cus);
  run_split3(w, out, clk, cus);
  return 0;
}
