// bf16 / fp16 inference forward in the 48-points-per-wave geometry (see mlp_geo48.h): fused positional encoding +
// FlexibleNeRFModel forward (reference run_network, nerf/train_utils.py:72-89; positional_encoding
// nerf/nerf_helpers.py:115-159; FlexibleNeRFModel.forward nerf/models.py:233-256), same design as mlp_fused.hip -
// persistent workgroups, register-resident activation chain, LDS-DMA weight ring (Pipe<8>) - on
// v_mfma_f32_16x16x32_bf16: lane l of a wave holds point (l & 15) of each of its three 16-point groups and lane group
// g = l >> 4 holds rows 4g..4g+3 of every 16-row accumulator tile, which is k-block g of the next layer's B operand.
// Every A fragment read from LDS feeds three MFMAs: 384 points per pass of the weight stream instead of 256.
// A-fragment FIFO of 2 pieces here (4 in the 32-point kernels): a piece lasts three MFMAs = 48 cycles, and the 256-VGPR
// budget of two waves per SIMD is spent on the two 96-register activation sets
#include <vector>

#include "mlp_fused48_kernel.h"

namespace dn {

// the explicit-schedule instances of the paper network are compiled in mlp_fused48_paper_{bf16,fp16}.hip, the W = 128 instances in
// mlp_fused48_w128.hip (same lists there): the instance list builds as four translation units in parallel
extern template __global__ void mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 0, 2>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<256, 1, 8, 0x10u, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<256, 2, 8, 0x10u, 1, 0, 2>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<256, 2, 8, 0x10u, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 2>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 0, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 0, 1>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 2>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 3>(FwdParams, G48Params);
extern template __global__ void mlp_forward48_kernel<128, 1, 0, 0u, 0, 2>(FwdParams, G48Params);

// ---- pack: nn.Linear tensors -> bias rows + encoding tables + 16x32 A pieces -----------------------------------
template <int F>
__device__ __forceinline__ void pack48_body(const NetLayout& L, const PackPtrs& ptrs, const G48Tables& tabs, char* __restrict__ region);

template <int F>
__global__ void pack48_kernel(NetLayout L, PackPtrs ptrs, G48Tables tabs, char* __restrict__ region) {
  pack48_body<F>(L, ptrs, tabs, region);
}

// two networks of one architecture (the coarse and the fine net of a training step) in one launch: blockIdx.y picks the net
template <int F>
__global__ void pack48_pair_kernel(NetLayout L, PackPtrs ptrs_a, PackPtrs ptrs_b, G48Tables tabs, char* __restrict__ region_a,
                                   char* __restrict__ region_b) {
  if (blockIdx.y == 0) pack48_body<F>(L, ptrs_a, tabs, region_a);
  else pack48_body<F>(L, ptrs_b, tabs, region_b);
}

template <int F>
__device__ __forceinline__ void pack48_body(const NetLayout& L, const PackPtrs& ptrs, const G48Tables& tabs, char* __restrict__ region) {
  using Elem = typename Prec<F>::Elem;
  const int n_rows = L.total_bias_tiles * 16;
  float* bias_out = reinterpret_cast<float*>(region);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < L.bias_bytes / 4; idx += gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (idx < n_rows) {
      const int tile = idx / 16, r = idx % 16;
      int s = 0;
      while (s + 1 < L.n_stages && L.st[s + 1].bias0 <= tile) ++s;
      const StageDesc& st = L.st[s];
      const int ts = tile - st.bias0;
      if (st.src2 >= 0) {
        if (ts == 0) v = (r == 0) ? ptrs.b[st.src2][0] : 0.0f;
        else { const int n = (ts - 1) * 16 + r; v = (n < st.n_real) ? ptrs.b[st.src][n] : 0.0f; }
      } else {
        const int n = ts * 16 + r;
        v = (n < st.n_real) ? ptrs.b[st.src][n] : 0.0f;
      }
    }
    bias_out[idx] = v;
  }
  // encoding tables: [4][16] xyz entries of 16 B, then (at byte 1024) [4][8] dir entries: (frequency, phase, identity, sine)
  f32x4* tab = reinterpret_cast<f32x4*>(region + L.bias_bytes);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < kG48TableBytes / 16; idx += gridDim.x * blockDim.x) {
    const int kind_pe = idx < 64 ? 1 : 2;
    const int gg = idx < 64 ? idx / 16 : (idx - 64) / 8, u = idx < 64 ? idx % 16 : (idx - 64) % 8;
    const int c = g48_pe_col(kind_pe, gg, u, kind_pe == 1 ? L.LX : L.LD);
    f32x4 entry = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c >= 0 && c < 3) entry[2] = 1.0f;
    else if (c >= 3) {
      const int qq = c - 3, f = qq / 6, r = qq % 6;
      entry[0] = (kind_pe == 1 ? tabs.fx[f] : tabs.fd[f]) * 0.15915494309189535f;   // revolutions per unit (pe_value)
      entry[1] = r < 3 ? 0.0f : 0.25f;
      entry[3] = 1.0f;
    }
    tab[idx] = entry;
  }
  // pieces
  Elem* wout = reinterpret_cast<Elem*>(region + L.bias_bytes + kG48TableBytes);
  const long long n_elems = static_cast<long long>(L.total_pieces) * 64 * 8;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < n_elems;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int e = static_cast<int>(idx % 8);
    const int lane = static_cast<int>((idx / 8) % 64);
    const int piece = static_cast<int>(idx / 512);
    const int i = lane & 15, gg = lane >> 4;
    float v = 0.0f;
    int s = 0;
    while (s + 1 < L.n_stages && L.st[s + 1].piece0 <= piece) ++s;
    const StageDesc& st = L.st[s];
    const int rel = piece - st.piece0;
    if (rel < st.n_tiles * st.pieces_per_tile) {
      const int ts = rel / st.pieces_per_tile;
      const int k = rel % st.pieces_per_tile;
      const int kh = st.hidden_in / 32;
      int col;
      if (k < kh) {
        col = st.col_hidden0 + g48_hidden_col(k, gg, e);
      } else {
        const int pc = g48_pe_col(st.pe_kind, gg, (k - kh) * 8 + e, st.pe_kind == 1 ? L.LX : L.LD);
        col = pc >= 0 ? st.col_pe0 + pc : -1;
      }
      if (col >= 0) {
        if (st.src2 >= 0) {
          if (ts == 0) v = (i == 0) ? ptrs.w[st.src2][col] : 0.0f;
          else { const int n = (ts - 1) * 16 + i; v = (n < st.n_real) ? ptrs.w[st.src][static_cast<long long>(n) * st.ld + col] : 0.0f; }
        } else {
          const int n = ts * 16 + i;
          v = (n < st.n_real) ? ptrs.w[st.src][static_cast<long long>(n) * st.ld + col] : 0.0f;
        }
      }
    }
    wout[idx] = static_cast<Elem>(v);
  }
}

void fill_freqs(float* f, int num_fns, int log_sampling);  // rays_sampling.hip

int launch_pack48(const dn_mlp_desc& d, int precision, const PackPtrs& ptrs, char* region, hipStream_t stream) {
  NetLayout L;
  build_layout48(d, &L);
  G48Tables tabs{};
  fill_freqs(tabs.fx, d.num_encoding_fn_xyz, d.log_sampling_xyz);
  if (d.use_viewdirs) fill_freqs(tabs.fd, d.num_encoding_fn_dir, d.log_sampling_dir);
  tabs.LX = d.num_encoding_fn_xyz; tabs.LD = d.num_encoding_fn_dir;
  if (precision == DN_PREC_F16) hipLaunchKernelGGL(pack48_kernel<2>, dim3(pack48_blocks(L)), dim3(256), 0, stream, L, ptrs, tabs, region);
  else hipLaunchKernelGGL(pack48_kernel<1>, dim3(pack48_blocks(L)), dim3(256), 0, stream, L, ptrs, tabs, region);
  return check_launch("mlp_pack48");
}

// 1 when an fp16 launch of this network reports EVERY hidden activation that leaves fp16's range (FwdParams::range_flag): the
// fixed-shape instances and the W = 128 ones; the run-time-shape W = 256 instance tracks only the head stages
bool g48_range_guard_complete(const dn_mlp_desc& d) {
  if (!g48_supported(d, DN_PREC_F16)) return false;
  NetLayout L;
  build_layout48(d, &L);
  const bool paper = d.hidden_size == 256 && d.num_layers == 8 && L.skip_mask == 0x10u && d.use_viewdirs;
  return paper || d.hidden_size == 128;
}

int launch_pack48_pair(const dn_mlp_desc& d, const PackPtrs& a, const PackPtrs& b, char* region_a, char* region_b, hipStream_t stream) {
  NetLayout L;
  build_layout48(d, &L);
  G48Tables tabs{};
  fill_freqs(tabs.fx, d.num_encoding_fn_xyz, d.log_sampling_xyz);
  if (d.use_viewdirs) fill_freqs(tabs.fd, d.num_encoding_fn_dir, d.log_sampling_dir);
  tabs.LX = d.num_encoding_fn_xyz; tabs.LD = d.num_encoding_fn_dir;
  static_assert(sizeof(NetLayout) + 2 * sizeof(PackPtrs) + sizeof(G48Tables) + 16 <= 4096, "kernel arguments of the pair pack");
  hipLaunchKernelGGL(pack48_pair_kernel<1>, dim3(pack48_blocks(L), 2), dim3(256), 0, stream, L, a, b, tabs, region_a, region_b);
  return check_launch("mlp_pack48_pair");
}

// In-kernel compositing is OFF unless DEXNERF_FUSED_COMPOSITE=1 (read per call).  It is bit-identical to the two-kernel path and
// takes the raw radiance field out of HBM (fine launch of the headline configuration: see HISTORY.md section 4.7f for the PMC bytes),
// but it costs time: a ray is composited by ONE wave with its SIMD to itself - exponentials, divisions and an fp64 scan as one
// dependency chain - where the standalone kernel hides that latency behind eight waves per SIMD; measured +3 % on a D8/W256
// render and +26 % on the as-shipped 4 x 128 nets.  The network kernels are not byte-bound, so the bytes saved buy nothing back.
static bool fused_composite_enabled() {
  const char* e = std::getenv("DEXNERF_FUSED_COMPOSITE");
  return e != nullptr && std::atoi(e) == 1;
}

int launch_forward48(const dn_mlp_desc& d, int precision, const FwdParams& p_in, const char* region, hipStream_t stream,
                     const CompParams* comp, int* composited) {
  if (composited) *composited = 0;
  NetLayout L;
  build_layout48(d, &L);
  FwdParams p = p_in;
  G48Params q{};
  q.base = region; q.bias_bytes = L.bias_bytes; q.total_pieces = L.total_pieces;
  p.n_tiles = (p.n_points + kG48PointsPerWg - 1) / kG48PointsPerWg;
  size_t lds = g48_lds_bytes(L);
  if (lds > 160 * 1024) { set_error("mlp_forward48: %zu bytes of LDS", lds); return DN_E_UNSUPPORTED; }
  const int cus = device_cus();
  long long grid = p.n_tiles < cus ? p.n_tiles : cus;
  auto launch = [&](auto kern) -> int {
    if (int rc = ensure_big_lds(reinterpret_cast<const void*>(kern))) return rc;
#ifdef DN_STAMP   // diagnostic build: synchronous, allocates, prints - never part of the shipped library
    static unsigned* dbg = nullptr;
    const size_t words = static_cast<size_t>(grid) * kG48Waves * 16;
    if (!dbg) (void)hipMalloc(&dbg, 256 * kG48Waves * 16 * sizeof(unsigned));
    (void)hipMemsetAsync(dbg, 0, words * sizeof(unsigned), stream);
    q.dbg = dbg;
#endif
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(kG48Waves * 64), lds, stream, p, q);
#ifdef DN_STAMP
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned> h(words);
    (void)hipMemcpy(h.data(), dbg, words * sizeof(unsigned), hipMemcpyDeviceToHost);
    double acc[2][16] = {};
    for (size_t w = 0; w < words / 16; ++w)
      for (int i = 0; i < 16; ++i) acc[(w % kG48Waves) >= 4][i] += h[w * 16 + i];
#if DN_STAMP == 2
    for (int g = 0; g < 2; ++g)
      fprintf(stderr, "[stamp] waves %d-%d: per pass: top-of-tile %.0f cycles, rest of the pass %.0f (%.1f passes per wave)\n", g * 4, g * 4 + 3,
              acc[g][8] / acc[g][4], acc[g][9] / acc[g][4], acc[g][4] / (words / 32.0));
#endif
#if DN_STAMP == 4
    for (int g = 0; g < 2; ++g)
      fprintf(stderr, "[stamp] waves %d-%d: per pass: top %.0f, layer1 (32 pieces, 16 tiles) %.0f, trunk (928 pieces, 112 tiles) %.0f, heads (212 pieces + 12 pad, 26 tiles) %.0f cycles\n",
              g * 4, g * 4 + 3, acc[g][8] / acc[g][4], acc[g][10] / acc[g][4], acc[g][11] / acc[g][4], acc[g][9] / acc[g][4]);
#endif
#if DN_STAMP == 3
    for (int g = 0; g < 2; ++g)
      fprintf(stderr, "[stamp] waves %d-%d: per phase: release -> next arrival %.1f, barrier (arrival -> release, incl. the vmcnt wait) %.1f cycles; per pass: top-of-tile %.0f\n",
              g * 4, g * 4 + 3, acc[g][3] / acc[g][4], acc[g][1] / acc[g][4], acc[g][8] / (acc[g][4] / 74.0));
#endif
    for (int g = 0; g < 2 && DN_STAMP == 1; ++g) {
      const double n = acc[g][4] > 0 ? acc[g][4] : 1;
      fprintf(stderr, "[stamp] waves %d-%d: per phase: vmcnt wait %.1f, barrier %.1f, DMA issue %.1f, quarters %.1f %.1f %.1f %.1f cycles (%.0f phases per wave); per PASS: tail %.0f, top-of-tile %.0f\n",
              g * 4, g * 4 + 3, acc[g][0] / n, acc[g][1] / n, acc[g][2] / n, acc[g][5] / n, acc[g][6] / n, acc[g][7] / n, acc[g][3] / n,
              n / (words / 32.0), acc[g][9] / (n / 74.0), acc[g][8] / (n / 74.0));
    }
#endif
    return check_launch("mlp_forward48");
  };
  // fixed-shape instances: the paper network (D8 / W256 / skip 4, view directions - BASELINE configs 2, 4, 5) and the fork's
  // as-shipped 4 x 128 nets (config 3); everything else runs the run-time-shape kernel
  const bool paper = d.hidden_size == 256 && d.num_layers == 8 && L.skip_mask == 0x10u && d.use_viewdirs;
  const bool shipped = d.hidden_size == 128 && d.num_layers == 4 && L.skip_mask == 0u && d.use_viewdirs;
  const bool fixed_ok = std::getenv("DEXNERF_G48_RUNTIME_SHAPE") == nullptr;
#ifdef DN_EXP_ONLY_PAPER   // experiment builds: only the headline instances are compiled (minutes -> seconds per build)
  if (paper && p.act == nullptr && p.mode == 0 && std::getenv("DEXNERF_G48_NO_OVERLAP") == nullptr)
    return precision == DN_PREC_F16 ? launch(mlp_forward48_kernel<256, 2, 8, 0x10u, 1, 0, 2>) : launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 0, 2>);
  if (paper && p.act == nullptr && precision == DN_PREC_F16) return launch(mlp_forward48_kernel<256, 2, 8, 0x10u, 1>);
  if (paper && p.act == nullptr) return launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1>);
  set_error("mlp_forward48: experiment build (DN_EXP_ONLY_PAPER)");
  return DN_E_UNSUPPORTED;
#else
  if (p.act != nullptr) {   // training forward (DN_PREC_BF16_S8): saved units + mask words
    if (precision != DN_PREC_BF16 || !p.save8) { set_error("mlp_forward48(train): the 48-point training forward is the bf16 / 8-bit-saved-tensor mode"); return DN_E_UNSUPPORTED; }
    if (g48_two_group_shape(d) && g48_train_groups(p.n_points, cus) == 2) {   // small launch: 256-point tiles (mlp_geo48.h; the backward asks the same question)
      p.n_tiles = (p.n_points + 255) / 256;
      grid = p.n_tiles < cus ? p.n_tiles : cus;
      return paper ? launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 3>) : launch(mlp_forward48_kernel<128, 1, 4, 0u, 1, 3>);
    }
    if (paper && fixed_ok) return launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 2>);
    if (shipped && fixed_ok) return launch(mlp_forward48_kernel<128, 1, 4, 0u, 1, 2>);
    return d.hidden_size == 256 ? launch(mlp_forward48_kernel<256, 1, 0, 0u, 0, 2>) : launch(mlp_forward48_kernel<128, 1, 0, 0u, 0, 2>);
  }
  // a render that asked for its rays to be composited by the launch itself (dn_render_rays): the fixed-shape instances, whole rays
  // per 384-point tile
  if (comp != nullptr && comp->rgb != nullptr && (paper || shipped) && fixed_ok && p.mode == 0 && p.act == nullptr && p.S >= 1 &&
      kG48PointsPerWg % p.S == 0 && comp->n_rays * p.S == p.n_points && fused_composite_enabled()) {
    q.comp = *comp;
    if (composited) *composited = 1;
    if (precision == DN_PREC_F16) return paper ? launch(mlp_forward48_kernel<256, 2, 8, 0x10u, 1, 0, 0, 1>) : launch(mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 0, 1>);
    return paper ? launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 0, 0, 1>) : launch(mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 0, 1>);
  }
  // the as-shipped nets on rays + depths (the render path): the instance that encodes tile t + 1 inside tile t (a third set of
  // view-direction rows in LDS)
  if (shipped && fixed_ok && p.mode == 0 && std::getenv("DEXNERF_G48_NO_OVERLAP") == nullptr) {
    lds += static_cast<size_t>(kG48Waves) * 3 * kG48PointsPerWave * sizeof(float);
    if (precision == DN_PREC_F16) return launch(mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 1>);
    return launch(mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 1>);
  }
  // the paper network on rays + depths (the render path): tile t + 1's xyz encoding inside tile t's view-direction stage
  if (paper && fixed_ok && p.mode == 0 && std::getenv("DEXNERF_G48_NO_OVERLAP") == nullptr) {
    if (precision == DN_PREC_F16) return launch(mlp_forward48_kernel<256, 2, 8, 0x10u, 1, 0, 2>);
    return launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1, 0, 2>);
  }
  if (precision == DN_PREC_F16) {
    if (paper && fixed_ok) return launch(mlp_forward48_kernel<256, 2, 8, 0x10u, 1>);
    if (shipped && fixed_ok) return launch(mlp_forward48_kernel<128, 2, 4, 0u, 1>);
    return d.hidden_size == 256 ? launch(mlp_forward48_kernel<256, 2>) : launch(mlp_forward48_kernel<128, 2>);
  }
  if (paper && fixed_ok) return launch(mlp_forward48_kernel<256, 1, 8, 0x10u, 1>);
  if (shipped && fixed_ok) return launch(mlp_forward48_kernel<128, 1, 4, 0u, 1>);
  return d.hidden_size == 256 ? launch(mlp_forward48_kernel<256, 1>) : launch(mlp_forward48_kernel<128, 1>);
#endif
}

}  // namespace dn

