// The 48-points-per-wave bf16 inference geometry (mlp_fused48.hip): v_mfma_f32_16x16x32_bf16, 16-row output tiles,
// 32-deep A pieces, every A fragment feeding three MFMAs (three groups of 16 points).  Same network, same 1 KiB piece
// stream discipline as mlp_layout.h, different tile shape - so it has its own stream, stored behind the 32-point one in
// the packed buffer:  [ 32-point layout: bias tiles | pieces ][ 48-point layout: bias rows | PE tables | pieces ].
// Why: the launch is power-limited; 384 instead of 256 points per weight pass moves a third fewer weight
// bytes per FLOP (profiles/r01_pmc_ablations.md: +16 % on the trunk at equal matrix-pipe utilisation).
#pragma once
#include "mlp_internal.h"

namespace dn {

constexpr int kG48Waves = 8;
constexpr int kG48PointsPerWave = 48;
constexpr int kG48PointsPerWg = kG48Waves * kG48PointsPerWave;  // 384
constexpr int kG48XyzPieces = kXyzPanel / 32;                   // 2
constexpr int kG48DirPieces = 1;                                // 27 columns in one 32-deep piece
constexpr int kG48TableBytes = 1536;                            // [4 groups][16 slots] xyz + [4][8] dir entries of 16 B
constexpr int kG48InRows = 13;                                  // per-wave input rows: 7 + two sets of 3 view-direction rows

// Encoding column held in slot u of lane group g (= lane / 16), or -1 for padding.  The coordinate a slot encodes is
// (u + g) % 3, so that a lane rotates its point once ((x, y, z) -> starting at g % 3) and every slot then reads a
// compile-time element of the rotated point; which function of that coordinate (identity / sin f / cos f) is table
// data.  Per coordinate the slots, taken in (g, u) order, receive its columns of the reference encoding
// [x(3), sin(f0 x)(3), cos(f0 x)(3), ...] (nerf/nerf_helpers.py:132-159) in order: identity, sin f0, cos f0, sin f1, ...
__host__ __device__ inline int g48_pe_col(int kind, int g, int u, int L) {
  const int slots = kind == 1 ? 16 : 8;
  const int comp = (u + g) % 3;
  int rank = 0;
  for (int gg = 0; gg <= g; ++gg)
    for (int uu = 0; uu < (gg == g ? u : slots); ++uu) rank += ((uu + gg) % 3 == comp) ? 1 : 0;
  if (rank == 0) return comp;                       // identity column
  const int f = (rank - 1) / 2, is_cos = (rank - 1) % 2;
  return f < L ? 3 + 6 * f + 3 * is_cos + comp : -1;
}

// Hidden feature held by element e of lane group g in B piece k of a hidden vector: two 16-row output tiles (2k, 2k+1)
// make one 32-deep piece; the 16x16 accumulator keeps rows 4g..4g+3 of a tile in lane group g.
__host__ __device__ constexpr int g48_hidden_col(int k, int g, int e) { return (2 * k + e / 4) * 16 + 4 * g + (e % 4); }

struct G48Tables {   // what pack48 writes between the bias rows and the pieces
  float fx[16];
  float fd[8];
  int LX, LD;
};

struct G48Params {
  const char* base;  // start of the 48-point region of the packed buffer
  int bias_bytes;    // bias rows (padded to 1 KiB); the tables follow, then the pieces
  int total_pieces;
#ifdef DN_STAMP
  unsigned* dbg;     // diagnostic build: 8 words per wave
#endif
};


// NetLayout reused with 16-row tiles: n_tiles / bias0 count 16-row tiles, pieces_per_tile counts 32-deep pieces.
inline int build_layout48(const dn_mlp_desc& d, NetLayout* out) {
  const int W = d.hidden_size, D = d.num_layers;
  const int DX = 3 + 6 * d.num_encoding_fn_xyz, DD = 3 + 6 * d.num_encoding_fn_dir;
  NetLayout& L = *out;
  L = NetLayout{};
  L.W = W; L.LX = d.num_encoding_fn_xyz; L.LD = d.num_encoding_fn_dir; L.D = D; L.use_viewdirs = d.use_viewdirs;
  int piece = 0, bias = 0, s = 0;
  auto add = [&](int n_tiles, int hidden_in, int pe_kind, int src, int src2, int n_real, int ld, int col_h0, int col_p0) {
    StageDesc& t = L.st[s++];
    t = StageDesc{};
    t.n_tiles = n_tiles; t.hidden_in = hidden_in; t.pe_kind = pe_kind; t.src = src; t.src2 = src2;
    t.n_real = n_real; t.ld = ld; t.col_hidden0 = col_h0; t.col_pe0 = col_p0;
    t.first_tile2 = (src2 >= 0) ? 0 : -1;
    t.pieces_per_tile = hidden_in / 32 + (pe_kind == 1 ? kG48XyzPieces : (pe_kind == 2 ? kG48DirPieces : 0));
    t.piece0 = piece; t.bias0 = bias;
    piece += n_tiles * t.pieces_per_tile;
    bias += n_tiles;
  };
  const int NT = W / 16;
  add(NT, 0, 1, 0, -1, W, DX, 0, 0);
  for (int i = 0; i < D - 1; ++i) {
    const bool wide = (i % d.skip_connect_every == 0) && i > 0 && i != D - 1;
    if (wide) L.skip_mask |= (1u << i);
    add(NT, W, wide ? 1 : 0, 1 + i, -1, W, wide ? W + DX : W, 0, W);
  }
  if (d.use_viewdirs) {
    const int i_dir = D, i_alpha = D + 1, i_rgb = D + 2, i_feat = D + 3;
    add(NT + 1, W, 0, i_feat, i_alpha, W, W, 0, 0);      // tile 0: row 0 = fc_alpha; tiles 1..NT: fc_feat
    add(NT / 2, W, 2, i_dir, -1, W / 2, W + DD, 0, W);   // layers_dir.0 on cat(feat, view)
    add(1, W / 2, 0, i_rgb, -1, 3, W / 2, 0, 0);         // fc_rgb
  } else {
    add(1, W, 0, D, -1, 4, W, 0, 0);                     // fc_out
  }
  L.n_stages = s;
  L.total_pieces = round_up(piece, kPhasePieces);
  L.total_bias_tiles = bias;
  L.bias_bytes = round_up(bias * 64, 1024);
  return 0;
}

// LDS of the forward kernel: weight ring | bias rows | tables | xyz-encoding stash | input rows
inline size_t g48_lds_bytes(const NetLayout& L) {
  return static_cast<size_t>(kRingBytes) + L.bias_bytes + kG48TableBytes + kG48Waves * 3 * kG48XyzPieces * kPieceBytes +
         kG48Waves * kG48InRows * kG48PointsPerWave * sizeof(float);
}

// bf16 / fp16 nets whose bias rows leave room for the stash in 160 KiB of LDS (W = 256: D <= 9 with view directions)
inline bool g48_supported(const dn_mlp_desc& d, int precision) {
  if ((precision != DN_PREC_BF16 && precision != DN_PREC_F16) || (d.hidden_size != 256 && d.hidden_size != 128)) return false;
  NetLayout L;
  build_layout48(d, &L);
  return g48_lds_bytes(L) <= 160 * 1024;
}

inline size_t g48_region_bytes(const dn_mlp_desc& d) {
  NetLayout L;
  build_layout48(d, &L);
  return static_cast<size_t>(L.bias_bytes) + kG48TableBytes + static_cast<size_t>(L.total_pieces) * kPieceBytes;
}

bool g48_range_guard_complete(const dn_mlp_desc& d);
int launch_pack48(const dn_mlp_desc& d, int precision, const PackPtrs& ptrs, char* region, hipStream_t stream);
int launch_forward48(const dn_mlp_desc& d, int precision, const FwdParams& p, const char* region, hipStream_t stream);

}  // namespace dn
