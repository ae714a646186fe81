// Layout of the FlexibleNeRFModel "weight stream" consumed by the fused PE+MLP kernel.
//
// The network (reference nerf/models.py:185-256) is evaluated TRANSPOSED: for a tile of 32 sample
// points per wave, Y^T[n_out x 32] = W[n_out x K] . X^T[K x 32], so that
//   * the A operand of every MFMA is a 32-row slab of an nn.Linear weight ((out,in) row-major - the
//     layout PyTorch already stores), pre-permuted once per optimizer step into 1 KiB "pieces"
//     (64 lanes x 16 B, lane-linear so a single LDS-DMA instruction lands one piece), and
//   * the B operand is the previous layer's accumulator tile itself (column = lane = sample point,
//     rows = registers), so activations never leave the register file between layers
//     (guide: "An accumulator tile as the next MFMA's operand").
//
// A piece feeds one v_mfma_f32_32x32x16_bf16 (bf16: 8 k-values per lane) or four
// v_mfma_f32_32x32x2_f32 (fp32: 4 k-values per lane).  The k order inside a stage is whatever the
// accumulator layout dictates; the pack kernel applies the same permutation to the weight columns.
#pragma once
#include "dn_common.h"

namespace dn {

constexpr int kPieceBytes = 1024;
constexpr int kPhasePieces = 16;  // pieces per pipeline phase (one ring slot = 16 KiB)
constexpr int kMaxStages = 40;

__host__ __device__ constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Padded width of the xyz positional encoding as a K panel.  Fixed at 64 (L_xyz <= 10) rather than round_up(3+6L,16):
// every stage then holds a multiple of 16 pieces for W in {128, 256}, so the trunk stages all start on a pipeline
// phase boundary (an L=6 panel of 48 would leave W=128 stages at 12 / 24 pieces).
constexpr int kXyzPanel = 64;

// Row of a 32x32 accumulator tile held in register r of a lane in half h (C/D layout of the 32x32 MFMAs).
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Positional-encoding "slot" u of lane-half h -> column of the reference encoding
// [x(3), sin(f0 x)(3), cos(f0 x)(3), ...] (nerf/nerf_helpers.py:132-159), or -1 for padding.
// Half 0 owns the first L/2 frequencies + identity x,y ; half 1 the other L/2 + identity z (L even).
__host__ __device__ inline int pe_slot_col(int L, int h, int u) {
  const int nf = L / 2;
  if (u < 6 * nf) return 3 + 6 * ((h ? nf : 0) + u / 6) + (u % 6);
  const int v = u - 6 * nf;
  if (h == 0) return (v < 2) ? v : -1;
  return (v == 0) ? 2 : -1;
}

struct StageDesc {
  int32_t n_tiles;      // output tiles of 32 rows
  int32_t hidden_in;    // width of the hidden input block (0, W or W/2)
  int32_t pe_kind;      // 0 none, 1 xyz, 2 dir
  int32_t src;          // index into the weight pointer list
  int32_t src2;         // second source for the extra tile (fc_alpha), or -1
  int32_t n_real;       // real output rows taken from src
  int32_t ld;           // in_features of src
  int32_t col_hidden0;  // first column of the hidden block in src
  int32_t col_pe0;      // first column of the PE block in src
  int32_t first_tile2;  // tile index served by src2 (n_tiles-1 -> stored FIRST in the stream), or -1
  int32_t piece0;       // first piece of this stage in the stream
  int32_t bias0;        // first bias tile of this stage
  int32_t pieces_per_tile;
  // backward (dX) stages: A[i][k] = W[k][row0 + i] (the nn.Linear weight read transposed); `hidden_in` then counts
  // the forward layer's OUTPUT features (the k dimension), n_real its hidden INPUT features (the rows).
  int32_t transposed;
  int32_t custom_k;     // extra "custom" input piece: rows of src2 (fc_alpha: 1) or of src itself (fc_rgb: 3)
};

struct NetLayout {
  int32_t n_stages;
  int32_t total_pieces;  // multiple of kPhasePieces
  int32_t total_bias_tiles;
  int32_t bias_bytes;    // padded to 1 KiB
  int32_t W, LX, LD, D, use_viewdirs;
  uint32_t skip_mask;    // bit i set: layers_xyz[i] takes cat(x, xyz)
  StageDesc st[kMaxStages];
};

// kpp = k-values per piece: 16 (bf16) or 8 (fp32).
inline int build_layout(const dn_mlp_desc& d, int precision, NetLayout* out) {
  const int kpp = (precision != DN_PREC_F32) ? 16 : 8;
  const int W = d.hidden_size, D = d.num_layers;
  const int KX = kXyzPanel;
  const int KD = round_up(3 + 6 * d.num_encoding_fn_dir, 16);
  const int DX = 3 + 6 * d.num_encoding_fn_xyz, DD = 3 + 6 * d.num_encoding_fn_dir;
  NetLayout& L = *out;
  L = NetLayout{};
  L.W = W; L.LX = d.num_encoding_fn_xyz; L.LD = d.num_encoding_fn_dir; L.D = D; L.use_viewdirs = d.use_viewdirs;
  int piece = 0, bias = 0, s = 0;
  auto add = [&](int n_tiles, int hidden_in, int pe_kind, int src, int src2, int n_real, int ld, int col_h0,
                 int col_p0) {
    StageDesc& t = L.st[s++];
    t.n_tiles = n_tiles; t.hidden_in = hidden_in; t.pe_kind = pe_kind; t.src = src; t.src2 = src2;
    t.n_real = n_real; t.ld = ld; t.col_hidden0 = col_h0; t.col_pe0 = col_p0;
    t.first_tile2 = (src2 >= 0) ? n_tiles - 1 : -1;
    const int kpe = pe_kind == 1 ? KX : (pe_kind == 2 ? KD : 0);
    t.pieces_per_tile = (hidden_in + kpe) / kpp;
    t.piece0 = piece; t.bias0 = bias;
    piece += n_tiles * t.pieces_per_tile;
    bias += n_tiles;
  };
  const int NT = W / 32;
  // parameter order of the reference: layer1, layers_xyz.*, [layers_dir.0, fc_alpha, fc_rgb, fc_feat | fc_out]
  add(NT, 0, 1, 0, -1, W, DX, 0, 0);
  for (int i = 0; i < D - 1; ++i) {
    const bool wide = (i % d.skip_connect_every == 0) && i > 0 && i != D - 1;
    if (wide) L.skip_mask |= (1u << i);
    add(NT, W, wide ? 1 : 0, 1 + i, -1, W, wide ? W + DX : W, 0, W);
  }
  if (d.use_viewdirs) {
    const int i_dir = D, i_alpha = D + 1, i_rgb = D + 2, i_feat = D + 3;
    add(NT + 1, W, 0, i_feat, i_alpha, W, W, 0, 0);          // fc_feat rows + one extra tile: row 0 = fc_alpha
    add(NT / 2, W, 2, i_dir, -1, W / 2, W + DD, 0, W);       // layers_dir.0 on cat(feat, view)
    add(1, W / 2, 0, i_rgb, -1, 3, W / 2, 0, 0);             // fc_rgb
  } else {
    add(1, W, 0, D, -1, 4, W, 0, 0);                         // fc_out
  }
  L.n_stages = s;
  L.total_pieces = round_up(piece, kPhasePieces);
  L.total_bias_tiles = bias;
  L.bias_bytes = round_up(bias * 128, 1024);
  return 0;
}

// Saved-activation slots of the training forward (units: pieces of 1 KiB per 32-point tile), in this order:
//   [xyz PE | dir PE | layer1 out | trunk 0..D-2 out | fc_feat out | layers_dir.0 out]
// and one 16-byte ReLU bit-mask word per lane per masked stage (trunk 0..D-2, fc_feat, layers_dir.0).
// ReLU mask words (training): one 128-bit word per lane per masked stage.  Bit of accumulator register r (0..15) of
// output tile nt: chosen so that the 16-bit-mode forward can build it from the PACKED outputs (dword j = r/2 of the
// tile's two pieces holds elements r = 2j (low half) and 2j+1 (high half): v_pk_min_u16(dword, 1) << (j + 8*(nt&1))).
__host__ __device__ constexpr int relu_mask_bit(int nt, int r) { return 32 * (nt / 2) + (r & 1) * 16 + (r >> 1) + 8 * (nt & 1); }

struct TrainLayout {
  int32_t kpp, epp, ppt;
  int32_t kxp, kdp, kh;             // pieces: xyz PE, dir PE, a W-wide hidden vector
  int32_t act_pieces;               // saved forward pieces per 32-point tile
  int32_t slot_xyz, slot_dir, slot_layer1, slot_trunk0, slot_feat, slot_dirout;
  int32_t mask_words;               // 16-byte mask words per lane per tile: (D-1) trunk + feat + dirout
  int32_t grad_pieces;              // saved dL/d(pre-activation) pieces per tile
  int32_t gslot_dirout, gslot_feat, gslot_trunk0, gslot_layer1;  // gslot_trunk0 + i*kh for layers_xyz[i]
  int32_t gslot_out;  // the output gradient as "custom" pieces: [d rgb | d alpha] (viewdirs) or [d out]
};

inline void build_train_layout(const dn_mlp_desc& d, int precision, TrainLayout* t) {
  const bool bf = precision != DN_PREC_F32;
  t->kpp = bf ? 16 : 8; t->epp = bf ? 8 : 4; t->ppt = bf ? 2 : 4;
  const int W = d.hidden_size, D = d.num_layers;
  t->kxp = kXyzPanel / t->kpp;
  t->kdp = d.use_viewdirs ? round_up(3 + 6 * d.num_encoding_fn_dir, 16) / t->kpp : 0;
  t->kh = W / t->kpp;
  int s = 0;
  t->slot_xyz = s; s += t->kxp;
  t->slot_dir = s; s += t->kdp;
  t->slot_layer1 = s; s += t->kh;
  t->slot_trunk0 = s; s += (D - 1) * t->kh;
  t->slot_feat = s; s += d.use_viewdirs ? t->kh : 0;
  t->slot_dirout = s; s += d.use_viewdirs ? t->kh / 2 : 0;
  t->act_pieces = s;
  t->mask_words = (D - 1) + (d.use_viewdirs ? 2 : 0);
  int g = 0;
  t->gslot_dirout = g; g += d.use_viewdirs ? t->kh / 2 : 0;
  t->gslot_feat = g; g += d.use_viewdirs ? t->kh : 0;
  t->gslot_trunk0 = g; g += (D - 1) * t->kh;
  t->gslot_layer1 = g; g += t->kh;
  t->gslot_out = g; g += d.use_viewdirs ? 2 : 1;
  t->grad_pieces = g;
}

// Backward-data stream (dL/dX chain), stages in the order the backward kernel consumes them:
//   fc_rgb^T, layers_dir.0^T (feat rows only), [fc_feat^T | fc_alpha^T], layers_xyz[D-2..0]^T (hidden rows only).
// (no-viewdirs nets: fc_out^T, then the trunk.)  No bias tiles.
inline int build_backward_layout(const dn_mlp_desc& d, int precision, NetLayout* out) {
  const int kpp = (precision != DN_PREC_F32) ? 16 : 8;
  const int W = d.hidden_size, D = d.num_layers;
  const int DX = 3 + 6 * d.num_encoding_fn_xyz, DD = 3 + 6 * d.num_encoding_fn_dir;
  NetLayout& L = *out;
  L = NetLayout{};
  L.W = W; L.LX = d.num_encoding_fn_xyz; L.LD = d.num_encoding_fn_dir; L.D = D; L.use_viewdirs = d.use_viewdirs;
  int piece = 0, s = 0;
  // n_rows: forward hidden-input width (rows of the transposed weight); k_hidden: forward output width fed by
  // accumulator pieces; custom: number of extra k columns taken from `src2` (or from `src` when k_hidden == 0)
  auto add = [&](int n_rows, int k_hidden, int custom, int src, int src2, int ld, int row0) {
    StageDesc& t = L.st[s++];
    t = StageDesc{};
    t.n_tiles = n_rows / 32; t.hidden_in = k_hidden; t.pe_kind = 0; t.src = src; t.src2 = src2;
    t.n_real = n_rows; t.ld = ld; t.col_hidden0 = row0; t.col_pe0 = 0; t.first_tile2 = -1;
    t.transposed = 1; t.custom_k = custom;
    t.pieces_per_tile = k_hidden / kpp + (custom > 0 ? 1 : 0);
    t.piece0 = piece; t.bias0 = 0;
    piece += t.n_tiles * t.pieces_per_tile;
  };
  for (int i = 0; i < D - 1; ++i)
    if ((i % d.skip_connect_every == 0) && i > 0 && i != D - 1) L.skip_mask |= (1u << i);
  if (d.use_viewdirs) {
    const int i_dir = D, i_alpha = D + 1, i_rgb = D + 2, i_feat = D + 3;
    add(W / 2, 0, 3, i_rgb, -1, W / 2, 0);            // d g      = fc_rgb^T d rgb
    add(W, W / 2, 0, i_dir, -1, W + DD, 0);           // d feat   = layers_dir.0[:, :W]^T d dirpre
    add(W, W, 1, i_feat, i_alpha, W, 0);              // d h      = fc_feat^T d featpre + fc_alpha^T d alpha
  } else {
    add(W, 0, 4, D, -1, W, 0);                        // d h      = fc_out^T d out
  }
  for (int i = D - 2; i >= 0; --i) {
    const bool wide = (L.skip_mask >> i) & 1u;
    add(W, W, 0, 1 + i, -1, wide ? W + DX : W, 0);    // d x_i    = layers_xyz[i][:, :W]^T d pre_i
  }
  L.n_stages = s;
  L.total_pieces = round_up(piece, kPhasePieces);
  L.total_bias_tiles = 0;
  L.bias_bytes = 0;
  return 0;
}

inline int validate_desc(const dn_mlp_desc* d, int precision) {
  if (!d) { set_error("mlp: NULL descriptor"); return DN_E_INVAL; }
  if (precision != DN_PREC_F32 && precision != DN_PREC_BF16 && precision != DN_PREC_F16) { set_error("mlp: unknown precision %d", precision); return DN_E_INVAL; }
  const bool ok = (d->hidden_size == 128 || d->hidden_size == 256) && d->num_layers >= 2 && d->num_layers <= 32 &&
                  d->skip_connect_every >= 1 && d->include_input_xyz == 1 && (d->include_input_dir == 1 || !d->use_viewdirs) &&
                  (d->num_encoding_fn_xyz == 10 || d->num_encoding_fn_xyz == 6) &&
                  (!d->use_viewdirs || d->num_encoding_fn_dir == 4);
  if (!ok) {
    set_error("mlp: configuration outside the fused HIP kernel (need W in {128,256}, 2<=D<=32, include_input, "
              "L_xyz in {10,6}, L_dir=4): W=%d D=%d L_xyz=%d L_dir=%d", d->hidden_size, d->num_layers,
              d->num_encoding_fn_xyz, d->num_encoding_fn_dir);
    return DN_E_UNSUPPORTED;
  }
  return 0;
}

}  // namespace dn
