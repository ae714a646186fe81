// The 48-points-per-wave bf16 inference geometry (mlp_fused48.hip): v_mfma_f32_16x16x32_bf16, 16-row output tiles,
// 32-deep A pieces, every A fragment feeding three MFMAs (three groups of 16 points).  Same network, same 1 KiB piece
// stream discipline as mlp_layout.h, different tile shape - so it has its own stream, stored behind the 32-point one in
// the packed buffer:  [ 32-point layout: bias tiles | pieces ][ 48-point layout: bias rows | PE tables | pieces ].
// Why: the launch is power-limited; 384 instead of 256 points per weight pass moves a third fewer weight
// bytes per FLOP (profiles/r01_pmc_ablations.md: +16 % on the trunk at equal matrix-pipe utilisation).
#pragma once
#include "mlp_internal.h"
#include "composite_body.h"

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
  CompParams comp;   // the instances that composite their own rays (COMP): where the maps go; otherwise unread
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

// Workgroups (of 256 threads) of a pack launch: two output elements per thread - an element is a page of index arithmetic in
// front of one dependent load, so the launch is latency-bound and wants threads, not a grid-stride loop (a training step packs
// both streams of both networks every iteration: 27 + 16 us at 256 workgroups per network on the D8 / W256 nets)
inline unsigned pack48_blocks(const NetLayout& L) {
  const long long elems = static_cast<long long>(L.total_pieces) * 512;
  const long long b = (elems + 511) / 512;
  return static_cast<unsigned>(b < 64 ? 64 : (b > 4096 ? 4096 : b));
}

inline size_t g48_region_bytes(const dn_mlp_desc& d) {
  NetLayout L;
  build_layout48(d, &L);
  return static_cast<size_t>(L.bias_bytes) + kG48TableBytes + static_cast<size_t>(L.total_pieces) * kPieceBytes;
}

bool g48_range_guard_complete(const dn_mlp_desc& d);
// ---- training in the 48-point geometry: the 8-bit saved tensors of DN_PREC_BF16_S8 ("s8-48" layout) ------------------------
// A wave's 48 points are three 16-point groups; groups are numbered along the point sequence (group G = point / 16) and two
// consecutive groups form one 32-point record T = G / 2 - the unit of the weight-gradient kernel's K = 64 contraction.
// Saved unit (1 KiB = 64 lanes x 16 bytes) of group G, slot s: lane (g = lane / 16, j = lane % 16) holds, for point j of the
// group, the 8 + 8 bytes of B pieces 2s and 2s + 1 of the saved vector (byte b: piece 2s + b / 8, element b % 8 - feature
// g48_hidden_col(piece, g, element) of a hidden vector, slot g48_pe_col of an encoding panel) - at row g * 16 + (j ^ 8 (g & 1)) of
// the unit, not at its own lane's: the odd lane groups are stored with their two 8-point halves exchanged, which puts the
// weight-gradient kernel's paired reads on different LDS banks (mlp_device.h store16_unit48).  Address of a unit:
//   buffer + ((T * units_per_group + s) * 2 + (G & 1)) * 1 KiB
// so that, seen from the weight-gradient kernel, a record is a run of 2 * units_per_group 1-KiB units and the two groups'
// units of one slot sit side by side.  Buffers are sized for whole workgroup tiles (384 points = 12 records).
// ReLU masks: per wave tile (48 points) and masked stage two 1-KiB words, lane = 16 bytes:
//   word 0 = [group 0: lo, hi | group 1: lo, hi], word 1 = [group 2: lo, hi | 0, 0]; bit of accumulator register r of 16-row
//   tile nt: dword nt / 8, bit ((nt % 8) * 2 + r / 2) + 16 * (r % 2) - i.e. straight from the packed 16-bit ReLU outputs
//   (v_pk_min_u16(pair, 1) << position) and back onto packed pairs in the backward (((w >> position) & 0x00010001) * 0xFFFF).
// Behind the saved-gradient units of a launch: a 256-byte record (uint32 words).  The backward-data kernel records the scale it
// used and zeroes the three statistics words; the weight-gradient kernel (same stream, next) counts into them while it reads the
// gradients anyway: of the dY bytes of one 32-point record in sixteen, how many are non-zero, how many sit at e5m2's largest
// magnitude (saturated: the gradient was at or beyond 57344 / scale) and how many at its smallest (the edge of being flushed to
// zero: below 2^-17 / scale a gradient is lost).
constexpr int kS8BlockBytes = 256;
// The three statistics are kept in kS8BlockReplicas copies (word kS8BlockStats + 4 r + {0, 1, 2}; a workgroup adds to copy
// blockIdx % replicas, the reader sums the copies): atomics on ONE address serialise at ~10 ns each - with one set of counters and
// one atomic per wave they were 40-90 us of the D8/W256 training step.
enum { kS8BlockSaturated = 0, kS8BlockFloor = 1, kS8BlockSampled = 2,   // offsets inside a replica
       kS8BlockScale = 4,           // bits of the scale this launch used (the weight-gradient kernel divides it out)
       kS8BlockStats = 8,           // first replica
       kS8BlockReplicas = 8,        // words 8 .. 39
       kS8BlockPartials = 40,       // auto scale: kS8BlockPartialCount partial maxima of |upstream gradient| (bits)
       kS8BlockPartialCount = 24 };

struct TrainLayout48 {
  int32_t kh_u;                     // units of a W-wide hidden vector
  int32_t act_units;                // saved forward units per 16-point group
  int32_t slot_xyz, slot_dir, slot_layer1, slot_trunk0, slot_feat, slot_dirout;
  int32_t mask_stages;              // masked stages: (D-1) trunk + feat + dirout
  int32_t grad_units;               // saved dL/d(pre-activation) units per group
  int32_t gslot_dirout, gslot_feat, gslot_trunk0, gslot_layer1;   // gslot_trunk0 + i * kh_u for layers_xyz[i]
  int32_t gslot_out;                // custom unit [d rgb piece | d alpha piece] (viewdirs) or [d out piece | 0]
};

inline void build_train_layout48(const dn_mlp_desc& d, TrainLayout48* t) {
  const int W = d.hidden_size, D = d.num_layers;
  t->kh_u = W / 64;
  int s = 0;
  t->slot_xyz = s; s += 1;                                   // 64-wide xyz panel = two 32-deep pieces
  t->slot_dir = s; s += d.use_viewdirs ? 1 : 0;              // one piece: lanes g < 2 carry [own 8 bytes | the 8 bytes of g + 2]
  t->slot_layer1 = s; s += t->kh_u;
  t->slot_trunk0 = s; s += (D - 1) * t->kh_u;
  t->slot_feat = s; s += d.use_viewdirs ? t->kh_u : 0;
  t->slot_dirout = s; s += d.use_viewdirs ? W / 128 : 0;
  t->act_units = s;
  t->mask_stages = (D - 1) + (d.use_viewdirs ? 2 : 0);
  int g = 0;
  t->gslot_dirout = g; g += d.use_viewdirs ? W / 128 : 0;
  t->gslot_feat = g; g += d.use_viewdirs ? t->kh_u : 0;
  t->gslot_trunk0 = g; g += (D - 1) * t->kh_u;
  t->gslot_layer1 = g; g += t->kh_u;
  t->gslot_out = g; g += 1;
  t->grad_units = g;
}

// 32-point records the buffers of an n_points launch hold: whole workgroup tiles of whichever training tile shape pads further - 384
// points (three point groups per wave) or 256 (two: g48_train_groups)
inline long long g48_padded_records(long long n_points) {
  const long long r3 = (n_points + 383) / 384 * 12, r2 = (n_points + 255) / 256 * 8;
  return r3 > r2 ? r3 : r2;
}
// wave tiles (one 2 KiB mask slot per masked stage each) the mask buffer of an n_points launch must hold: the 256-point tiling has the most
inline long long g48_mask_wave_tiles(long long n_points) {
  return (n_points + 255) / 256 * kG48Waves;
}
// Point groups per wave the TRAINING forward / backward of an n_points launch run on `cus` compute units: 3 (48 points per wave, 384 per
// workgroup tile - one pass of the weight stream serves the most points) unless 256-point tiles finish the launch in clearly fewer
// tile-rounds x points: launches of a few hundred tiles, where 384-point tiles leave units idle or make a short last round (a 1024-ray
// step of the as-shipped nets: 171 and 342 tiles on 256 units -> 256 and 512 tiles of two thirds the work).  Forward and backward
// of a launch must agree (the mask words are per wave tile): both call this.  DEXNERF_G48_TRAIN_GROUPS=2|3 (read per call) forces one.
// (only the two fixed shapes have two-group instances: the paper network and the as-shipped 4 x 128 nets, as mlp_fused48.hip knows them)
inline bool g48_two_group_shape(const dn_mlp_desc& d) {
  NetLayout L;
  build_layout48(d, &L);
  const bool paper = d.hidden_size == 256 && d.num_layers == 8 && L.skip_mask == 0x10u && d.use_viewdirs;
  const bool shipped = d.hidden_size == 128 && d.num_layers == 4 && L.skip_mask == 0u && d.use_viewdirs;
  return (paper || shipped) && std::getenv("DEXNERF_G48_RUNTIME_SHAPE") == nullptr;
}
inline int g48_train_groups(long long n_points, int cus) {
  if (const char* e = std::getenv("DEXNERF_G48_TRAIN_GROUPS")) {
    const int v = std::atoi(e);
    if (v == 2 || v == 3) return v;
  }
  if (cus < 1) cus = 1;
  const long long t3 = (n_points + 383) / 384, t2 = (n_points + 255) / 256;
  const long long c3 = (t3 + cus - 1) / cus * 3, c2 = (t2 + cus - 1) / cus * 2;
  // (a 256-point pass streams the same weights as a 384-point one and costs more than two thirds of it: measured on the D8/W256 step,
  //  the 262,144-point coarse launch - 4 rounds against 3, a ratio of 8 / 9 - gains nothing; the as-shipped step's launches - 2 / 3 - a quarter)
  return c2 * 100 <= c3 * 75 ? 2 : 3;
}

// Which networks train in the 48-point geometry (DN_PREC_BF16_S8): W = 128 needs whole 64-feature units for layers_dir.0's
// 64 outputs (it has them), both widths need L_xyz = 10 panels like the other training kernels
inline bool g48_train_supported(const dn_mlp_desc& d) {
  return g48_supported(d, DN_PREC_BF16) && d.num_encoding_fn_xyz == 10;
}

// Backward-data stream of the 48-point chain: 16-row tiles of the TRANSPOSED weights, 32-deep pieces whose k order is the
// forward layer's output features in accumulator order (g48_hidden_col), plus one optional custom piece (k = 8 g + e:
// d rgb / d alpha / d out rows).  Stages in consumption order, as build_backward_layout (mlp_layout.h).  No bias tiles.
inline int build_backward_layout48(const dn_mlp_desc& d, NetLayout* out) {
  const int W = d.hidden_size, D = d.num_layers;
  const int DX = 3 + 6 * d.num_encoding_fn_xyz, DD = 3 + 6 * d.num_encoding_fn_dir;
  NetLayout& L = *out;
  L = NetLayout{};
  L.W = W; L.LX = d.num_encoding_fn_xyz; L.LD = d.num_encoding_fn_dir; L.D = D; L.use_viewdirs = d.use_viewdirs;
  int piece = 0, s = 0;
  auto add = [&](int n_rows, int k_hidden, int custom, int src, int src2, int ld) {
    StageDesc& t = L.st[s++];
    t = StageDesc{};
    t.n_tiles = n_rows / 16; t.hidden_in = k_hidden; t.src = src; t.src2 = src2;
    t.n_real = n_rows; t.ld = ld; t.first_tile2 = -1; t.transposed = 1; t.custom_k = custom;
    t.pieces_per_tile = k_hidden / 32 + (custom > 0 ? 1 : 0);
    t.piece0 = piece;
    piece += t.n_tiles * t.pieces_per_tile;
  };
  for (int i = 0; i < D - 1; ++i)
    if ((i % d.skip_connect_every == 0) && i > 0 && i != D - 1) L.skip_mask |= (1u << i);
  if (d.use_viewdirs) {
    const int i_dir = D, i_alpha = D + 1, i_rgb = D + 2, i_feat = D + 3;
    add(W / 2, 0, 3, i_rgb, -1, W / 2);              // d g    = fc_rgb^T d rgb
    add(W, W / 2, 0, i_dir, -1, W + DD);             // d feat = layers_dir.0[:, :W]^T d dirpre
    add(W, W, 1, i_feat, i_alpha, W);                // d h    = fc_feat^T d featpre + fc_alpha^T d alpha
  } else {
    add(W, 0, 4, D, -1, W);                          // d h    = fc_out^T d out
  }
  for (int i = D - 2; i >= 0; --i) {
    const bool wide = (L.skip_mask >> i) & 1u;
    add(W, W, 0, 1 + i, -1, wide ? W + DX : W);      // d x_i  = layers_xyz[i][:, :W]^T d pre_i
  }
  L.n_stages = s;
  L.total_pieces = round_up(piece, kPhasePieces);
  return 0;
}

int launch_pack48(const dn_mlp_desc& d, int precision, const PackPtrs& ptrs, char* region, hipStream_t stream);
// comp != NULL: the caller would like the launch to composite its rays itself (rays + depths input, no density noise); *composited
// says whether it did (the fixed-shape instances, samples per ray dividing the 384-point workgroup tile) - if not, `out` holds the
// raw radiance field as always and the caller runs the compositing kernel
int launch_forward48(const dn_mlp_desc& d, int precision, const FwdParams& p, const char* region, hipStream_t stream,
                     const CompParams* comp = nullptr, int* composited = nullptr);
int backward48_entry(const dn_mlp_desc* desc, const void* packed_bwd, const float* g_out, const void* masks, int64_t n_points,
                     void* grads, float grad_scale, hipStream_t stream, const unsigned* partials = nullptr, int n_partials = 0);   // mlp_train48.hip
int unpack48_entry(const dn_mlp_desc* desc, int which, const void* native, int64_t n_points, int slot, int width, int kind, float* out,
                   int ld_out, int col0, hipStream_t stream);                             // mlp_train48.hip
int launch_pack48_pair(const dn_mlp_desc& d, const PackPtrs& a, const PackPtrs& b, char* region_a, char* region_b, hipStream_t stream);
int launch_pack48_backward_pair(const dn_mlp_desc& d, const PackPtrs& a, const PackPtrs& b, char* packed_a, char* packed_b, hipStream_t stream);
int launch_pack48_backward(const dn_mlp_desc& d, const PackPtrs& ptrs, char* packed, hipStream_t stream);   // mlp_train48.hip

}  // namespace dn
