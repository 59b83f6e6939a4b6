// Compile-time list scheduler of the "zipped" field backward (see umhs_field_zip.h) and the slot / operation tables of its kernels.
// Pure constexpr C++17, no device code: tools/zip_plan_dump.cpp prints the plans on the host.
#pragma once

namespace zip {
constexpr int KIND_M = 0, KIND_V = 1;
struct Op {
  int kind, job, idx;
  int ready;   // first slot the operation may run in (its chain-side operands exist from there on)
  int dep[3];  // operations that must have run at least `lat` slots earlier (-1: none)
  int lat;
  int ovw;     // slot in which the chain starts to overwrite its chain-side operands (the NEXT tile's, for a carried operation); -1: none
};

template <int NOPS, int NSLOTS>
struct Plan {
  int slot_of[NOPS];
  int m_at[2 * NSLOTS], v_at[2 * NSLOTS];  // slot s >= NSLOTS: slot s - NSLOTS of the wave's NEXT tile
  bool ok;
  int last;
};

// List scheduling in operation order: slot by slot, the first not yet placed operation of each kind whose operands are ready.
template <int NOPS, int NSLOTS>
constexpr Plan<NOPS, NSLOTS> make_plan(const Op (&ops)[NOPS]) {
  Plan<NOPS, NSLOTS> p{};
  for (int i = 0; i < NOPS; ++i) p.slot_of[i] = -1;
  for (int s = 0; s < 2 * NSLOTS; ++s) p.m_at[s] = p.v_at[s] = -1;
  int done = 0;
  p.last = 0;
  for (int s = 0; s < 2 * NSLOTS && done < NOPS; ++s) {
    for (int kind = KIND_V; kind >= KIND_M; --kind) {  // inside a slot the VALU operation runs first
      for (int i = 0; i < NOPS; ++i) {
        const Op& o = ops[i];
        if (o.kind != kind || p.slot_of[i] >= 0 || o.ready > s) continue;
        // an operation carried into the next tile's slots must run before anything it reads is overwritten there: chain-side
        // operands from slot ovw on, operand tiles by the next tile's own producer operation
        if (s >= NSLOTS && o.ovw >= 0 && !(s - NSLOTS < o.ovw)) continue;
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
          const int j = o.dep[d];
          if (j >= 0 && (p.slot_of[j] < 0 || p.slot_of[j] + o.lat > s)) ok = false;
          if (j >= 0 && s >= NSLOTS && p.slot_of[j] >= 0 && !(s - NSLOTS < p.slot_of[j])) ok = false;
        }
        if (!ok) continue;
        p.slot_of[i] = s, (kind == KIND_M ? p.m_at : p.v_at)[s] = i, ++done, p.last = s;
        break;
      }
    }
  }
  p.ok = done == NOPS;
  return p;
}
}  // namespace zip


namespace zp1 {
// Slots of the chain.  A three-piece split of a value pair is P3 pinned steps with a slot behind each -- [ReLU / mask, cvt hi]
// [unpack hi] [residual, cvt mid] [unpack mid] [residual, cvt lo] -- a two-piece one the first P2 of them: 2-3 VALU instructions per
// slot, about what issues beside one MFMA of the matrix pipe (8 of its 16 cycles hold the issue port, a lone wave's VALU takes 4 each).
constexpr int P3 = 5, P2 = 3;
constexpr int M_AT = 2;  // step of a pair behind which its mid piece exists (and with it everything a transpose needs)
constexpr int S_PE = 0;                   // positional encoding: 2 slots
constexpr int S_ENC = S_PE + 2;           // hash features, 4 pairs x P3
constexpr int S_I0 = S_ENC + 4 * P3;      // in27 pair 0 = (pe0, pe1)
constexpr int S_X0 = S_I0 + P3;           // tile pair (pe2, 0), two pieces
constexpr int S_H = S_X0 + P2;            // base hidden layer, 8 pairs x P3   (behind gemm B0)
constexpr int S_I1 = S_H + 8 * P3;        // in27 pairs 1..3                   (behind gemm B1)
constexpr int S_X1 = S_I1 + 3 * P3;       // tile pairs (bo0, bo1), (bo2, bo3), two pieces
constexpr int S_A1 = S_X1 + 2 * P2;       // feature hidden 1, 8 pairs x P3    (behind gemm F0)
constexpr int S_A2 = S_A1 + 8 * P3;       // feature hidden 2, 8 pairs x P2    (behind gemm F1)
constexpr int S_O = S_A2 + 8 * P2;        // d(feature logits), 2 pairs x P2
constexpr int S_Z1 = S_O + 2 * P2;        // dZ of feature hidden 2, 8 pairs x P3 (behind the fp32 gemm T_F2)
constexpr int S_Z0 = S_Z1 + 8 * P3;       // dZ of feature hidden 1            (behind gemm T_F1)
constexpr int S_B1 = S_Z0 + 8 * P3;       // dZ of the base outputs: 1 slot + 2 pairs x P2 (behind gemm T_F0)
constexpr int S_ZB = S_B1 + 1 + 2 * P2;   // dZ of the base hidden layer, 8 pairs x P3 (behind the fp32 gemm T_B1)
constexpr int S_END = S_ZB + 8 * P3;      // behind gemm T_B0: stores, N_END slots
constexpr int N_END = 6;
constexpr int NSLOTS = S_END + N_END;

// tiles (transposed operands): index into the tile tables below
enum Tile { T_E0 = 0, T_E1, T_HS0, T_HS1, T_HS2, T_HS3, T_X0, T_X1, T_A10, T_A11, T_A12, T_A13, T_A20, T_A21, T_A22, T_A23,  // X operands
            T_ZO, T_Z10, T_Z11, T_Z12, T_Z13, T_Z00, T_Z01, T_Z02, T_Z03, T_ZB1, T_ZB00, T_ZB01, T_ZB02, T_ZB03, NTILES };
// first slot of the tile's first pair (the chain starts to overwrite its pieces there); the pitch of its pairs
constexpr int tile_first(int t) {
  if (t <= T_E1) return S_ENC + 2 * P3 * t;
  if (t <= T_HS3) return S_H + 2 * P3 * (t - T_HS0);
  if (t == T_X0) return S_I0;
  if (t == T_X1) return S_X1;
  if (t <= T_A13) return S_A1 + 2 * P3 * (t - T_A10);
  if (t <= T_A23) return S_A2 + 2 * P2 * (t - T_A20);
  if (t == T_ZO) return S_O;
  if (t <= T_Z13) return S_Z1 + 2 * P3 * (t - T_Z10);
  if (t <= T_Z03) return S_Z0 + 2 * P3 * (t - T_Z00);
  if (t == T_ZB1) return S_B1 + 1;
  return S_ZB + 2 * P3 * (t - T_ZB00);
}
constexpr int tile_pitch(int t) { return (t == T_X1 || (t >= T_A20 && t <= T_ZO) || t == T_ZB1) ? P2 : P3; }
// slot from which both pieces (hi, mid) of the tile's two pairs exist: one behind the mid piece of its second pair
constexpr int tile_ready(int t) {
  if (t == T_X0) return S_X0 + M_AT + 1;  // (pe0, pe1) = in27 pair 0, then the two-piece pair (pe2, 0)
  return tile_first(t) + tile_pitch(t) + M_AT + 1;
}
// The X operands (activations) are transposed late, next to the dZ tiles of their dW job -- not when the forward recompute produces
// them: either their bf16 pieces or their packed tiles occupy registers in between (the same 4 per tile), and a tile packed early
// would be overwritten by the next tile before the products carried into that tile's first slots have read it.
constexpr int tile_gate(int t) {
  if (t <= T_E1) return S_ZB;
  if (t <= T_HS3) return S_B1;
  if (t <= T_X1) return S_Z0;
  return 0;  // (the operands of the jobs that finish inside their tile are transposed as soon as they exist)
}
constexpr bool tile_colsum(int t) { return t >= T_ZO; }
// operand tiles of the jobs that may be carried into the next tile (mlp_base's two layers, last in the chain) live in AGPRs
constexpr bool tile_in_agpr(int t) { return t <= T_HS3 || t >= T_ZB1; }

// jobs of the off-chain operations
enum Job { J_TR = 0,   // transposing MFMA: idx = 2 * tile + piece
           J_PACK,     // pack (+ column sums) of a tile: idx = tile
           J_DW_F2, J_DW_F1, J_DW_F0, J_DW_B1, J_DW_B0 };
constexpr int N_TR = 2 * NTILES, N_PACK = NTILES;
constexpr int N_DW[5] = {12, 48, 24, 12, 24};
constexpr int NOPS = N_TR + N_PACK + 12 + 48 + 24 + 12 + 24;
constexpr int OP_PACK0 = N_TR, OP_DW0 = N_TR + N_PACK;

struct OpTable {
  zip::Op op[NOPS];
};
// (Z tile base, TO, X tile base, TI) of the five dW jobs
constexpr int DW_Z[5] = {T_ZO, T_Z10, T_Z00, T_ZB1, T_ZB00}, DW_TO[5] = {1, 4, 4, 1, 4};
constexpr int DW_X[5] = {T_A20, T_A10, T_X0, T_HS0, T_E0}, DW_TI[5] = {4, 4, 2, 4, 2};
constexpr OpTable make_ops() {
  OpTable t{};
  int n = 0;
  // program order = priority: every tile's transposes and pack in the order the chain produces the tiles, each dW job behind the
  // pack of its last operand tile
  auto tile_ops = [&](int tile) {
    for (int piece = 0; piece < 2; ++piece) t.op[2 * tile + piece] = zip::Op{zip::KIND_M, J_TR, 2 * tile + piece, tile_ready(tile) > tile_gate(tile) ? tile_ready(tile) : tile_gate(tile), {-1, -1, -1}, 1, tile_first(tile)};
    t.op[OP_PACK0 + tile] = zip::Op{zip::KIND_V, J_PACK, tile, 0, {2 * tile, 2 * tile + 1, -1}, 3, -1};
  };
  for (int tile = 0; tile < NTILES; ++tile) tile_ops(tile);
  n = OP_DW0;
  for (int j = 0; j < 5; ++j)
    for (int i = 0; i < N_DW[j]; ++i) {
      const int to = i / (3 * DW_TI[j]), ti = i % DW_TI[j];
      t.op[n++] = zip::Op{zip::KIND_M, J_DW_F2 + j, i, 0, {OP_PACK0 + DW_Z[j] + to, OP_PACK0 + DW_X[j] + ti, -1}, 2, -1};
    }
  return t;
}
// priority order for the scheduler = a permutation of the table
struct Order {
  zip::Op op[NOPS];
  int id[NOPS];  // original operation id of each entry
};
constexpr Order make_order() {
  const OpTable t = make_ops();
  Order o{};
  int n = 0;
  bool used[NOPS] = {};
  auto put = [&](int id) {
    if (!used[id]) o.op[n] = t.op[id], o.id[n] = id, used[id] = true, ++n;
  };
  auto tile = [&](int tl) { put(2 * tl), put(2 * tl + 1), put(OP_PACK0 + tl); };
  auto dw = [&](int j) {
    int base = OP_DW0;
    for (int k = 0; k < j; ++k) base += N_DW[k];
    for (int i = 0; i < N_DW[j]; ++i) put(base + i);
  };
  // transposes and packs first (they gate everything behind them and cannot wait long: their operands are chain registers), in the
  // order the chain produces the tiles; the dW products fill what is left, job by job
  for (int tl = 0; tl < NTILES; ++tl) tile(tl);
  for (int jd = 0; jd < 5; ++jd) dw(jd);
  // dependencies refer to original ids: remap them to positions in this order
  int pos[NOPS] = {};
  for (int i = 0; i < NOPS; ++i) pos[o.id[i]] = i;
  for (int i = 0; i < NOPS; ++i)
    for (int d = 0; d < 3; ++d)
      if (o.op[i].dep[d] >= 0) o.op[i].dep[d] = pos[o.op[i].dep[d]];
  return o;
}
constexpr Order ORDER = make_order();
constexpr zip::Plan<NOPS, NSLOTS> PLAN = zip::make_plan<NOPS, NSLOTS>(ORDER.op);
// only products whose operand tiles live in AGPRs may wait for the next tile
constexpr bool carried_in_agpr() {
  for (int i = 0; i < NOPS; ++i)
    if (PLAN.slot_of[i] >= NSLOTS && !(ORDER.op[i].job >= J_DW_F2 && tile_in_agpr(DW_Z[ORDER.op[i].job - J_DW_F2]) && tile_in_agpr(DW_X[ORDER.op[i].job - J_DW_F2])))
      return false;
  return true;
}
#ifndef ZIP_PLAN_NO_ASSERT  // (tools/zip_plan_dump.cpp prints a plan that does not close instead of failing to compile)
static_assert(PLAN.ok, "zipped part 1: the off-chain operations do not fit into two tiles' slots");
static_assert(carried_in_agpr(), "zipped part 1: an operation with VGPR operand tiles was carried into the next tile");
#endif
}  // namespace zp1

// ---------------------------------------------------------------------------------------------------------------------------------
// Part 0 (mlp_head + mlp_directional + mixing): only its two MLP stretches are zipped -- the head MLP's forward recompute and its
// backward (the same shapes as the feature MLP's in part 1); the stretch between them (head epilogue, directional layers, band tiles,
// head outputs) keeps the order field_bwd_tf_kernel gives it.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace zp0 {
using zp1::M_AT;
using zp1::P2;
using zp1::P3;
constexpr int S_PE = 0;                  // positional encoding: 2 slots
constexpr int S_I = S_PE + 2;            // in27 = (pe0, pe1) (pe2, e0) (e1, e2) (e3, 0): 4 pairs x P3
constexpr int S_X = S_I + 4 * P3;        // tile pairs (pe2, 0) (e0, e1) (e2, e3), two pieces
constexpr int S_A1 = S_X + 3 * P2;       // head hidden 1, 8 pairs x P3     (behind gemm H0)
constexpr int S_A2 = S_A1 + 8 * P3;      // head hidden 2, 8 pairs x P3     (behind gemm H1; gemm H2 reads all three pieces)
constexpr int S_O = S_A2 + 8 * P3;       // d(head logits), 2 pairs x P2    (behind the unzipped middle stretch)
constexpr int S_Z1 = S_O + 2 * P2;       // dZ of head hidden 2             (behind the fp32 gemm T_H2)
constexpr int S_Z0 = S_Z1 + 8 * P3;      // dZ of head hidden 1             (behind gemm T_H1)
constexpr int S_END = S_Z0 + 8 * P3;     // behind gemm T_H0: store, N_END slots
constexpr int N_END = 6;
constexpr int NSLOTS = S_END + N_END;

enum Tile { T_X0 = 0, T_X1, T_A10, T_A11, T_A12, T_A13, T_A20, T_A21, T_A22, T_A23, T_ZO, T_Z10, T_Z11, T_Z12, T_Z13, T_Z00, T_Z01, T_Z02, T_Z03, NTILES };
constexpr int tile_first(int t) {
  if (t == T_X0) return S_I;
  if (t == T_X1) return S_X + P2;
  if (t <= T_A13) return S_A1 + 2 * P3 * (t - T_A10);
  if (t <= T_A23) return S_A2 + 2 * P3 * (t - T_A20);
  if (t == T_ZO) return S_O;
  if (t <= T_Z13) return S_Z1 + 2 * P3 * (t - T_Z10);
  return S_Z0 + 2 * P3 * (t - T_Z00);
}
constexpr int tile_pitch(int t) { return (t == T_X1 || t == T_ZO) ? P2 : P3; }
constexpr int tile_ready(int t) {
  if (t == T_X0) return S_X + M_AT + 1;  // (pe0, pe1) = in27 pair 0, then the two-piece pair (pe2, 0)
  return tile_first(t) + tile_pitch(t) + M_AT + 1;
}
constexpr int tile_gate(int t) { return t <= T_X1 ? S_Z0 : 0; }  // the operands of the carried job are transposed late (see zp1)
constexpr bool tile_colsum(int t) { return t >= T_ZO; }
constexpr bool tile_in_agpr(int t) { return t <= T_X1 || t >= T_Z00; }  // job H0 (last in the chain) may wait for the next tile

enum Job { J_TR = 0, J_PACK, J_DW_H2, J_DW_H1, J_DW_H0 };
constexpr int NJOBS = 3;
constexpr int N_TR = 2 * NTILES, N_PACK = NTILES;
constexpr int N_DW[NJOBS] = {12, 48, 24};
constexpr int NOPS = N_TR + N_PACK + 12 + 48 + 24;
constexpr int OP_PACK0 = N_TR, OP_DW0 = N_TR + N_PACK;
constexpr int DW_Z[NJOBS] = {T_ZO, T_Z10, T_Z00}, DW_TO[NJOBS] = {1, 4, 4};
constexpr int DW_X[NJOBS] = {T_A20, T_A10, T_X0}, DW_TI[NJOBS] = {4, 4, 2};
struct Order {
  zip::Op op[NOPS];
};
constexpr Order make_order() {  // priority order: every tile's transposes and pack in chain order, then the dW jobs
  Order o{};
  for (int tile = 0; tile < NTILES; ++tile) {
    const int rdy = tile_ready(tile) > tile_gate(tile) ? tile_ready(tile) : tile_gate(tile);
    o.op[3 * tile + 0] = zip::Op{zip::KIND_M, J_TR, 2 * tile + 0, rdy, {-1, -1, -1}, 1, tile_first(tile)};
    o.op[3 * tile + 1] = zip::Op{zip::KIND_M, J_TR, 2 * tile + 1, rdy, {-1, -1, -1}, 1, tile_first(tile)};
    o.op[3 * tile + 2] = zip::Op{zip::KIND_V, J_PACK, tile, 0, {3 * tile, 3 * tile + 1, -1}, 3, -1};
  }
  int n = 3 * NTILES;
  for (int j = 0; j < NJOBS; ++j)
    for (int i = 0; i < N_DW[j]; ++i) {
      const int to = i / (3 * DW_TI[j]), ti = i % DW_TI[j];
      o.op[n++] = zip::Op{zip::KIND_M, J_DW_H2 + j, i, 0, {3 * (DW_Z[j] + to) + 2, 3 * (DW_X[j] + ti) + 2, -1}, 2, -1};
    }
  return o;
}
constexpr Order ORDER = make_order();
constexpr zip::Plan<NOPS, NSLOTS> PLAN = zip::make_plan<NOPS, NSLOTS>(ORDER.op);
constexpr bool carried_in_agpr() {
  for (int i = 0; i < NOPS; ++i)
    if (PLAN.slot_of[i] >= NSLOTS && !(ORDER.op[i].job >= J_DW_H2 && tile_in_agpr(DW_Z[ORDER.op[i].job - J_DW_H2]) && tile_in_agpr(DW_X[ORDER.op[i].job - J_DW_H2])))
      return false;
  return true;
}
#ifndef ZIP_PLAN_NO_ASSERT
static_assert(PLAN.ok, "zipped part 0: the off-chain operations do not fit into two tiles' slots");
static_assert(carried_in_agpr(), "zipped part 0: an operation with VGPR operand tiles was carried into the next tile");
#endif
}  // namespace zp0
