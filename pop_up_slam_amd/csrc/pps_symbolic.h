// pps_symbolic.h -- host-side symbolic analysis of the plane-SLAM normal equations.
//
// Replaces cholmod_analyze (reference: Thirdparty/isam/isamlib/Cholesky.cpp:98,105), which the
// reference re-runs on every factorisation.  Here it runs once per graph topology and is cached:
//   1. nested-dissection ordering of the node graph along the pose chain (planes that are seen
//      from both sides of a cut, and the cut pose, form the separator; very-high-degree nodes
//      such as the ground plane are pulled out first and become the root "border" block --
//      SURVEY.md section 7, hard part 2),
//   2. separator tree -> supernodes ("fronts") with bounded pivot counts,
//   3. boundary (update) index sets, child->parent scatter maps, level schedule,
//   4. block-sparse layout of H = J'J and the per-block contribution lists the device reduces.
// All index arrays are flat int32 so they upload to the GPU as-is.
#pragma once

#include <cstdint>
#include <vector>

namespace pps {

enum { NODE_POSE = 0, NODE_PLANE = 1 };
enum { F_POSE_PRIOR = 0, F_ODOMETRY = 1, F_PLANE_OBS = 2, F_PLANE_PRIOR = 3 };

// doubles of Jacobian storage per factor type: [J_a | J_b | r]
constexpr int kJSize[4] = {36 + 6, 36 + 36 + 6, 18 + 9 + 3, 9 + 3};
constexpr int kFDim[4] = {6, 6, 3, 3};
// doubles of a factor's PRODUCT record (round 4): what it adds to the diagonal H blocks of its nodes -- J_a' J_a (row-major, full)
// followed by -J_a' r for node a, then the same for node b.  K1 writes it next to the Jacobian, K2 only sums such records.
constexpr int kPSize[4] = {36 + 6, 2 * (36 + 6), (36 + 6) + (9 + 3), 9 + 3};
constexpr int kProductFlag = 256;     // added to a contribution's row count m: `ju` is the offset of a product sub-record (plane observations only, so far)

struct SymNode { int type; int dim; int rank; };        // rank: insertion index among poses (time order), -1 for planes
struct SymFactor { int type; int a, b; int joff; int direct_ok = 0; int poff = 0; };   // compact node ids (b = -1 when unary); joff / poff: offsets in the J / product buffers;
                                                                       // direct_ok: a plain plane observation (slot = joff / 30) whose pose-plane block K1 may write itself

struct AnalysisParams {
  int leaf_poses = 4;      // a sub-chain with <= leaf_poses poses becomes one leaf front
  int arity = 2;           // sub-chains per dissection step (2 = bisection; 4 = three cut poses form ONE separator front)
  int aligned_cuts = 0;    // bisection at absolute (power-of-two aligned) pose ranks instead of the cheapest cut near the middle of
                           // the sub-chain: the tree of a chain that grows at its end then keeps its left part (frame loops); the
                           // balanced, cost-driven cuts give the smaller fronts on a graph that is analysed once
  int max_pivots = 48;     // split supernodes with more pivot scalars into a chain
  int seg_len = 32;        // contributions reduced per wave in the H-block kernel
  int band_levels = 3;     // tree levels walked by one workgroup inside one launch ("band"); 0 = chosen by the analysis from the tree (Analysis::band_levels)
  int ordering = 0;        // 0 = pose-chain dissection, minimum degree as well when its fronts exceed band_rows (cheaper wins);
                           // 1 = minimum degree only; 2 = chain dissection only
  int band_rows = 127;     // largest front of the wave-per-front kernels: graphs beyond it skip the packed extend-add lists
  int front_rows = 63;     // rows (without the rhs row) a front should not exceed where the dissection can arrange it: every front then lives in
                           // the register tiles of ONE wave at two waves per SIMD (kRegRows, pps_regtile.h).  A would-be leaf beyond it is dissected
                           // further; a cut whose separator front would exceed it moves to the nearest position that fits -- anywhere in the
                           // sub-chain for cost-driven cuts, at most cut_shift poses from its aligned rank for aligned cuts.  0 = off.
  int cut_shift = 8;
  int dense_min = 64;      // a node is "dense" if degree > max(dense_min, dense_mult*sqrt(N))
  double dense_mult = 8.0;
  int timing = 0;          // per-phase host times to stderr (Switches::analysis_timing of the handle)
};

struct Analysis {
  // What this analysis left exactly as the analysis it was built upon had it (all 0 after an analysis from scratch): LEADING
  // entries of the index arrays, which an uploader that still holds the previous arrays need not compare again.
  //   fronts        f_p f_b f_poff f_Loff f_Uoff [fronts]; f_bidx_off f_child_off f_cmap_off f_ea_off [fronts + 1]; pidx [f_poff[fronts]];
  //                 bidx [f_bidx_off[fronts]]; child [f_child_off[fronts]]
  //   fronts_lists  f_asm_off f_el_off [fronts_lists + 1]; asm_blk asm_lrow asm_lcol asm_el0 asm_fsz [f_asm_off[fronts_lists]]
  //   blocks        blk_rows blk_cols blk_size blk_nseg blk_hoff [blocks]; blk_doff [blocks + 1]
  //   segs          seg_blk seg_c0 seg_cnt seg_hoff [segs]; srec [8 segs];   contribs: contrib [4 contribs];   nd_segs: nd_segs
  struct Kept { int fronts = 0, fronts_lists = 0, blocks = 0, segs = 0, contribs = 0, nd_segs = 0; } kept;
  int n_nodes = 0, n_scalars = 0;
  std::vector<int> node_pos;    // [n_nodes] elimination position
  std::vector<int> node_voff;   // [n_nodes] scalar offset of the node in delta: nodes in CREATION order (compact id), so an offset
                                // never moves when nodes are appended
  std::vector<int> pidx;        // [n_scalars] elimination-ordered scalar (front s owns f_poff[s] .. + f_p[s]) -> index in delta
  std::vector<int> order;       // [n_nodes] order[pos] = node

  // ---- fronts (post-order: children before parents) ----
  int n_fronts = 0, n_levels = 0, max_front = 0;
  std::vector<int> f_p, f_b;        // pivot / boundary scalar counts
  std::vector<int> f_poff;          // elimination-order offset of the first pivot scalar (into pidx)
  std::vector<int> f_parent, f_level;
  std::vector<int64_t> f_Loff;      // offset of the (f+1) x p factor panel (row-major, ld = p)
  std::vector<int64_t> f_Uoff;      // offset of the (b+1) x (b+1) update matrix (row-major, ld = b+1)
  std::vector<int> f_bidx_off;      // [n_fronts+1] -> bidx
  std::vector<int> bidx;            // boundary scalar -> global scalar index
  std::vector<int> f_child_off;     // [n_fronts+1] -> child
  std::vector<int> child;
  std::vector<int> f_cmap_off;      // [n_fronts+1] -> cmap ; front c's boundary (+rhs) -> local index in parent
  std::vector<int> cmap;
  std::vector<int> level_off;       // [n_levels+1] -> level_fronts
  std::vector<int> level_fronts;
  std::vector<int> f_asm_off;       // [n_fronts+1] -> asm_* (original entries gathered by the front)
  std::vector<int> asm_blk, asm_lrow, asm_lcol;
  std::vector<int> asm_el0, asm_fsz;  // per assembled block: first index in the front-ordered H, rows of its front
  int64_t el_total = 0;              // length of el_tgt / Hf
  int64_t L_size = 0, U_size = 0;

  // ---- band schedule: levels [B*s, B*s+B) form stage s; inside a stage the connected sub-trees are
  // "groups", each walked by ONE workgroup (one wave per front, workgroup barrier between local levels) ----
  int band_levels = 3;     // levels per band of this analysis (AnalysisParams::band_levels, or the automatic choice)
  int n_stages = 0, n_groups = 0, n_glevels = 0;
  std::vector<int> stage_grp_off;    // [n_stages+1] -> groups
  std::vector<int> grp_lvl_off;      // [n_groups+1] -> local levels
  std::vector<int> glvl_front_off;   // [n_glevels+1] -> glvl_fronts
  std::vector<int> glvl_fronts;      // [n_fronts]
  std::vector<int> stage_max_front;  // [n_stages] largest front (scalars, without the rhs row)
  std::vector<int> stage_max_width;  // [n_stages] widest local level (fronts)

  // ---- flat gather / scatter lists of the wave-per-front kernels (front = packed lower triangle,
  // index(i,j) = i(i+1)/2 + j, last row = right-hand side) ----
  std::vector<int> blk_doff;         // [n_blocks+1] -> blk_dst : block element -> index in the front-ordered H ("Hf"), -1 = unused
  std::vector<int> blk_dst;          // left empty by analyze(): expanded on the device, or by expand_el_lists() for the dump
  std::vector<int> f_el_off;         // [n_fronts+1] -> el_src / el_tgt : original H entries of the front (= index range of Hf)
  std::vector<int> el_src;           // (dump layout only)
  std::vector<int> el_tgt;           // packed front index; bit 30 set = diagonal element (scaled by 1+lambda); expanded like blk_dst
  std::vector<int64_t> f_ea_off;     // [n_fronts+1] -> ea_tgt : this front's packed update matrix -> packed index in its parent
  std::vector<int> ea_tgt;           // left empty by analyze(): expanded on the device, or by expand_ea_tgt() for the dump
  int64_t ea_total = 0;              // its length

  // ---- packed metadata records: one coalesced load per wave instead of a chain of dependent loads ----
  // frec: 16 ints per front, in band-schedule order (index = position in glvl_fronts):
  //   0 front id, 1 p, 2 b, 3 el_off0, 4 el_off1, 5 first child record, 6 child count, 7 poff, 8 bidx_off,
  //   9/10 Loff lo/hi, 11/12 Uoff lo/hi, 13 number of packed entries of the own update matrix,
  //   14 position of the parent inside the group (counted from the group's first front) when it belongs to the same
  //   group (else -1), 15 cmap_off
  // crec: 8 ints per child edge: 0 packed entries of the child's update matrix, 1/2 its Uoff lo/hi, 3/4 its ea_off lo/hi
  // srec: 8 ints per H segment: 0 rows, 1 cols, 2 size, 3 c0, 4 cnt, 5 hoff (segment slot), 6 blk_doff, 7 nseg of the block
  std::vector<int> frec, crec, srec;

  // ---- level-per-launch form (many-graph batches): positions in frec of the fronts of tree level l with size class c --
  // 0: p + b <= 32, 1: <= 48, 2: larger -- at cls_fronts[cls_off[3 l + c] .. cls_off[3 l + c + 1]) ----
  std::vector<int> cls_off, cls_fronts;

  // ---- block-sparse H = J'J (lower triangle in elimination order) ----
  int n_blocks = 0;
  std::vector<int> blk_rows, blk_cols;   // dims (rows = later node, cols = earlier node); diagonal blocks carry g appended
  std::vector<int> blk_size;             // doubles per segment slot
  std::vector<int> blk_nseg;
  std::vector<int64_t> blk_hoff;         // offset of the block's first segment slot in the H buffer
  int n_segs = 0;
  std::vector<int> seg_blk, seg_c0, seg_cnt;
  std::vector<int64_t> seg_hoff;
  std::vector<int> contrib;              // 4 ints per contribution: jv, ju, roff, m -- a plain plane observation's contribution to a DIAGONAL block
                                         // carries the offset of its product sub-record for that node in ju and kProductFlag in m
  int64_t H_size = 0;
  int64_t J_size = 0, P_size = 0;

  // ---- "direct" blocks: the off-diagonal H block of a (pose, plane) pair that ONE plane observation contributes to is a
  // product of that factor's own two Jacobian blocks; the thread-per-factor sweep can write it itself (many-graph batches),
  // and the H-block kernel then only visits the segments listed in nd_segs ----
  std::vector<int> obs_dir;              // 3 ints per plane-observation slot: H offset (-1 = not direct), Hf offset, 1 if the pose is the row node
  std::vector<int> nd_segs;              // segments that are not direct
};

// What one analysis leaves behind for the next one of the same, grown graph (frame loops; opaque).
struct AnalysisCache;
AnalysisCache* analysis_cache_new();
void analysis_cache_free(AnalysisCache* c);
// fronts / tree nodes the last analysis with this cache took over from the one before (0 = it started from scratch)
void analysis_cache_stats(const AnalysisCache* c, int* fronts_reused, int* fronts_total);

// nodes/factors are the compacted live sets.  Returns false (with msg) on structural problems.
// cache (may be NULL): when the graph is the previous one plus appended nodes / factors (every new factor touches a new
// node) and prm.aligned_cuts is set, the part of the elimination tree left of the new poses -- with its fronts, boundaries,
// H blocks and all index arrays of `out`, which must still hold the previous result -- is kept and only the rest is redone.
// The result is identical, array for array, to an analysis from scratch.
bool analyze(const std::vector<SymNode>& nodes, const std::vector<SymFactor>& factors, const AnalysisParams& prm,
             Analysis& out, const char** msg, AnalysisCache* cache = nullptr);

// ea_tgt from cmap / f_ea_off on the host (what k_expand_ea does on the device)
void expand_ea_tgt(Analysis& a);
// el_tgt / blk_dst from the per-block records (what k_expand_el does on the device)
void expand_el_lists(Analysis& a);
// Serialise the analysis into one int32 vector for host-logic tests (pps_analysis_dump).
void dump_analysis(const Analysis& a, std::vector<int32_t>& out);

}  // namespace pps
