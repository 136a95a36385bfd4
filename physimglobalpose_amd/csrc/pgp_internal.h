// csrc/pgp_internal.h -- shared declarations of libpgp.so (host side + kernel launchers).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pgp.h"

namespace pgp {

void set_error(const char* fmt, ...);

#define PGP_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) {                                                             \
      pgp::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PGP_EHIP;                                                                  \
    }                                                                                   \
  } while (0)

// Uniform grid over the (centred) scene cloud.  cell(x) = floor((x - origin) * inv_h).
// The origin lies on the lattice of pitch 1 / inv_h: origin = (k0 - 0.5) / inv_h, so that the cell is also
// round_to_nearest(x * inv_h) - k0 -- ONE fused multiply-add against the constant 1.5 * 2^23 - k0 leaves
// the cell number in the low mantissa bits (lcp_score.hip cell_bits); the two statements differ by float
// rounding only, which the dilation margin of the candidate lists covers (grid_index.hip choose_grid).
struct GridDesc {
  float ox, oy, oz;   // origin (min corner of cell (0,0,0))
  float h, inv_h;     // cell edge (>= delta) and its float reciprocal
  int nx, ny, nz;     // cells per axis
  int k0x, k0y, k0z;  // lattice number of cell 0 per axis
  int magic_ok;       // |k0| small enough for the mantissa trick (else the scoring falls back to cell_run)
  // Cells are numbered in BLOCKS of 4 x 4 x 2 (x, y, z) = 32 cells = one occupancy word, blocks in
  // (z, y, x) order: a compact patch of query points then touches few words, few run descriptors
  // and neighbouring candidate runs -- the vector L1 counts distinct lines per instruction.
  int nbx, nby, nbz;  // blocks per axis = ceil(nx/4), ceil(ny/4), ceil(nz/2)
  float reach;        // delta + margin: a point is a candidate of every cell within `reach`
  // SPARSE form (grid_index.hip, scenes whose bounding box is mostly empty): only the blocks that hold a
  // candidate exist, in an open-addressing table of uint4 {key, occupancy bits, rank base, slot} keyed by
  // bx | by << key_sy | bz << key_sz (block coordinates); `words` then points at that table.
  int sparse;
  int key_sy, key_sz;
  uint32_t tab_mask;  // table entries - 1 (a power of two)
  int tab_shift;      // 32 - log2(entries): slot = key * 2654435761 >> tab_shift
};
constexpr uint32_t kBlockEmpty = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t block_key(const GridDesc& g, uint32_t bx, uint32_t by, uint32_t bz) {
  return bx | (by << g.key_sy) | (bz << g.key_sz);
}
__host__ __device__ inline uint32_t block_hash(const GridDesc& g, uint32_t key) {
  return (key * 2654435761u) >> g.tab_shift;
}

// A growable device buffer (never shrinks; freed with the context).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes);  // returns PGP_OK / PGP_EHIP
  void release();
  template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace pgp

struct pgp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;  // internal stream for the synchronous host-pointer API
  // a *_device call returns with its kernels still queued on the caller's stream; the next host-pointer
  // entry point drains the device before it touches the context's arrays (pgp_api.hip CtxGuard)
  bool device_work_pending = false;
  // completion words in host memory (lcp_score.hip HostPub, pgp_api.hip publish_and_wait): the value the next wait looks for
  unsigned int flag_seq = 0;
  unsigned int* h_flag = nullptr;   // pinned completion word of pgp::publish_and_wait
  pgp::DevBuf d_pub_ticket;         // its arrival counter when several workgroups publish (zero between launches)

  // scene
  int nP = 0;
  bool has_scene_normals = false;
  float delta = 0.f;
  pgp::DevBuf d_P;       // float4 {x,y,z,bits(i)}             [nP]
  pgp::DevBuf d_Pnw;     // float4 {nx,ny,nz,w}                [nP]
  // index
  bool has_index = false;
  pgp::GridDesc grid{};
  long long n_cells = 0, n_cand = 0;
  pgp::DevBuf d_cell_start;  // uint32 [n_cells+1]  (build-time scratch: full-grid CSR)
  pgp::DevBuf d_cell_tmp;    // uint32 [n_cells+1]  (counts, then fill cursors)
  pgp::DevBuf d_scan_tmp;    // uint32 block sums
  pgp::DevBuf d_bitmap;      // uint2 {occupancy bits, rank base} [nbz*nby*nbx]  (sparse: per block slot, build only)
  pgp::DevBuf d_blocktab;    // sparse form: uint4 {key, bits, base, slot} [tab_mask + 1]
  long long n_blocks = 0;    // sparse form: blocks that hold a candidate
  pgp::DevBuf d_occ_start;   // uint2 {start, count} per occupied cell [n_occ]
  long long n_occ = 0;
  pgp::DevBuf d_cand;        // float4 {x,y,z,bits(i)} [n_cand]
  float build_ms = 0.f;
  // Small scenes (the drop-in's segments): the index is built on build_stream behind pgp_set_scene's uploads, with its
  // buffers sized by upper bounds so that nothing is read back in between; whatever reads the index waits for ev_index
  // on its own stream first (await_index, grid_index.hip) -- base selection and the congruent sets, which only read
  // the points, run beside the build.
  hipStream_t build_stream = nullptr;
  bool build_stream_own = false;   // false: the device's shared build stream (grid_index.hip build_index_async)
  hipEvent_t ev_index = nullptr, ev_build0 = nullptr;
  bool index_pending = false;
  // the side-stream build of the current scene, prepared but not yet QUEUED: its dozen-and-a-half launches cost the host
  // ~0.1 ms to issue, and pgp_set_scene's caller has better things to put on the device first (the weights, the base
  // selection).  Whoever is about to wait for the device (pgp::HostOut::sync), or needs the index (await_index /
  // finish_index), queues it first: pgp::flush_deferred_build.  PGP_DEFER_BUILD=0: queued by pgp_set_scene itself.
  std::function<int()> deferred_build;
  uint32_t* h_build_counts = nullptr;   // pinned {candidates, occupied cells} of the pending build
  pgp::DevBuf d_build_scan;             // the build's own scan scratch (d_scan_tmp serves the congruent sets meanwhile)

  // model
  int nQ = 0;
  bool has_model_normals = false;
  pgp::DevBuf d_Q;       // float4 {x,y,z,bits(orig i)} in Morton order [nQ]
  pgp::DevBuf d_Qn;      // float4 {nx,ny,nz,0}          in Morton order [nQ]
  pgp::DevBuf d_Qpos;    // int: Morton position of ORIGINAL model point i [nQ] (Verify's early-out walks in model order)
  bool verify_early_out = false;   // pgp_set_verify_early_out: plain scores as the reference's Verify returns them
  pgp::DevBuf d_eo_ws;             // its workspace: terminate value per hypothesis

  // search model (congruent-set side, sampled_Q_3D_)
  int nQs = 0;
  pgp::DevBuf d_Qs;      // float4 {x,y,z,bits(i)} original order [nQs]
  pgp::DevBuf d_Qs_unit; // float4 unit-cube image (PairCreationFunctor::points) [nQs]
  float cs_gcenter[3] = {0.f, 0.f, 0.f};
  float cs_ratio = 1.f;
  pgp::DevBuf d_cs_cnt, d_cs_entries, d_cs_keys;   // congruent-set workspaces
  pgp::DevBuf d_cs_pairs, d_cs_out;                // host-API staging
  pgp::DevBuf d_ids;     // staged int4 base / quad ids (host API)
  pgp::DevBuf d_rig;     // staged rigid-fit outputs (host API)

  // base selection (base_select.hip): device hash set of the model's pair-feature keys + CSR pair lists
  bool ppf_ready = false;
  pgp::DevBuf d_ppf_keys, d_ppf_val, d_ppf_off, d_ppf_pairs;
  uint32_t ppf_mask = 0;
  int ppf_shift = 0, ppf_n_keys = 0;
  long long ppf_n_pairs = 0;
  std::vector<uint32_t> ppf_off_host;   // pair-list offsets per key, host copy
  float ppf_tpos[9] = {0}, ppf_tneg[9] = {0};   // ratio thresholds of the 10-degree angle bins (host atan2f)
  pgp::DevBuf d_csb, d_csb_picks;       // batched congruent sets: bases | cones | per-base starts; staged picks
  int csb_nb = 0;                       // bases of the last pgp_find_congruent_batch (its keys are still resident);
                                        // 0 as soon as d_cs_keys / d_ppf_pairs / the search model are rewritten
  int csb_fit_m = 0;                    // fits of the last pgp_congruent_batch_fit_score, resident in d_rig
  uint32_t csb_total = 0;
  uint32_t csb_keys_off = 0;            // the sorted keys start this many keys into d_cs_keys (the capacity of the unsorted ones)
  uint32_t csb_cap_hint = 0;            // twice the last batch's total: the next batch's key capacity
  std::vector<uint32_t> csb_starts;     // per-base starts in the sorted keys (nb + 1), host copy: picks are checked here
  pgp::DevBuf d_prob_cdf;               // double prefix sums of the scene weights (first draw)
  bool prob_cdf_valid = false;
  pgp::DevBuf d_sel_ws;                 // base-selection workspace / staging
  int sel_begun_A = 0;                  // attempts of a selection that was queued and not collected yet (pgp_select_bases_rows_begin)
  void* h_sel_pin = nullptr;            // its variates' pinned image
  size_t h_sel_cap = 0;

  pgp::DevBuf d_pre_ws, d_vg_ws, d_pre_io;   // preprocess.hip: bbox partials, voxel-grid workspace, host-API staging
  pgp::DevBuf d_mls_ws;                      // mls.hip: sort keys, sorted cloud, per-point results
  bool hd_attr_set = false;

  pgp::DevBuf d_depth;   // depth-cost staging: observed | rendered[n] | counts
  pgp::DevBuf d_render_ws, d_render_io;   // render.hip: projected vertices [n][n_vert]; host-API staging
  pgp::DevBuf d_bp;      // back-projection staging: image | mask | counters | scan scratch | xyz
  pgp::DevBuf d_cl_keys, d_cl_ws, d_cl_io;   // pose clustering: sort keys, pose tables + bit matrix, host-API staging

  // ICP (host API staging + per-pose correspondence workspace)
  pgp::DevBuf d_icp_src, d_icp_tgt, d_icp_tgt_n, d_icp_T, d_icp_out, d_icp_ws, d_icp_grid;
  pgp::DevBuf d_top_ws;      // select.hip: sort keys / indices / rocprim scratch of pgp_select_top_device
  pgp::DevBuf d_icp_x;       // clustered ICP: the workgroups' shares of (d2, correspondence), ping-pong + arrival counters
  int n_cus = 0;             // compute units of the device if it takes cooperative launches, else 0
  bool icp_attr_set = false;   // dynamic-LDS limit of the ICP kernels raised on this device
  // the exact index of the ICP target (icp.hip build_nn_index) stays valid across calls while the caller
  // vouches for the target: token != 0 and the same (pointer, size, token) = the same points
  bool icp_idx_valid = false;
  unsigned long long icp_idx_token = 0;
  const void* icp_idx_tgt = nullptr;
  int icp_idx_ntgt = 0, icp_idx_nq = 0;
  size_t icp_idx_vic_off = 0;   // byte offset of the vicinity graph inside d_icp_grid (0: none)
  alignas(8) unsigned char icp_idx_geom[96] = {0};
  // the uniform grid of the capped scene-sized search (icp.hip use_grid), kept across calls by the same rule
  // clustered ICP launches: the words of d_icp_x that hold the meeting counters of `icp_x_n` poses over `icp_x_need` meeting
  // records were the counters of the last such launch (which leaves them at zero)
  bool icp_x_clean = false;
  int icp_x_n = 0;
  size_t icp_x_need = 0;
  bool icp_grid_valid = false;
  unsigned long long icp_grid_token = 0;
  const void* icp_grid_tgt = nullptr;
  int icp_grid_ntgt = 0;
  float icp_grid_cap = 0.f;
  float icp_grid_geom[4] = {0.f, 0.f, 0.f, 0.f};
  int icp_grid_n[3] = {0, 0, 0};
  unsigned long long icp_user_token = 0;                    // pgp_icp_target_token: device-pointer calls
  unsigned long long icp_host_token = 0, icp_host_ntoken = 0;   // hash of the last uploaded host target / normals
  int icp_host_ntgt = 0;

  // scoring workspace
  int cap_h = 0;
  pgp::DevBuf d_T;        // staged transforms (host API)            [cap_h*16] float
  pgp::DevBuf d_partial;  // per (tile, hypothesis) partials          [n_tiles*cap_h] int2/float
  pgp::DevBuf d_acc;      // fused finalisation (lcp_score.hip FuseArgs): the near word, then one ticket per chunk, zero between launches
  pgp::DevBuf d_scores;   // [cap_h] float
  pgp::DevBuf d_counts;   // [cap_h] int
  pgp::DevBuf d_best;     // 2 x uint64 packed argmax + {index, score bits}
  pgp::DevBuf d_hits;     // [nQ] int (pgp_registered)
  pgp::DevBuf d_seq;      // [nQ rounded up to 4] float: registered weights in original model order
                          // (finalize_scores' exact re-score of near-best hypotheses)

  // pinned host staging for the host-pointer scoring call (transforms in, scores | counts | best out)
  void* h_pin = nullptr;
  size_t h_pin_cap = 0;
  // pinned staging of pgp_set_scene_weights alone (weights + their double prefix sums go up without a synchronisation;
  // ev_w: the copies out of it have finished -- waited for before the next call overwrites it)
  void* h_w_pin = nullptr;
  size_t h_w_cap = 0;
  hipEvent_t ev_w = nullptr;
  // the same for pgp_set_scene's points and normals (scenes up to 4 MB of them); the side-stream index build waits for
  // ev_s on its own stream, the host for nothing
  void* h_s_pin = nullptr;
  size_t h_s_cap = 0;
  hipEvent_t ev_s = nullptr;
  bool scene_upload_pending = false;   // ev_s has been recorded behind this scene's uploads
  // pinned landing area of small results on their way to the caller's (pageable) memory: pgp::HostOut
  void* h_out = nullptr;
  size_t h_out_cap = 0, h_out_used = 0;
  bool h_out_failed = false;
  pgp::DevBuf d_out;      // scores | counts | best of one host-pointer call, contiguous: ONE copy back

  // tuning knobs (env PGP_UNROLL / PGP_HPB at pgp_create; defaults chosen by measurement)
  int unroll = 0;      // <= 0: wave-flattened candidate phase (default); > 0: per-lane walk
  int hpb_override = 0;
  bool refine_best = true;   // PGP_REFINE=0 switches the exact near-tie re-score off (timing A/B only)
  bool exact_ties = false;      // pgp_set_exact_ties: exact distance ties go to the scene point the reference's kd-tree returns
  bool kd_valid = false;        // d_kd_* hold the reference's tree over the current scene (kd_ties.hip)
  int kd_n_nodes = 0;
  pgp::DevBuf d_kd_nodes;       // int4 per node: inner {bits(split), first child, dim, 0}, leaf {start, size, 0, 1}
  pgp::DevBuf d_kd_pts;         // float4 {x, y, z, bits(original index)} in the tree's order
  bool exact_records = false;   // pgp_set_exact_records: weighted scores exact at every running-best decision
  pgp::DevBuf d_rec_ws;         // its workspace: near-record list | count | weights [nQ][kRecordCap]

  // optional per-kernel timing (pgp_set_kernel_timing)
  int timing = 0;               // 0 off, N >= 1: every Nth scoring launch carries start/stop events
  unsigned timing_seq = 0;
  std::vector<hipEvent_t> ev;   // pairs: [2k] start, [2k+1] stop
  size_t ev_used = 0;           // events recorded since the last reset

  // normal gate (base.cc:1756-1758) as thresholds on the dot product, see gate_thresholds()
  float gate_deg_cached = -1.f;
  float gate_lo = 2.f, gate_hi = -2.f;
};

namespace pgp {

int flush_deferred_build(pgp_ctx* ctx);   // grid_index.hip
// pgp_api.hip: a small input from a PINNED host image to the device by a kernel on `st` (no copy-engine hand-over)
int stage_to_device(hipStream_t st, void* d_dst, const void* h_pinned, size_t bytes);

// Small results on their way to the caller's memory.  A device-to-host copy into PAGEABLE memory keeps the host 12 us
// longer than one into pinned memory, whatever its size (tools/copy_cost.hip on the bench box: 24.4 against 12.2 us for
// 64 B .. 4 KB, copy + stream synchronisation) -- and a drop-in call makes five of them.  A HostOut lands the copies in a
// pinned area of the context and hands them on after ONE synchronisation:
//     HostOut out(ctx, st);  out.to(h_dst, d_src, bytes); ...  rc = out.sync();
// fetch() instead returns where the bytes will be (valid after sync(), until the HostOut goes out of scope).  Stack
// discipline (a callee's HostOut sits above its caller's); results beyond the area's room, or all of them when the area
// cannot be allocated, are copied straight to their destination as before.  Leaving the scope without sync() drops the
// pending deliveries (the error paths).
// (pgp_api.hip) One kernel that copies up to 8 small device arrays into PINNED host memory and writes a completion word
// behind them, and the host's wait for that word: see HostOut::sync.
struct PubItem {
  const void* d_src;
  void* h_dst;      // pinned
  size_t bytes;     // a multiple of 4, both ends 4-byte aligned
};
constexpr int kPubMaxItems = 8;
constexpr size_t kPubMaxBytes = 1u << 20;   // (one workgroup per 32 KB, up to 32: a megabyte leaves at PCIe speed)
bool publish_usable(const PubItem* items, int n);
int publish_and_wait(pgp_ctx* ctx, hipStream_t st, const PubItem* items, int n);

struct HostOut {
  static constexpr size_t kArea = 1u << 20, kMaxItems = 8;
  pgp_ctx* ctx;
  hipStream_t st;
  size_t mark;
  struct Item {
    void* dst;          // NULL: fetch() -- the caller reads the landing area itself
    size_t off, n;
    const void* d_src;
    unsigned char* spill;   // fetch() without room in the area: a pageable landing buffer
  } items[kMaxItems + 4];
  int n_items = 0;
  bool direct = false;      // a copy straight into the caller's memory has been queued: the stream itself must be waited for
  std::vector<std::vector<unsigned char>> spill;   // fetch() without room in the area: pageable landing buffers of its own

  HostOut(pgp_ctx* c, hipStream_t s) : ctx(c), st(s), mark(c->h_out_used) {}
  ~HostOut() { ctx->h_out_used = mark; }
  HostOut(const HostOut&) = delete;
  HostOut& operator=(const HostOut&) = delete;

  unsigned char* room(size_t bytes) {
    if (!ctx->h_out && !ctx->h_out_failed) {
      if (hipHostMalloc(&ctx->h_out, kArea, hipHostMallocDefault) == hipSuccess) ctx->h_out_cap = kArea;
      else {
        (void)hipGetLastError();
        ctx->h_out = nullptr;
        ctx->h_out_failed = true;
      }
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!ctx->h_out || ctx->h_out_used + need > ctx->h_out_cap) return nullptr;
    unsigned char* p = static_cast<unsigned char*>(ctx->h_out) + ctx->h_out_used;
    ctx->h_out_used += need;
    return p;
  }
  // The copies are QUEUED BY sync(), not here: every caller's pattern is kernels -> to() ... -> sync() with nothing
  // launched in between, and sync() can then bring all of them home with ONE kernel (below).
  int to(void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return PGP_OK;
    unsigned char* p = n_items < (int)kMaxItems ? room(bytes) : nullptr;
    if (!p) {
      PGP_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
      direct = true;
      return PGP_OK;
    }
    items[n_items++] = Item{dst, (size_t)(p - static_cast<unsigned char*>(ctx->h_out)), bytes, d_src, nullptr};
    return PGP_OK;
  }
  int fetch(const unsigned char** where, const void* d_src, size_t bytes) {
    unsigned char* p = n_items < (int)kMaxItems ? room(bytes) : nullptr;
    if (!p) {
      spill.emplace_back(bytes ? bytes : 1);
      p = spill.back().data();
      *where = p;
      if (bytes) {
        PGP_HIP(hipMemcpyAsync(p, d_src, bytes, hipMemcpyDeviceToHost, st));
        direct = true;
      }
      return PGP_OK;
    }
    *where = p;
    if (bytes) items[n_items++] = Item{nullptr, (size_t)(p - static_cast<unsigned char*>(ctx->h_out)), bytes, d_src, nullptr};
    return PGP_OK;
  }
  // Results home after ONE wait.  Small results (<= 96 KB in all, word-aligned) are copied into the pinned area by ONE kernel
  // on the stream, which writes a completion word behind them that the host polls (publish_and_wait): a copy-engine transfer
  // behind a kernel and the stream's completion signal cost 10-15 us per wait, a kernel behind a kernel ~2.5 us
  // (profiles/r06_ab/stage_kernel.log).  Anything else: one device-to-host copy per item and a stream synchronisation.
  int sync() {
    // (the host is about to wait: what it has put off issuing -- the scene's index build -- goes to its stream first and
    //  runs beside whatever this wait is for)
    if (ctx->deferred_build) {
      const int rc = flush_deferred_build(ctx);
      if (rc != PGP_OK) return rc;
    }
    unsigned char* area = static_cast<unsigned char*>(ctx->h_out);
    PubItem pub[kMaxItems + 4];
    for (int k = 0; k < n_items; ++k) pub[k] = PubItem{items[k].d_src, area + items[k].off, items[k].n};
    if (!direct && n_items > 0 && publish_usable(pub, n_items)) {
      const int rc = publish_and_wait(ctx, st, pub, n_items);
      if (rc != PGP_OK) return rc;
    } else {
      for (int k = 0; k < n_items; ++k)
        PGP_HIP(hipMemcpyAsync(area + items[k].off, items[k].d_src, items[k].n, hipMemcpyDeviceToHost, st));
      PGP_HIP(hipStreamSynchronize(st));
    }
    for (int k = 0; k < n_items; ++k)
      if (items[k].dst) std::memcpy(items[k].dst, area + items[k].off, items[k].n);
    n_items = 0;
    direct = false;
    return PGP_OK;
  }
};

// cell (x, y, z) -> position in the blocked numbering (word index * 32 + bit)
__host__ __device__ inline uint32_t grid_word(const GridDesc& g, int x, int y, int z) {
  return ((uint32_t)(z >> 1) * (uint32_t)g.nby + (uint32_t)(y >> 2)) * (uint32_t)g.nbx + (uint32_t)(x >> 2);
}
__host__ __device__ inline uint32_t grid_bit(int x, int y, int z) {
  return (uint32_t)(((z & 1) << 4) | ((y & 3) << 2) | (x & 3));
}

// grid_index.hip
int build_index(pgp_ctx* ctx, const float* h_xyz, float delta);
// kd_ties.hip
int build_kd_ties(pgp_ctx* ctx, const float* h_xyz, int n);
int build_index_bbox(pgp_ctx* ctx, const float mn[3], const float mx[3], float delta);
int device_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, hipStream_t st);

// congruent.hip
void unit_cube_image(const float* xyz, int n, float gcenter[3], float* ratio, std::vector<float4>* unit);
int launch_extract_pairs(pgp_ctx* ctx, float pair_distance, float eps, int* d_pairs, int cap,
                         int* n_pairs_host, hipStream_t st);
int launch_find_congruent(pgp_ctx* ctx, const float base[12], float inv1, float inv2, float threshold,
                          const int* d_Pp, int nP, const int* d_Qp, int nQ, int* d_quads, int cap,
                          int* n_quads_host, hipStream_t st);

int launch_find_congruent_4pcs(pgp_ctx* ctx, float inv1, float inv2, float threshold, const int* d_Pp, int nP,
                               const int* d_Qp, int nQ, int* d_quads, int cap, int* n_quads_host, hipStream_t st);
int launch_find_congruent_batch(pgp_ctx* ctx, const int* h_base_ids, const float* h_base_xyz, const float* h_inv,
                                const int* h_rows /* nullable */, int nb, float threshold, int* h_n_quads, hipStream_t st);
int launch_congruent_batch_gather(pgp_ctx* ctx, const int* h_picks, int m, int4* d_quads, hipStream_t st,
                                  const int2* d_picks_there = nullptr);
const uint32_t* congruent_batch_starts_device(pgp_ctx* ctx);

// lcp_score.hip
int tiles_for(int nQ);
// host (nullable): the caller wants the results in HOST memory as well, written by finalize_scores itself, with a completion
// word behind them (lcp_score.hip HostPub).  `published` says on return whether the launch will do so (an empty batch does
// not).  best[2] = 1: a weighted near-tie was settled on the device -- copy d_scores / d_counts / d_best back for that call.
struct ScoreHostOut {
  float* scores;
  int* counts;
  int* best;            // {index, score bits, settled-on-device, spare}
  unsigned int* flag;
  unsigned int flag_value;
  bool published;
};
int launch_score(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg,
                 float* d_scores, int* d_counts, int* d_best, hipStream_t stream, ScoreHostOut* host = nullptr);
// seq_ws (nullable): the exact re-score's workspace, the size of ctx->d_seq -- a caller that settles on ANOTHER stream than the
// one the context's scoring launches run on (the device group's streaming form) brings its own: finalize_scores uses ctx->d_seq
int launch_settle_best(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                       int* d_best, hipStream_t stream, float* seq_ws = nullptr);
// the launches that follow on `stream` read the scene's index: orders them behind a build still running on
// ctx->build_stream (a no-op otherwise); finish_index() waits for it on the host and takes over its counts
int await_index(pgp_ctx* ctx, hipStream_t stream);
int finish_index(pgp_ctx* ctx);
int launch_settle_records(pgp_ctx* ctx, const float* d_T, int n_h, int mode, float gate_deg, float* d_scores,
                          hipStream_t stream);
int launch_verify_early_out(pgp_ctx* ctx, const float* d_T, int n_h, float* d_scores, int* d_counts, hipStream_t stream);
int records_workspace_bytes(int nQ);
int launch_registered_model(pgp_ctx* ctx, const float* d_T16, const float4* d_q, const float4* d_qn, int n,
                            float gate_deg, int* d_hits, hipStream_t stream);
int launch_registered(pgp_ctx* ctx, const float* d_T16, int mode, float gate_deg, int* d_hits,
                      hipStream_t stream);
void gate_thresholds(float gate_deg, float* c_aligned_min, float* c_anti_max);
int launch_count_neighbours(pgp_ctx* ctx, float radius, int* d_counts, hipStream_t stream);

// icp.hip
// select.hip
int launch_select_top(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, int k, int invert, float* d_T_out,
                      int* d_idx_out, int* d_n_out, hipStream_t st);
// one (segment, target) pair of a multi-target ICP launch (pgp_icp_refine_multi_device)
struct IcpJob {
  pgp_ctx* ctx;            // holds the target's index between calls
  const float4* d_src;
  int n_src;
  const float4* d_tgt;
  int n_tgt;
  float* d_T;
  int n;
  float* d_energy;         // nullable
  int* d_iters;            // nullable
  unsigned long long token;
};
int launch_icp_multi(const IcpJob* jobs, int n_jobs, const pgp_icp_options* prm, hipStream_t stream);
int launch_icp(pgp_ctx* ctx, const float4* d_src, int n_src, const float4* d_tgt, const float4* d_tgt_n, int n_tgt,
               float* d_T, int n, const pgp_icp_options* prm, float* d_energy, int* d_iters, hipStream_t stream,
               unsigned long long tgt_token = 0);

// pgp_api.hip: a host-pointer ICP job staged in a context's buffers (pgp_icp_refine_ex; the device group's pose shards)
struct IcpHostStage {
  const float4* d_src;
  const float4* d_tgt;
  float* d_T;
  float* d_energy;
  int* d_iters;
  unsigned long long token;          // hash of the target's coordinates: the key of its resident index
  size_t off_T, off_e, off_i, total; // byte offsets inside d_icp_src and its pinned image
};
int icp_host_stage(pgp_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt, const float* T, int n,
                   hipStream_t st, IcpHostStage* out);
int icp_host_collect_enqueue(pgp_ctx* ctx, const IcpHostStage& g, hipStream_t st);
void icp_host_collect(pgp_ctx* ctx, const IcpHostStage& g, int n, float* T, float* energy, int* iters);
pgp_icp_options icp_options_of(const pgp_icp_params* p);
void icp_scene_form_off(bool off);   // icp.hip: this THREAD's next launch_icp calls take the host-driven scene-sized form

// base_select.hip
int set_ppf_map(pgp_ctx* ctx, const int* keys, const int* counts, const int* pairs, int n_keys);
int ppf_thresholds(float tpos[9], float tneg[9]);
int launch_select_bases(pgp_ctx* ctx, const double* h_u, int n_attempts, int* h_ids, float* h_inv, int* h_status,
                        int* h_rows /* nullable */, hipStream_t st, int phase = 0);
int launch_ppf_features(pgp_ctx* ctx, const int* h_pairs, int m, int* h_f, int* h_row, hipStream_t st);
int launch_stage_weights(pgp_ctx* ctx, int stage, int b1, int b2, int b3, float* h_cur, float* h_sum, int* h_present,
                         hipStream_t st);
int launch_base_invariants(pgp_ctx* ctx, int* h_ids, int m, float* h_inv, int* h_ok, hipStream_t st);

// preprocess.hip
int device_bbox(pgp_ctx* ctx, const float* d_pts, int n, int stride, float mn[3], float mx[3], hipStream_t st);
int launch_scene_weights(pgp_ctx* ctx, const float* d_w, int n, hipStream_t st);
int launch_explained_points(pgp_ctx* ctx, const float* d_seg, int n, const float* d_model, const int* d_model_off,
                            const float* d_T, int n_obj, float radius, unsigned int* d_explained, hipStream_t st);
int launch_voxel_grid(pgp_ctx* ctx, const float* d_xyz, int n, float leaf, float* d_out, int cap, int* n_out,
                      hipStream_t st);
int launch_pose_hausdorff(pgp_ctx* ctx, const float4* d_hull, int n_hull, const float* d_T, int n_poses,
                          const int2* d_pairs, int m, float* d_max, float* d_sum, hipStream_t st);
int set_scene_device(pgp_ctx* ctx, const float* d_xyz, const float* d_nrm, const float* d_w, int n, float delta,
                     hipStream_t st);

// mls.hip
int launch_mls(pgp_ctx* ctx, const float* d_xyz, int n, float radius, float* d_out_xyz, float* d_out_nrm,
               float* d_out_curv, int* d_out_index, int cap, int* n_out, hipStream_t st);

// depth_cost.hip
int launch_depth_cost(pgp_ctx* ctx, const float* d_obs, const float* d_ren, int n, int n_pix, float thr,
                      int* d_counts, hipStream_t stream);

// render.hip
int launch_render_depth(pgp_ctx* ctx, const float* d_verts, int stride, int n_vert, const int* d_tris, int n_tri,
                        const float* d_T, int n, const pgp_camera* cam, const float* d_parent, size_t parent_stride,
                        float* d_depth, hipStream_t st);
int launch_cost_scores(const int* d_counts, int n, float* d_scores, hipStream_t st);

// cluster.hip
int launch_cluster(pgp_ctx* ctx, const float* d_T, const float* d_scores, int n, float best_score,
                   const float sym[3], const pgp_cluster_params* prm, int* d_rep, int* d_assign,
                   int* h_m, int* h_n_rep, hipStream_t st);

int launch_pose_error(pgp_ctx* ctx, const float* d_test, const float* d_gt, int n, const float sym[3],
                      float* d_rot, float* d_trans, hipStream_t st);

// backproject.hip
int launch_backproject(pgp_ctx* ctx, const void* d_img, bool raw16, const unsigned char* d_mask, int rows,
                       int cols, const float K[9], double z_min, double z_max, uint32_t* d_ctr,
                       uint32_t* d_scan_tmp, float* d_xyz, int cap, int* n_host, hipStream_t st);

// rigid_fit.hip
int launch_rigid(pgp_ctx* ctx, const int* d_base_ids, const int* d_quad_ids, int n, const float cP[3],
                 const float cQ[3], float* d_T, double* d_pose, int* d_status, float* d_rms,
                 hipStream_t stream);

}  // namespace pgp
