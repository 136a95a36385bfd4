// csrc/grid_index.hip -- device spatial index over the scene cloud.
//
// Replaces Match4PCSBase::initKdTree (S4/algorithms/match4pcsBase.cc:1046-1056) and the kd-tree
// it builds (S4/accelerators/kdtree.h:355-370,560-641).  The reference answers "nearest scene
// point within delta of x" by a stack-driven kd-tree walk (kdtree.h:394-459); a data-dependent
// tree walk per lane is the wrong shape for 64-wide wavefronts, so the device index is a uniform
// grid with DILATED per-cell candidate lists:
//
//   cell(x)      = floor((x - origin) * inv_h), h = 0.85 delta by default (choose_grid)
//   cand(c)      = every scene point within `reach` = delta + margin of the box of cell c
//   words        = per block of 4 x 4 x 2 cells (cells are numbered block-major, pgp_internal.h
//                  grid_word / grid_bit): {occupancy bits, rank base}   (2.9 MB at C2, L2-resident)
//   occ_run      = {start, count} per OCCUPIED cell (rank = base + popcount of lower bits) into
//                  one float4 array {x,y,z,bits(id)} of all candidate lists (3.3 MB at C2)
//
// A query is then: one 8-byte word, (if the bit is set) one 8-byte offset pair, one contiguous
// float4 run -- no neighbour-cell gather, no tree; ~78 % of C2 queries end at the word.  Exactness: the scoring kernel applies the reference's float
// test d2 <= delta^2 (kdtree.h:423-424) to every candidate; the dilation margin (see
// choose_grid) covers the float rounding of cell(x), so cand(cell(x)) is a superset of the
// scene points that pass the test for x.  The result equals an exhaustive scan.
//
// SPARSE form.  The dense block array and its build scratch grow with the bounding-box VOLUME (8 bytes of
// scratch per cell).  A scene whose box is mostly empty -- a whole room at delta = 5 mm -- takes the sparse
// form instead: the same 4 x 4 x 2 blocks, but only those that hold a candidate, in an open-addressing table
// (uint4 {key, bits, base, slot}, load factor <= 1/8, multiplicative hash, linear probing).  A query reads
// one 16-byte entry where the dense form reads one 8-byte word; scratch and table scale with the occupied
// blocks, the cell edge stays 0.85 delta up to 16 384 cells per axis (70 m at delta = 5 mm).
//
// HBM layout: 28 copies of each point on average (h = 0.85 delta, two dilation rings) = 22.6 MB at
// |P| = 50 k; the index trades capacity (288 GB) for one-run locality.

#include <mutex>

#include "pgp_internal.h"
#include "host_worker.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace pgp {

namespace {

constexpr int kMaxDim = 1024;                  // cells per axis, dense form (the flat scoring kernel's bit fields)
constexpr long long kMaxCells = 1LL << 26;     // dense form: 67 M cells (2 x 256 MiB of build scratch); beyond, sparse
constexpr int kMaxDimSparse = 16384;           // cells per axis, sparse form

__device__ __forceinline__ float box_dist2(float px, float lo, float h) {
  // squared distance from coordinate px to the interval [lo, lo+h]
  float a = lo - px;
  float b = px - (lo + h);
  float d = fmaxf(fmaxf(a, b), 0.f);
  return d * d;
}

// Every cell whose box is within `reach` of point p (the (2r+1)^3 cells around its own): f(x, y, z).
template <class F>
__device__ __forceinline__ void for_cells_in_reach(const GridDesc& g, int r, float4 p, F f) {
  float fx = (p.x - g.ox) * g.inv_h, fy = (p.y - g.oy) * g.inv_h, fz = (p.z - g.oz) * g.inv_h;
  if (!(fx >= 0.f && fx < (float)g.nx && fy >= 0.f && fy < (float)g.ny && fz >= 0.f && fz < (float)g.nz))
    return;  // NaN / inf points can never be inliers (d2 <= eps is false for NaN)
  int cx = (int)fx, cy = (int)fy, cz = (int)fz;
  float reach2 = g.reach * g.reach;
  for (int dz = -r; dz <= r; ++dz) {
    int z = cz + dz;
    if (z < 0 || z >= g.nz) continue;
    float ez = box_dist2(p.z, g.oz + (float)z * g.h, g.h);
    for (int dy = -r; dy <= r; ++dy) {
      int y = cy + dy;
      if (y < 0 || y >= g.ny) continue;
      float ey = box_dist2(p.y, g.oy + (float)y * g.h, g.h);
      for (int dx = -r; dx <= r; ++dx) {
        int x = cx + dx;
        if (x < 0 || x >= g.nx) continue;
        float ex = box_dist2(p.x, g.ox + (float)x * g.h, g.h);
        if (ex + ey + ez > reach2) continue;
        f(x, y, z);
      }
    }
  }
}

// slot of an existing block of the sparse table (build time: every block asked for was inserted)
__device__ __forceinline__ uint32_t block_slot(const GridDesc& g, const uint4* __restrict__ tab, int x, int y, int z) {
  const uint32_t key = block_key(g, (uint32_t)x >> 2, (uint32_t)y >> 2, (uint32_t)z >> 1);
  uint32_t i = block_hash(g, key);
  for (uint32_t probe = 0; probe <= g.tab_mask; ++probe) {
    const uint4 e = tab[i];
    if (e.x == key) return e.w;
    i = (i + 1u) & g.tab_mask;
  }
  return 0u;
}

// One thread per scene point: for each cell within `reach`, either count it (FILL=false) or append the
// point to the cell's list.  SPARSE: the cell's number is (slot of its block) * 32 + bit.
template <bool FILL, bool SPARSE>
__global__ __launch_bounds__(256) void scatter_points(GridDesc g, int r, const float4* __restrict__ P,
                                                      int nP, uint32_t* __restrict__ cell_ctr,
                                                      const uint32_t* __restrict__ cell_start,
                                                      float4* __restrict__ cand, const uint4* __restrict__ tab) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nP) return;
  const float4 p = P[i];
  for_cells_in_reach(g, r, p, [&](int x, int y, int z) {
    const size_t w = SPARSE ? (size_t)block_slot(g, tab, x, y, z) : (size_t)grid_word(g, x, y, z);
    const size_t c = w * 32 + grid_bit(x, y, z);
    uint32_t slot = atomicAdd(&cell_ctr[c], 1u);
    if (FILL) cand[cell_start[c] + slot] = p;
  });
}

// The same for a SMALL scene (the drop-in's segments: a few thousand points, 8 workgroups of the kernel above, each thread
// 27 .. 125 returning atomics one behind the other): kScatterLanes threads per point share the (2r+1)^3 cells around it.
// Which slot of a cell a point gets is first come, first served in both kernels.
constexpr int kScatterLanes = 32;
template <bool FILL>
__global__ __launch_bounds__(256) void scatter_points_lanes(GridDesc g, int r, const float4* __restrict__ P, int nP,
                                                            uint32_t* __restrict__ cell_ctr, const uint32_t* __restrict__ cell_start,
                                                            float4* __restrict__ cand) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(t / kScatterLanes), sub = (int)(t % kScatterLanes);
  if (i >= nP) return;
  const float4 p = P[i];
  const float fx = (p.x - g.ox) * g.inv_h, fy = (p.y - g.oy) * g.inv_h, fz = (p.z - g.oz) * g.inv_h;
  if (!(fx >= 0.f && fx < (float)g.nx && fy >= 0.f && fy < (float)g.ny && fz >= 0.f && fz < (float)g.nz)) return;   // (for_cells_in_reach)
  const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
  const float reach2 = g.reach * g.reach;
  const int w = 2 * r + 1, n_off = w * w * w;
  for (int k = sub; k < n_off; k += kScatterLanes) {
    const int dz = k / (w * w) - r, dy = (k / w) % w - r, dx = k % w - r;
    const int x = cx + dx, y = cy + dy, z = cz + dz;
    if (x < 0 || x >= g.nx || y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
    const float ez = box_dist2(p.z, g.oz + (float)z * g.h, g.h), ey = box_dist2(p.y, g.oy + (float)y * g.h, g.h),
                ex = box_dist2(p.x, g.ox + (float)x * g.h, g.h);
    if (ex + ey + ez > reach2) continue;
    const size_t c = (size_t)grid_word(g, x, y, z) * 32 + grid_bit(x, y, z);
    // counting pass: up from zero; filling pass: DOWN from the count the first pass left (no fill of the counters between
    // the passes: 13 us for the 26 MB of a sparse segment's grid)
    if (FILL) cand[cell_start[c] + (atomicSub(&cell_ctr[c], 1u) - 1u)] = p;
    else atomicAdd(&cell_ctr[c], 1u);
  }
}

// Sparse form, pass 1: the distinct blocks that hold a cell within reach of some point, counted through a
// scratch table of bare keys (capacity = a bound on the blocks all points can touch, load <= 1/2).
__global__ __launch_bounds__(256) void blocks_count(GridDesc g, int r, const float4* __restrict__ P, int nP,
                                                    uint32_t* __restrict__ keys, uint32_t mask, int shift,
                                                    uint32_t* __restrict__ n_blocks) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nP) return;
  uint32_t last = kBlockEmpty, fresh = 0;
  for_cells_in_reach(g, r, P[i], [&](int x, int y, int z) {
    const uint32_t key = block_key(g, (uint32_t)x >> 2, (uint32_t)y >> 2, (uint32_t)z >> 1);
    if (key == last) return;
    last = key;
    uint32_t j = (key * 2654435761u) >> shift;
    for (;;) {
      uint32_t k = keys[j];
      if (k == kBlockEmpty) k = atomicCAS(&keys[j], kBlockEmpty, key);
      if (k == kBlockEmpty) { ++fresh; break; }
      if (k == key) break;
      j = (j + 1u) & mask;
    }
  });
  if (fresh) atomicAdd(n_blocks, fresh);
}

// pass 2: the same blocks into the final table (load <= 1/8); the inserting thread numbers the block
__global__ __launch_bounds__(256) void blocks_insert(GridDesc g, int r, const float4* __restrict__ P, int nP,
                                                     uint4* __restrict__ tab, uint32_t* __restrict__ n_blocks) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nP) return;
  uint32_t last = kBlockEmpty;
  uint32_t* tab32 = reinterpret_cast<uint32_t*>(tab);
  for_cells_in_reach(g, r, P[i], [&](int x, int y, int z) {
    const uint32_t key = block_key(g, (uint32_t)x >> 2, (uint32_t)y >> 2, (uint32_t)z >> 1);
    if (key == last) return;
    last = key;
    uint32_t j = block_hash(g, key);
    for (;;) {
      uint32_t k = tab32[4 * (size_t)j];
      if (k == kBlockEmpty) k = atomicCAS(&tab32[4 * (size_t)j], kBlockEmpty, key);
      if (k == kBlockEmpty) { tab32[4 * (size_t)j + 3] = atomicAdd(n_blocks, 1u); break; }
      if (k == key) break;
      j = (j + 1u) & g.tab_mask;
    }
  });
}

// last pass: {occupancy bits, rank base} of each block from the per-slot words into its table entry
__global__ __launch_bounds__(256) void blocks_finish(uint4* __restrict__ tab, uint32_t n_entries,
                                                     const uint2* __restrict__ words) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_entries) return;
  uint4 e = tab[i];
  if (e.x == kBlockEmpty) return;
  const uint2 w = words[e.w];
  e.y = w.x;
  e.z = w.y;
  tab[i] = e;
}

// ---- exclusive scan over uint32 (3 passes) ----
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

// A thread's kScanItems consecutive words: as two 16-byte accesses where the arrays allow it (VEC: both 16-byte aligned) --
// eight 4-byte accesses at a 32-byte stride across the lanes made the scan of a sparse scene's 6.5 M cell counts 48 + 40 us
// of a 190 us index build, a quarter of the memory's rate.
template <bool VEC>
__device__ __forceinline__ void scan_load(const uint32_t* __restrict__ in, size_t base, size_t n, uint32_t v[kScanItems]) {
  static_assert(kScanItems == 8, "two uint4 per thread");
  if (VEC && base + kScanItems <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(in + base);
    const uint4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
  }
}

// pass 1: the tiles' totals (reads only)
template <bool VEC>
__global__ __launch_bounds__(kScanThreads) void scan_tile_totals(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  scan_load<VEC>(in, base, n, v);
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sum += v[k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) t += s_wave[w];
    tile_sums[blockIdx.x] = t;
  }
}

// pass 3 (after scan_tile_sums has turned the totals into the tiles' offsets): the exclusive prefix of every word.
// `in` and `out` may alias (every thread reads its items before it writes them)
template <bool VEC>
__global__ __launch_bounds__(kScanThreads) void scan_tiles(const uint32_t* in, uint32_t* out, size_t n,
                                                           const uint32_t* __restrict__ tile_offsets) {
  __shared__ uint32_t s_wave[kScanThreads / 64];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  scan_load<VEC>(in, base, n, v);
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sum += v[k];
  // inclusive scan of `sum` across the wave, then across the 4 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t wave_off = tile_offsets[blockIdx.x];
  for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
  uint32_t excl = wave_off + incl - sum;
  uint32_t o[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    o[k] = excl;
    excl += v[k];
  }
  if (VEC && base + kScanItems <= n) {
    uint4* q = reinterpret_cast<uint4*>(out + base);
    q[0] = make_uint4(o[0], o[1], o[2], o[3]);
    q[1] = make_uint4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (base + k < n) out[base + k] = o[k];
  }
}

__global__ __launch_bounds__(1024) void scan_tile_sums(uint32_t* __restrict__ tile_sums, int n_tiles) {
  // single block: sequential over chunks of 1024, carry in a shared word
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n_tiles; base += 1024) {
    int i = base + threadIdx.x;
    uint32_t v = (i < n_tiles) ? tile_sums[i] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    uint32_t carry = s_carry;
    if (i < n_tiles) tile_sums[i] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
    __syncthreads();
  }
}

// occupancy bits + popcount per 32-cell word
__global__ __launch_bounds__(256) void make_words(GridDesc g, const uint32_t* __restrict__ cell_start,
                                                  uint2* __restrict__ words,
                                                  uint32_t* __restrict__ word_cnt, size_t n_words) {
  size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > n_words) return;
  if (w == n_words) {  // scan sentinel
    word_cnt[w] = 0;
    return;
  }
  uint32_t bits = 0;
  const size_t c0 = w * 32;  // the word's 32 cells are consecutive in the blocked numbering
  for (int b = 0; b < 32; ++b)
    if (cell_start[c0 + b + 1] > cell_start[c0 + b]) bits |= (1u << b);
  words[w].x = bits;
  word_cnt[w] = __popc(bits);
}

// rank base per word + compact offsets of the occupied cells
__global__ __launch_bounds__(256) void fill_occupied(GridDesc g, const uint32_t* __restrict__ cell_start,
                                                     const uint32_t* __restrict__ word_base,
                                                     uint2* __restrict__ words,
                                                     uint2* __restrict__ occ_run, size_t n_words,
                                                     size_t n_cells) {
  size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint32_t base = word_base[w];
  uint32_t bits = words[w].x;
  words[w].y = base;
  const size_t c0 = w * 32;
  uint32_t k = base;
  while (bits) {
    int b = __ffs(bits) - 1;
    bits &= bits - 1;
    uint32_t st = cell_start[c0 + b];
    occ_run[k++] = make_uint2(st, cell_start[c0 + b + 1] - st);  // {start, count}
  }
}

int ceil_log2(long long n) {
  int b = 0;
  while ((1LL << b) < n) ++b;
  return b;
}

// Choose cell size, origin, dims and form (dense block array or sparse block table) for a bounding box
// and a radius.
int choose_grid(const float mn[3], const float mx[3], float delta, GridDesc* g, int* r_out) {
  float ext[3], maxext = 0.f, maxabs = 0.f;
  for (int k = 0; k < 3; ++k) {
    ext[k] = mx[k] - mn[k];
    maxext = fmaxf(maxext, ext[k]);
    maxabs = fmaxf(maxabs, fmaxf(fabsf(mn[k]), fabsf(mx[k])));
  }
  // Cell edge h = 0.85 delta: the dilation then spans two rings (r = ceil(reach / h) = 2, r is always
  // computed from reach / h, so any h is exact) and a cell's list holds the points within `reach`
  // of a smaller box -- 1.5 M candidates instead of 1.0 M at C2, but 20 % fewer per query.  Measured
  // on the round-1 kernel (tools/tune.py, PGP_CELL_RATIO): 0.75-0.85 -> 85.5 us plain / 116 us
  // weighted, 1.02 -> 88.4 / 116.6, 0.6 -> 88 / 122, 0.45 -> 102 / 133, 1.3 -> 91 / 121.
  // h grows further only when neither form can hold the grid (beyond kMaxDimSparse cells per axis).
  float h = delta * 0.85f;
  if (const char* v = getenv("PGP_CELL_RATIO")) {  // experiment knob: cell edge / delta
    float ratio = (float)atof(v);
    if (ratio > 0.05f && ratio < 64.f) h = delta * ratio;
  }
  int only_form = -1;  // experiment / test knob: PGP_INDEX=dense|sparse
  if (const char* v = getenv("PGP_INDEX")) only_form = (v[0] == 's') ? 1 : (v[0] == 'd') ? 0 : -1;
  for (int iter = 0; iter < 200; ++iter, h *= 1.25f) {
    for (int form = 0; form < 2; ++form) {  // 0: dense block array, 1: sparse block table
      if (only_form >= 0 && form != only_form) continue;
      const int dimcap = form ? kMaxDimSparse : kMaxDim;
      // margin: rounding of cell(x) is <= ~4 ulp of (x-origin)*inv_h (value up to dimcap) plus
      // the ulp of the coordinates themselves; 0.4 % of a cell + 64 ulp(extent) covers it.
      float margin = 0.004f * h + 1e-6f * (float)dimcap * h + 64.f * FLT_EPSILON * (maxabs + maxext);
      float reach = delta * (1.f + 4.f * FLT_EPSILON) + margin;
      int r = (int)ceilf(reach / h);
      if (r < 1) r = 1;
      // origin one dilation ring (+1 cell) outside the box so that every position within reach of
      // a point has a valid cell; a position outside the grid is farther than delta from all of P.
      float pad = (float)(r + 1) * h;
      g->h = h;
      g->inv_h = 1.0f / h;
      g->reach = reach;
      g->sparse = form;
      g->key_sy = g->key_sz = 0;
      g->tab_mask = 0;
      g->tab_shift = 0;
      // origin ON the lattice of pitch 1 / inv_h (GridDesc): cell 0 is lattice cell k0, at most 1.5 cells
      // below mn - pad.  The scoring kernel then finds a cell as round(x * inv_h) - k0 with one fused
      // multiply-add; that and floor((x - origin) * inv_h) differ by the rounding of `origin`, of h vs
      // 1 / inv_h and of the products -- a few ulp of the coordinate, far inside `margin`.
      const double k0d[3] = {floor((double)(mn[0] - pad) * g->inv_h), floor((double)(mn[1] - pad) * g->inv_h),
                             floor((double)(mn[2] - pad) * g->inv_h)};
      g->magic_ok = 1;
      for (int k = 0; k < 3; ++k)
        if (!(fabs(k0d[k]) < (double)((1 << 22) - 4 * dimcap))) g->magic_ok = 0;
      if (g->magic_ok) {
        g->k0x = (int)k0d[0];
        g->k0y = (int)k0d[1];
        g->k0z = (int)k0d[2];
        g->ox = ((float)g->k0x - 0.5f) * h;
        g->oy = ((float)g->k0y - 0.5f) * h;
        g->oz = ((float)g->k0z - 0.5f) * h;
        // h and 1 / inv_h differ in the last place: with a large lattice number the product can land
        // above mn - pad by that much; one more cell of padding restores the invariant
        if (!(g->ox <= mn[0] - pad)) { g->k0x -= 1; g->ox = ((float)g->k0x - 0.5f) * h; }
        if (!(g->oy <= mn[1] - pad)) { g->k0y -= 1; g->oy = ((float)g->k0y - 0.5f) * h; }
        if (!(g->oz <= mn[2] - pad)) { g->k0z -= 1; g->oz = ((float)g->k0z - 0.5f) * h; }
      } else {
        g->k0x = g->k0y = g->k0z = 0;
        g->ox = mn[0] - pad;
        g->oy = mn[1] - pad;
        g->oz = mn[2] - pad;
      }
      double nx = floor((double)(mx[0] - g->ox) / h) + r + 2;
      double ny = floor((double)(mx[1] - g->oy) / h) + r + 2;
      double nz = floor((double)(mx[2] - g->oz) / h) + r + 2;
      if (!(nx <= dimcap && ny <= dimcap && nz <= dimcap)) continue;
      const long long nbx = ((long long)nx + 3) / 4, nby = ((long long)ny + 3) / 4, nbz = ((long long)nz + 1) / 2;
      if (form == 0) {
        if (nbx * nby * nbz * 32 > kMaxCells) continue;
      } else {
        // block key = bx | by << sy | bz << sz in 31 bits (0xFFFFFFFF marks an empty table entry)
        const int wx = ceil_log2(nbx), wy = ceil_log2(nby), wz = ceil_log2(nbz);
        if (wx + wy + wz > 31) continue;
        g->key_sy = wx;
        g->key_sz = wx + wy;
      }
      g->nx = (int)nx;
      g->ny = (int)ny;
      g->nz = (int)nz;
      g->nbx = (int)nbx;
      g->nby = (int)nby;
      g->nbz = (int)nbz;
      *r_out = r;
      return PGP_OK;
    }
  }
  set_error("scene extent %.3g m cannot be gridded for delta %.3g", maxext, delta);
  return PGP_EINVAL;
}

}  // namespace

// exclusive scan of n uint32 (in -> out, may alias), tile sums in `tmp` (>= n/2048 + 2 words): the tiles' totals, their
// scan, then every word's prefix -- the input is read twice and the output written once
int device_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, hipStream_t st) {
  const int n_tiles = (int)((n + kScanTile - 1) / kScanTile);
  const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  if (vec) hipLaunchKernelGGL(scan_tile_totals<true>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, n, tmp);
  else hipLaunchKernelGGL(scan_tile_totals<false>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, n, tmp);
  hipLaunchKernelGGL(scan_tile_sums, dim3(1), dim3(1024), 0, st, tmp, n_tiles);
  if (vec) hipLaunchKernelGGL(scan_tiles<true>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, out, n, (const uint32_t*)tmp);
  else hipLaunchKernelGGL(scan_tiles<false>, dim3(n_tiles), dim3(kScanThreads), 0, st, in, out, n, (const uint32_t*)tmp);
  PGP_HIP(hipGetLastError());
  return PGP_OK;
}

int build_index(pgp_ctx* ctx, const float* h_xyz, float delta) {
  const int nP = ctx->nP;
  ctx->has_index = false;

  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int finite = 0;
  for (int i = 0; i < nP; ++i) {
    const float* p = h_xyz + 3 * (size_t)i;
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    ++finite;
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], p[k]);
      mx[k] = fmaxf(mx[k], p[k]);
    }
  }
  if (finite == 0) {
    for (int k = 0; k < 3; ++k) mn[k] = mx[k] = 0.f;
  }
  return build_index_bbox(ctx, mn, mx, delta);
}

// Queues a build that pgp_set_scene prepared and put off (pgp_ctx::deferred_build).
int flush_deferred_build(pgp_ctx* ctx) {
  if (!ctx->deferred_build) return PGP_OK;
  std::function<int()> fn = std::move(ctx->deferred_build);
  ctx->deferred_build = nullptr;
  const int rc = fn();
  if (rc != PGP_OK) {   // a step failed half way: nothing of it may keep running over the index buffers
    if (ctx->build_stream) (void)hipStreamSynchronize(ctx->build_stream);
    ctx->index_pending = false;
    ctx->has_index = false;
  }
  return rc;
}

int finish_index(pgp_ctx* ctx) {
  if (!ctx->index_pending) return PGP_OK;
  if (int rc = flush_deferred_build(ctx)) return rc;
  PGP_HIP(hipEventSynchronize(ctx->ev_index));
  ctx->index_pending = false;
  ctx->n_cand = (long long)ctx->h_build_counts[0];
  ctx->n_occ = (long long)ctx->h_build_counts[1];
  PGP_HIP(hipEventElapsedTime(&ctx->build_ms, ctx->ev_build0, ctx->ev_index));
  return PGP_OK;
}

int await_index(pgp_ctx* ctx, hipStream_t stream) {
  if (!ctx->index_pending) return PGP_OK;
  if (int rc = flush_deferred_build(ctx)) return rc;
  {
    // a stream that is being captured into a graph can neither query nor wait for an event recorded outside the capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      set_error("the scene's index is still being built: call pgp_get_index_info (or any scoring call) once before capturing a graph");
      return PGP_ESTATE;
    }
  }
  if (hipEventQuery(ctx->ev_index) == hipSuccess) return finish_index(ctx);
  (void)hipGetLastError();   // hipErrorNotReady is not an error
  PGP_HIP(hipStreamWaitEvent(stream, ctx->ev_index, 0));
  return PGP_OK;
}

// The dense index of a SMALL scene, queued on ctx->build_stream and left running (see pgp_ctx::build_stream): the
// candidate array and the occupied-cell table are sized by what nP points can reach at most, so no count has to come
// back before the next launch.  The points are resident, or on their way (the build waits for pgp_set_scene's upload event on its stream).
static int build_index_async(pgp_ctx* ctx, const GridDesc& g, int r, float delta) {
  const int nP = ctx->nP;
  int rc;
  // (each on its own: a failure half way leaves what exists, and the next call goes on from there)
  if (!ctx->build_stream) {
    // ONE side stream per device for the builds of all contexts: the runtime gives a process four hardware queues
    // (GPU_MAX_HW_QUEUES), and a second context's own side stream came to share a queue with that context's main stream --
    // its build then ran IN FRONT of the congruent-set search instead of beside it (the drop-in's second object: 0.75 ms
    // per call instead of 0.55, profiles/r05_ab/hardware_queues.log).  PGP_BUILD_STREAM_PER_CONTEXT=1: a stream per context.
    if (getenv("PGP_BUILD_STREAM_PER_CONTEXT")) {
      PGP_HIP(hipStreamCreateWithFlags(&ctx->build_stream, hipStreamNonBlocking));
      ctx->build_stream_own = true;
    } else {
      static PerDeviceTable<hipStream_t> shared;   // (host_worker.h; never destroyed)
      hipStream_t got = nullptr;
      if (shared.get(ctx->device, [](hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess; }, &got)) {
        ctx->build_stream = got;
        ctx->build_stream_own = false;
      } else {
        // a device id beyond the table (or a stream that could not be created there): a stream of the context's own, never
        // another device's
        (void)hipGetLastError();
        PGP_HIP(hipStreamCreateWithFlags(&ctx->build_stream, hipStreamNonBlocking));
        ctx->build_stream_own = true;
      }
    }
  }
  if (!ctx->ev_index) PGP_HIP(hipEventCreate(&ctx->ev_index));
  if (!ctx->ev_build0) PGP_HIP(hipEventCreate(&ctx->ev_build0));
  if (!ctx->h_build_counts) PGP_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_build_counts), 64, hipHostMallocDefault));
  hipStream_t st = ctx->build_stream;
  const size_t n_words = (size_t)g.nbx * g.nby * g.nbz;
  const size_t n_cells = n_words * 32, n_scan = n_cells + 1;
  const int n_tiles = (int)((n_scan + kScanTile - 1) / kScanTile);
  const size_t reach = (size_t)(2 * r + 1) * (2 * r + 1) * (2 * r + 1);
  const size_t cand_max = (size_t)nP * reach, occ_max = std::min(n_cells, cand_max);
  if ((rc = ctx->d_cell_start.ensure(n_scan * 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_cell_tmp.ensure(n_scan * 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_build_scan.ensure((size_t)n_tiles * 4 + 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_bitmap.ensure(n_words * 8)) != PGP_OK) return rc;
  if ((rc = ctx->d_cand.ensure((cand_max + 1 + 256) * sizeof(float4))) != PGP_OK) return rc;
  if ((rc = ctx->d_occ_start.ensure((occ_max + 2) * 8)) != PGP_OK) return rc;
  uint32_t* ctr = ctx->d_cell_tmp.as<uint32_t>();
  uint32_t* start = ctx->d_cell_start.as<uint32_t>();
  uint32_t* scan_tmp = ctx->d_build_scan.as<uint32_t>();
  uint2* words = ctx->d_bitmap.as<uint2>();
  // everything from here on is queued on the non-blocking build stream over the context's index buffers: a step that
  // fails half way must not leave that work running beside whatever reuses (or reallocates) the buffers next
  float4* const d_cand = ctx->d_cand.as<float4>();
  const float4* const d_P = ctx->d_P.as<float4>();
  uint2* const d_occ = ctx->d_occ_start.as<uint2>();
  const hipEvent_t ev_up = ctx->scene_upload_pending ? ctx->ev_s : nullptr;   // pgp_set_scene's queued uploads
  auto enqueue = [=]() -> int {   // (by value: it may run after this function has returned, see below)
    int rc;
    if (ev_up) PGP_HIP(hipStreamWaitEvent(st, ev_up, 0));
    PGP_HIP(hipEventRecord(ctx->ev_build0, st));
    PGP_HIP(hipMemsetAsync(ctr, 0, n_scan * 4, st));
    const unsigned pbl = (unsigned)(((size_t)nP * kScatterLanes + 255) / 256);
    hipLaunchKernelGGL(scatter_points_lanes<false>, dim3(pbl), dim3(256), 0, st, g, r, d_P, nP, ctr, (const uint32_t*)nullptr, (float4*)nullptr);
    if ((rc = device_exclusive_scan(ctr, start, n_scan, scan_tmp, st)) != PGP_OK) return rc;
    PGP_HIP(hipMemcpyAsync(&ctx->h_build_counts[0], start + n_cells, 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(scatter_points_lanes<true>, dim3(pbl), dim3(256), 0, st, g, r, d_P, nP, ctr, (const uint32_t*)start, d_cand);
    hipLaunchKernelGGL(make_words, dim3((unsigned)((n_words + 1 + 255) / 256)), dim3(256), 0, st, g,
                       (const uint32_t*)start, words, ctr, n_words);
    if ((rc = device_exclusive_scan(ctr, ctr, n_words + 1, scan_tmp, st)) != PGP_OK) return rc;
    PGP_HIP(hipMemcpyAsync(&ctx->h_build_counts[1], ctr + n_words, 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(fill_occupied, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, g,
                       (const uint32_t*)start, (const uint32_t*)ctr, words, d_occ, n_words, n_cells);
    PGP_HIP(hipGetLastError());
    PGP_HIP(hipEventRecord(ctx->ev_index, st));
    return PGP_OK;
  };
  // Queued now, or put off until the caller is about to wait for the device anyway (pgp_ctx::deferred_build): the
  // drop-in's next steps -- weights, base selection -- then reach the device ~0.1 ms earlier, and the build runs beside
  // the base selection instead of in front of it.
  static const bool defer = !(getenv("PGP_DEFER_BUILD") && atoi(getenv("PGP_DEFER_BUILD")) == 0);
  if (defer) {
    ctx->deferred_build = enqueue;
  } else if ((rc = enqueue()) != PGP_OK) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  ctx->index_pending = true;
  ctx->grid = g;
  ctx->n_cells = (long long)n_cells;
  ctx->n_blocks = 0;
  ctx->n_cand = ctx->n_occ = 0;   // known when the build is through (finish_index)
  ctx->build_ms = 0.f;
  ctx->delta = delta;
  ctx->has_index = true;
  return PGP_OK;
}

// The index of the scene already resident in ctx->d_P, given the bounding box of its finite points.
int build_index_bbox(pgp_ctx* ctx, const float mn[3], const float mx[3], float delta) {
  const int nP = ctx->nP;
  hipStream_t st = ctx->stream;
  ctx->has_index = false;
  if (ctx->deferred_build) {   // the previous scene's build was never queued: nobody asked for its index
    ctx->deferred_build = nullptr;
    ctx->index_pending = false;
  }
  int rc = finish_index(ctx);   // a build still running (its consumers are drained by the caller) owns these buffers
  if (rc != PGP_OK) return rc;
  GridDesc g{};
  int r = 1;
  rc = choose_grid(mn, mx, delta, &g, &r);
  if (rc != PGP_OK) return rc;
  {
    // PGP_ASYNC_BUILD=0: every scene takes the synchronous build below (A/B knob).  The bound on the candidate
    // array (every point in every cell of its reach) keeps the side-stream build to scenes of a few thousand points.
    static const bool async_on = !(getenv("PGP_ASYNC_BUILD") && atoi(getenv("PGP_ASYNC_BUILD")) == 0);
    const size_t reach = (size_t)(2 * r + 1) * (2 * r + 1) * (2 * r + 1);
    if (async_on && !g.sparse && nP > 0 && (size_t)nP * reach * sizeof(float4) <= ((size_t)32 << 20))
      return build_index_async(ctx, g, r, delta);
  }
  struct EventPair {  // destroyed on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } ev;
  PGP_HIP(hipEventCreate(&ev.a));
  PGP_HIP(hipEventCreate(&ev.b));
  const hipEvent_t e0 = ev.a, e1 = ev.b;
  PGP_HIP(hipEventRecord(e0, st));
  const int pb = (nP + 255) / 256;

  size_t n_words = (size_t)g.nbx * g.nby * g.nbz;
  uint32_t n_blk = 0;
  if (g.sparse) {
    // pass 1: count the distinct blocks (scratch key table sized by what nP points can touch at most)
    const double per_pt = (double)((2 * r + 4) / 4 + 1) * ((2 * r + 4) / 4 + 1) * ((2 * r + 2) / 2 + 1);
    const double bound = std::max(1.0, std::min((double)nP * per_pt, (double)n_words));
    const int tmp_log = std::max(10, ceil_log2((long long)(2.0 * bound)));
    if (tmp_log > 31) {
      set_error("scene of %d points: block table scratch out of range", nP);
      return PGP_EINVAL;
    }
    const size_t tmp_cap = (size_t)1 << tmp_log;
    if ((rc = ctx->d_cell_tmp.ensure((tmp_cap + 1) * 4)) != PGP_OK) return rc;
    uint32_t* keys = ctx->d_cell_tmp.as<uint32_t>();
    PGP_HIP(hipMemsetAsync(keys, 0xFF, tmp_cap * 4, st));
    PGP_HIP(hipMemsetAsync(keys + tmp_cap, 0, 4, st));
    if (nP > 0)
      hipLaunchKernelGGL(blocks_count, dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP, keys,
                         (uint32_t)(tmp_cap - 1), 32 - tmp_log, keys + tmp_cap);
    PGP_HIP(hipMemcpyAsync(&n_blk, keys + tmp_cap, 4, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipStreamSynchronize(st));
    // pass 2: the table itself, at most one eighth full
    // table entries per block (knob 2 .. 16): at C2 forced sparse 2 -> 29.6, 4 -> 38.5, 8 -> 40.9 M hyp/s weighted
    // (tools/sparse_index_time.py; dense 46.2) -- the wave waits for the lane with the longest probe chain
    long long slack = 8;
    if (const char* v = getenv("PGP_TAB_SLACK")) slack = std::min(16, std::max(2, atoi(v)));
    const int tab_log = std::max(10, ceil_log2(slack * (long long)n_blk));
    const size_t tab_cap = (size_t)1 << tab_log;
    g.tab_mask = (uint32_t)(tab_cap - 1);
    g.tab_shift = 32 - tab_log;
    if ((rc = ctx->d_blocktab.ensure(tab_cap * 16)) != PGP_OK) return rc;
    keys = ctx->d_cell_tmp.as<uint32_t>();
    PGP_HIP(hipMemsetAsync(ctx->d_blocktab.p, 0xFF, tab_cap * 16, st));
    PGP_HIP(hipMemsetAsync(keys, 0, 4, st));
    if (nP > 0)
      hipLaunchKernelGGL(blocks_insert, dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP,
                         ctx->d_blocktab.as<uint4>(), keys);
    PGP_HIP(hipStreamSynchronize(st));  // d_cell_tmp is re-sized below
    n_words = std::max<size_t>(n_blk, 1);  // words are per block SLOT from here on
  }
  const size_t n_cells = n_words * 32;  // blocked numbering, padded to whole 4 x 4 x 2 blocks
  const size_t n_scan = n_cells + 1;
  const int n_tiles = (int)((n_scan + kScanTile - 1) / kScanTile);

  if ((rc = ctx->d_cell_start.ensure(n_scan * 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_cell_tmp.ensure(n_scan * 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_scan_tmp.ensure((size_t)n_tiles * 4 + 4)) != PGP_OK) return rc;
  if ((rc = ctx->d_bitmap.ensure(n_words * 8)) != PGP_OK) return rc;
  const uint4* tab = g.sparse ? ctx->d_blocktab.as<uint4>() : nullptr;

  uint32_t* ctr = ctx->d_cell_tmp.as<uint32_t>();
  uint32_t* start = ctx->d_cell_start.as<uint32_t>();
  PGP_HIP(hipMemsetAsync(ctr, 0, n_scan * 4, st));
  if (nP > 0) {
    if (g.sparse)
      hipLaunchKernelGGL((scatter_points<false, true>), dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP,
                         ctr, (const uint32_t*)nullptr, (float4*)nullptr, tab);
    else
      hipLaunchKernelGGL((scatter_points<false, false>), dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP,
                         ctr, (const uint32_t*)nullptr, (float4*)nullptr, tab);
  }
  if ((rc = device_exclusive_scan(ctr, start, n_scan, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  uint32_t total = 0;
  PGP_HIP(hipMemcpyAsync(&total, start + n_cells, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  if ((rc = ctx->d_cand.ensure(((size_t)total + 1 + 256) * sizeof(float4))) != PGP_OK) return rc;
  PGP_HIP(hipMemsetAsync(ctr, 0, n_scan * 4, st));
  if (nP > 0) {
    if (g.sparse)
      hipLaunchKernelGGL((scatter_points<true, true>), dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP, ctr,
                         (const uint32_t*)start, ctx->d_cand.as<float4>(), tab);
    else
      hipLaunchKernelGGL((scatter_points<true, false>), dim3(pb), dim3(256), 0, st, g, r, ctx->d_P.as<float4>(), nP, ctr,
                         (const uint32_t*)start, ctx->d_cand.as<float4>(), tab);
  }
  // two-level compaction: words {bits, rank base} + offsets of occupied cells only
  uint2* words = ctx->d_bitmap.as<uint2>();
  hipLaunchKernelGGL(make_words, dim3((unsigned)((n_words + 1 + 255) / 256)), dim3(256), 0, st, g,
                     (const uint32_t*)start, words, ctr, n_words);
  if ((rc = device_exclusive_scan(ctr, ctr, n_words + 1, ctx->d_scan_tmp.as<uint32_t>(), st)) != PGP_OK) return rc;
  uint32_t n_occ = 0;
  PGP_HIP(hipMemcpyAsync(&n_occ, ctr + n_words, 4, hipMemcpyDeviceToHost, st));
  PGP_HIP(hipStreamSynchronize(st));
  if ((rc = ctx->d_occ_start.ensure(((size_t)n_occ + 2) * 8)) != PGP_OK) return rc;
  hipLaunchKernelGGL(fill_occupied, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, g,
                     (const uint32_t*)start, (const uint32_t*)ctr, words,
                     ctx->d_occ_start.as<uint2>(), n_words, n_cells);
  if (g.sparse)
    hipLaunchKernelGGL(blocks_finish, dim3((g.tab_mask + 256) / 256), dim3(256), 0, st, ctx->d_blocktab.as<uint4>(),
                       g.tab_mask + 1, (const uint2*)words);
  PGP_HIP(hipGetLastError());
  PGP_HIP(hipEventRecord(e1, st));
  PGP_HIP(hipStreamSynchronize(st));
  ctx->n_occ = (long long)n_occ;
  PGP_HIP(hipEventElapsedTime(&ctx->build_ms, e0, e1));

  ctx->grid = g;
  ctx->n_cells = (long long)n_cells;
  ctx->n_blocks = (long long)n_blk;
  ctx->n_cand = (long long)total;
  ctx->delta = delta;
  ctx->has_index = true;
  return PGP_OK;
}

}  // namespace pgp
