// Host-side import of the reference's SparsityPatternSIMD storage into the device layout.
//
// Input : ryujin_hip_offline in the reference layout (source/sparse_matrix_simd.h:311-350,
//         403-418; SIMD-interleaved internal rows + CSR remainder, diagonal first).
// Output: SELL-64 ("sliced ELL", slice height = one wavefront):
//           owned rows are cut into slices of 64 consecutive rows; slice s is padded to its
//           longest row and stored column-major, i.e. entry (row, col_idx) of a scalar matrix at
//             pos = (slice_off[s] + col_idx) * 64 + row % 64
//           -- the reference's own interleave with simd_length -> 64, so that lane l of a wave
//           reads consecutive addresses for every col_idx (fully coalesced, no LDS transpose).
//           Multi-component matrices pair components so that every lane moves 16 bytes per load:
//             comp d of entry pos: base = (slice_off[s]+col_idx)*64*n_comp,
//               full pair g = d/2:  base + g*128 + lane*2 + d%2
//               odd tail component: base + g*128 + lane
//         ghost rows (only transposes of owned entries, needed for l_ji / c_ji look-ups) are
//         appended as plain CSR behind the SELL region; a transposed-position table addresses
//         both regions uniformly.

#pragma once

#include <algorithm>
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "ryujin_hip.h"

namespace ryujin_hip
{
  constexpr uint32_t kWave = 64;

  /* fn(begin, end) over [0, n) in contiguous chunks on up to 16 host threads (setup only: the conversion of a
   * 200^3 mesh touches 2e8 matrix entries several times); the first exception of a chunk is rethrown */
  template <typename F>
  void parallel_chunks(const uint64_t n, F &&fn)
  {
    const unsigned hw = std::thread::hardware_concurrency();
    const uint64_t n_threads = std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), n / 4096 + 1);
    if (n_threads <= 1) {
      fn((uint64_t)0, n);
      return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errors(n_threads);
    const uint64_t chunk = (n + n_threads - 1) / n_threads;
    for (uint64_t t = 0; t < n_threads; ++t)
      pool.emplace_back([&, t]() {
        try {
          fn(std::min(n, t * chunk), std::min(n, (t + 1) * chunk));
        } catch (...) {
          errors[t] = std::current_exception();
        }
      });
    for (auto &th : pool)
      th.join();
    for (auto &e : errors)
      if (e)
        std::rethrow_exception(e);
  }

  /* logical (row, col_idx) view of the reference layout */
  struct RefView {
    uint32_t n_internal, sl;
    const uint64_t *row_starts;

    explicit RefView(const ryujin_hip_offline &o)
        : n_internal(o.n_internal)
        , sl(o.simd_length ? o.simd_length : 1)
        , row_starts(o.row_starts)
    {
      if (sl > 1 && n_internal % sl != 0)
        throw std::invalid_argument("n_internal must be a multiple of simd_length");
    }

    uint32_t row_length(uint32_t row) const
    {
      if (row < n_internal) {
        const uint32_t g = row / sl;
        return (uint32_t)((row_starts[g + 1] - row_starts[g]) / sl);
      }
      return (uint32_t)(row_starts[row + 1] - row_starts[row]);
    }

    uint64_t scalar_pos(uint32_t row, uint32_t c) const
    {
      return row < n_internal ? row_starts[row / sl] + (uint64_t)c * sl + row % sl
                              : row_starts[row] + c;
    }

    uint64_t data_pos(uint32_t row, uint32_t c, uint32_t n_comp, uint32_t d) const
    {
      return row < n_internal
                 ? (row_starts[row / sl] + (uint64_t)c * sl) * n_comp + (uint64_t)d * sl + row % sl
                 : (row_starts[row] + c) * n_comp + d;
    }
  };


  /* TILE MAP. A tile is one column of one SELL-64 slice: 64 matrix entries, one per row of the slice. On a
   * structured patch of a mesh the 64 rows of a slice see their c-th neighbour at the SAME index offset, j = i + delta,
   * and the transposed entries (j, i) then sit at consecutive positions as well -- in (at most) two runs, because the
   * rows i + delta straddle a slice boundary unless delta is a multiple of 64. Such a tile is described by 16 bytes
   * instead of the 64 column indices and 64 transposed positions (512 bytes) the sweeps would otherwise stream:
   *   j(lane)            = row + delta
   *   transposed(lane)   = (lane < 64 - (delta & 63) ? ta : tb) + lane
   * Tiles that do not fit (boundary rows, short rows, unstructured patches) carry delta = kTileIrregular and the
   * sweeps read the explicit arrays for them, which always exist. One wave-uniform 16-byte load per tile.
   *
   * CHAINS (`chain`, `chain_loads`). The columns of a row are sorted (diagonal first, then ascending), so on a lattice
   * the neighbours come in runs of consecutive indices: i + d - 1, i + d, i + d + 1 -- three columns whose 64 x node
   * data are the SAME 64 nodes shifted by one lane. What row l needs in column c is what row l + 1 fetched in column
   * c - 1; the neighbours i - 1 and i + 1 are the rows l - 1 and l + 1 of the slice themselves. Every tile of a full
   * slice -- regular or not -- therefore says where its node data can be had without a gather, lane by lane:
   *   kChainPrevColumn  lane l takes lane l + 1's data of column c - 1      where cols(l, c) == cols(l + 1, c - 1)
   *   kChainOwnNext     lane l takes the slice's own row data of lane l + 1 where cols(l, c) == row(l) + 1
   *   kChainOwnPrev     ... of lane l - 1                                   where cols(l, c) == row(l) - 1
   * with `chain_loads[tile]`, a 64-bit mask of the lanes for which that does NOT hold and which fetch their node from
   * memory: the lane at the end of the wave, the rows at the end of a lattice row, boundary rows. (The invariant:
   * behind column c every lane holds the data of node cols(l, c), padding entries included, however it got them.) The
   * kind that serves the most lanes wins; below kChainMinLanes the tile is gathered as ever. Bit 2 of `chain` marks
   * the tiles whose mask is just the lane at the end of the wave (every regular tile of a run): the 2-D sweeps chain
   * those only and never read a mask (with masks their step 5 measured 2 - 3 % slower for one more chained tile in a
   * hundred; in 3-D, where the lattice rows are short against a slice, the masks take the chained tiles of C4 from 24
   * to 67 %, profiles/r06ai_ab_chain_masks_*.log). Of the eight gathers of a
   * 2-D Q1 row two are left, of the 26 in 3-D eight (kernels_euler.hpp, chained gathers). */
  struct TileDesc {
    int32_t delta;
    uint32_t ta, tb;
    uint32_t chain;
  };
  constexpr int32_t kTileIrregular = INT32_MIN;
  constexpr uint32_t kChainNone = 0, kChainPrevColumn = 1, kChainOwnPrev = 2, kChainOwnNext = 3;
  constexpr int kChainMinLanes = 32;
  constexpr uint32_t kChainKindMask = 3, kChainEndLaneOnly = 4; /* bit 2 of `chain`: chain_loads is just the lane at the end of the wave */

  struct SellLayout {
    std::vector<TileDesc> tiles;     /* [slice_off[n_slices]] */
    std::vector<uint64_t> chain_loads; /* [slice_off[n_slices]] the lanes of a chained tile that fetch their node themselves */
    uint64_t n_regular_tiles = 0;
    uint64_t n_chained_tiles = 0;    /* tiles with node data from a neighbouring lane (TileDesc::chain) */
    uint64_t n_chained_entries = 0;  /* ... and the matrix entries (lanes) of them that are served that way */
    uint64_t n_end_lane_tiles = 0;   /* of the chained tiles: those in which only the lane at the end of the wave loads (what the 2-D sweeps chain) */
    uint32_t n_owned = 0, n_relevant = 0, n_slices = 0, rows_padded = 0;
    uint32_t max_row_len = 0;
    std::vector<uint32_t> slice_off; /* [n_slices+1], in units of 64-entry columns */
    std::vector<uint16_t> row_len;    /* [rows_padded] logical row length (0 beyond n_owned) */
    std::vector<uint64_t> ghost_ptr; /* [n_ghost+1] */
    uint64_t nnz_sell = 0, nnz_total = 0, nnz_owned_logical = 0;
    std::vector<uint32_t> cols;      /* [nnz_total]; padding entries point at their own row */
    std::vector<uint32_t> idx_t;     /* [nnz_total] position of the transposed entry */
    std::vector<uint64_t> logical_ptr; /* [n_owned+1] logical CSR offsets (debug_fetch order) */

    uint32_t n_ghost() const { return n_relevant - n_owned; }

    uint64_t pos(uint32_t row, uint32_t c) const
    {
      if (row < n_owned)
        return ((uint64_t)slice_off[row / kWave] + c) * kWave + row % kWave;
      return nnz_sell + ghost_ptr[row - n_owned] + c;
    }

    /* position of component d of the n_comp matrix entry with scalar position p */
    uint64_t comp_pos(uint64_t p, uint32_t n_comp, uint32_t d) const
    {
      if (p >= nnz_sell)
        return nnz_sell * n_comp + (p - nnz_sell) * n_comp + d;
      const uint64_t lane = p % kWave, colbase = p / kWave;
      const uint64_t base = colbase * kWave * n_comp;
      const uint32_t g = d / 2;
      if (2 * g + 1 < n_comp)
        return base + (uint64_t)g * 128 + lane * 2 + d % 2;
      return base + (uint64_t)g * 128 + lane;
    }

    void build(const ryujin_hip_offline &o)
    {
      const RefView ref(o);
      n_owned = o.n_owned;
      n_relevant = o.n_relevant;
      n_slices = (n_owned + kWave - 1) / kWave;
      rows_padded = n_slices * kWave;

      row_len.assign(rows_padded, 0);
      slice_off.assign((size_t)n_slices + 1, 0);
      logical_ptr.assign((size_t)n_owned + 1, 0);
      max_row_len = 0;
      for (uint32_t i = 0; i < n_owned; ++i) {
        const uint32_t len = ref.row_length(i);
        /* (a row is a lane of its SELL-64 slice and may be as wide as it likes: cG Q2 / Q3 and dG stencils have 125 to
         * several hundred entries in 3-D, discretization.h:131-151; 1023: the column field of the limiter's list of
         * undecided pairs, kernels_limiter.hpp) */
        if (len == 0 || len > 1023)
          throw std::invalid_argument("row length must be in [1,1023]");
        row_len[i] = (uint16_t)len;
        max_row_len = std::max(max_row_len, len);
        logical_ptr[i + 1] = logical_ptr[i] + len;
      }
      nnz_owned_logical = logical_ptr[n_owned];
      for (uint32_t s = 0; s < n_slices; ++s) {
        uint32_t m = 0;
        for (uint32_t l = 0; l < kWave; ++l)
          m = std::max<uint32_t>(m, row_len[(size_t)s * kWave + l]);
        slice_off[s + 1] = slice_off[s] + m;
      }
      nnz_sell = (uint64_t)slice_off[n_slices] * kWave;

      ghost_ptr.assign((size_t)n_ghost() + 1, 0);
      for (uint32_t g = 0; g < n_ghost(); ++g)
        ghost_ptr[g + 1] = ghost_ptr[g] + ref.row_length(n_owned + g);
      nnz_total = nnz_sell + ghost_ptr[n_ghost()];
      if (nnz_total >= 0xFFFFFFFFull)
        throw std::invalid_argument("more than 2^32 matrix entries per rank are not supported "
                                    "(same limit as sparse_matrix_simd.template.h:88-92)");

      /* columns; padding -> own row */
      cols.assign(nnz_total, 0);
      parallel_chunks(n_slices, [&](const uint64_t s0, const uint64_t s1) {
        for (uint32_t s = (uint32_t)s0; s < (uint32_t)s1; ++s) {
          const uint32_t width = slice_off[s + 1] - slice_off[s];
          for (uint32_t l = 0; l < kWave; ++l) {
            const uint32_t row = s * kWave + l;
            const uint32_t self = row < n_owned ? row : (n_owned ? n_owned - 1 : 0);
            for (uint32_t c = 0; c < width; ++c) {
              const uint64_t p = ((uint64_t)slice_off[s] + c) * kWave + l;
              cols[p] = (row < n_owned && c < row_len[row]) ? o.columns[ref.scalar_pos(row, c)] : self;
            }
          }
        }
      });
      for (uint32_t i = n_owned; i < n_relevant; ++i) {
        const uint32_t len = ref.row_length(i);
        for (uint32_t c = 0; c < len; ++c)
          cols[pos(i, c)] = o.columns[ref.scalar_pos(i, c)];
      }
      for (uint32_t i = 0; i < n_relevant; ++i)
        if (o.columns[ref.scalar_pos(i, 0)] != i)
          throw std::invalid_argument("row " + std::to_string(i) +
                                      " does not start with its diagonal entry");

      /* transposed positions (rows are: diagonal, then ascending columns) */
      auto logical_len = [&](uint32_t row) {
        return row < n_owned ? (uint32_t)row_len[row]
                             : (uint32_t)(ghost_ptr[row - n_owned + 1] - ghost_ptr[row - n_owned]);
      };
      auto find_in_row = [&](uint32_t row, uint32_t target) -> int64_t {
        const uint32_t len = logical_len(row);
        uint32_t lo = 1, hi = len;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) / 2;
          if (cols[pos(row, mid)] < target)
            lo = mid + 1;
          else
            hi = mid;
        }
        if (lo < len && cols[pos(row, lo)] == target)
          return (int64_t)lo;
        return -1;
      };
      idx_t.assign(nnz_total, 0);
      parallel_chunks(n_relevant, [&](const uint64_t i0, const uint64_t i1) {
        for (uint32_t i = (uint32_t)i0; i < (uint32_t)i1; ++i) {
          const uint32_t len = logical_len(i);
          for (uint32_t c = 0; c < len; ++c) {
            const uint64_t p = pos(i, c);
            const uint32_t j = cols[p];
            if (c == 0 || j == i) {
              idx_t[p] = (uint32_t)pos(i, 0);
              continue;
            }
            const int64_t ct = find_in_row(j, i);
            if (ct < 0) {
              if (i < n_owned)
                throw std::invalid_argument("sparsity pattern is not structurally symmetric");
              idx_t[p] = (uint32_t)p;
              continue;
            }
            idx_t[p] = (uint32_t)pos(j, (uint32_t)ct);
          }
        }
      });
      /* padding entries: transpose -> themselves */
      parallel_chunks(n_slices, [&](const uint64_t s0, const uint64_t s1) {
        for (uint32_t s = (uint32_t)s0; s < (uint32_t)s1; ++s) {
          const uint32_t width = slice_off[s + 1] - slice_off[s];
          for (uint32_t l = 0; l < kWave; ++l) {
            const uint32_t row = s * kWave + l;
            for (uint32_t c = (row < n_owned ? row_len[row] : 0); c < width; ++c) {
              const uint64_t p = ((uint64_t)slice_off[s] + c) * kWave + l;
              idx_t[p] = (uint32_t)p;
            }
          }
        }
      });
    }

    /* The distance, in rows, between a row and its neighbour in the next lattice row (dim 2) / lattice plane (dim 3)
     * of a structured patch: the middle of the topmost cluster of index offsets that whole 64-row tiles agree on
     * ({W-1, W, W+1} in 2-D; {P-W-1 ... P+W+1} around P in 3-D). 0: no such structure (unstructured mesh, 1-D). */
    uint32_t lattice_stride(const int dim) const
    {
      if (dim < 2)
        return 0;
      std::vector<std::pair<uint32_t, uint64_t>> hist; /* (|delta|, tiles) */
      uint64_t total = 0;
      for (uint32_t s = 0; s < n_slices; s += 5) {
        if ((uint64_t)(s + 1) * kWave > n_owned)
          continue;
        uint32_t min_len = 0xffffffffu;
        for (uint32_t l = 0; l < kWave; ++l)
          min_len = std::min<uint32_t>(min_len, row_len[(size_t)s * kWave + l]);
        for (uint32_t c = 1; c < min_len; ++c) {
          const uint64_t p0 = ((uint64_t)slice_off[s] + c) * kWave;
          const int64_t delta = (int64_t)cols[p0] - (int64_t)((uint64_t)s * kWave);
          bool ok = true;
          for (uint32_t l = 1; l < kWave && ok; ++l)
            ok = (int64_t)cols[p0 + l] - (int64_t)((uint64_t)s * kWave + l) == delta;
          if (!ok)
            continue;
          const uint32_t a = (uint32_t)(delta < 0 ? -delta : delta);
          ++total;
          bool found = false;
          for (auto &h : hist)
            if (h.first == a) {
              ++h.second;
              found = true;
              break;
            }
          if (!found && hist.size() < 4096)
            hist.emplace_back(a, 1);
        }
      }
      std::vector<uint32_t> v;
      for (const auto &h : hist)
        if (h.first >= 16 && h.second * 200 >= total)
          v.push_back(h.first);
      std::sort(v.begin(), v.end());
      if (dim == 2 && v.size() >= 2)
        return v[v.size() - 2];
      if (dim == 3 && v.size() >= 12)
        return v[v.size() - 5];
      return 0;
    }

    /* tile map (see TileDesc): call after build() */
    void build_tiles()
    {
      tiles.assign(slice_off[n_slices], TileDesc{kTileIrregular, 0u, 0u, 0u});
      chain_loads.assign(slice_off[n_slices], ~0ull);
      std::vector<uint64_t> regular(n_slices, 0), chained(n_slices, 0), chained_lanes(n_slices, 0), end_lane(n_slices, 0);
      parallel_chunks(n_slices, [&](const uint64_t s0, const uint64_t s1) {
        for (uint32_t s = (uint32_t)s0; s < (uint32_t)s1; ++s) {
          const uint32_t width = slice_off[s + 1] - slice_off[s];
          if ((uint64_t)(s + 1) * kWave > n_owned)
            continue; /* a slice with padding rows */
          uint32_t min_len = 0xffffffffu;
          for (uint32_t l = 0; l < kWave; ++l)
            min_len = std::min<uint32_t>(min_len, row_len[(size_t)s * kWave + l]);
          for (uint32_t c = 0; c < std::min(width, min_len); ++c) {
            const uint64_t p0 = ((uint64_t)slice_off[s] + c) * kWave;
            const int64_t delta = (int64_t)cols[p0] - (int64_t)((uint64_t)s * kWave);
            if (delta <= (int64_t)INT32_MIN || delta > (int64_t)INT32_MAX)
              continue;
            bool ok = true;
            for (uint32_t l = 1; l < kWave && ok; ++l)
              ok = (int64_t)cols[p0 + l] - (int64_t)((uint64_t)s * kWave + l) == delta;
            if (!ok)
              continue;
            const uint32_t dm = (uint32_t)((int32_t)delta & 63);
            const uint32_t split = kWave - dm; /* lanes [0, split): run a, lanes [split, 64): run b */
            const uint32_t ta = idx_t[p0];
            const uint32_t tb = dm != 0 ? idx_t[p0 + split] - split : ta;
            for (uint32_t l = 0; l < kWave && ok; ++l)
              ok = idx_t[p0 + l] == (l < split ? ta : tb) + l;
            if (!ok)
              continue;
            tiles[(uint64_t)slice_off[s] + c] = TileDesc{(int32_t)delta, ta, tb, kChainNone};
            ++regular[s];
          }
          /* chains (TileDesc): every row of such a slice is an owned row, row s * 64 + l: "the data of lane l + 1" is
           * the data of row i + 1. Column 0 is the diagonal; the sweeps gather column 1 as ever and chain from 2 on
           * (kChainPrevColumn) / from 1 on (the own rows). */
          for (uint32_t c = 1; c < width; ++c) {
            const uint64_t p0 = ((uint64_t)slice_off[s] + c) * kWave;
            uint64_t ok_prev = 0, ok_own_next = 0, ok_own_prev = 0;
            for (uint32_t l = 0; l < kWave; ++l) {
              const uint64_t row = (uint64_t)s * kWave + l;
              if (c >= 2 && l + 1 < kWave && cols[p0 + l] == cols[p0 - kWave + l + 1])
                ok_prev |= 1ull << l;
              if (l + 1 < kWave && cols[p0 + l] == row + 1)
                ok_own_next |= 1ull << l;
              if (l >= 1 && cols[p0 + l] + 1 == row)
                ok_own_prev |= 1ull << l;
            }
            const int n_prev = __builtin_popcountll(ok_prev), n_next = __builtin_popcountll(ok_own_next),
                      n_own_prev = __builtin_popcountll(ok_own_prev);
            uint32_t kind = kChainNone;
            uint64_t ok = 0;
            if (n_next >= kChainMinLanes && n_next >= n_prev && n_next >= n_own_prev)
              kind = kChainOwnNext, ok = ok_own_next;
            else if (n_own_prev >= kChainMinLanes && n_own_prev >= n_prev)
              kind = kChainOwnPrev, ok = ok_own_prev;
            else if (n_prev >= kChainMinLanes)
              kind = kChainPrevColumn, ok = ok_prev;
            const uint64_t end_lane_mask = kind == kChainOwnPrev ? 1ull : 1ull << 63;
            tiles[(uint64_t)slice_off[s] + c].chain = kind | ((kind != kChainNone && ~ok == end_lane_mask) ? kChainEndLaneOnly : 0u);
            chain_loads[(uint64_t)slice_off[s] + c] = ~ok;
            chained[s] += kind != kChainNone;
            end_lane[s] += kind != kChainNone && ~ok == end_lane_mask;
            chained_lanes[s] += (uint64_t)__builtin_popcountll(ok);
          }
        }
      });
      n_regular_tiles = n_chained_tiles = 0;
      for (const uint64_t n : regular)
        n_regular_tiles += n;
      for (const uint64_t n : chained)
        n_chained_tiles += n;
      n_chained_entries = n_end_lane_tiles = 0;
      for (const uint64_t n : chained_lanes)
        n_chained_entries += n;
      for (const uint64_t n : end_lane)
        n_end_lane_tiles += n;
    }

    /* reference layout -> device layout (padding = 0) */
    std::vector<double> scatter(const ryujin_hip_offline &o, const double *data, uint32_t n_comp) const
    {
      const RefView ref(o);
      std::vector<double> out(nnz_total * n_comp, 0.);
      parallel_chunks(n_relevant, [&](const uint64_t i0, const uint64_t i1) {
        for (uint32_t i = (uint32_t)i0; i < (uint32_t)i1; ++i) {
          const uint32_t len = ref.row_length(i);
          for (uint32_t c = 0; c < len; ++c) {
            const uint64_t p = pos(i, c);
            for (uint32_t d = 0; d < n_comp; ++d)
              out[comp_pos(p, n_comp, d)] = data[ref.data_pos(i, c, n_comp, d)];
          }
        }
      });
      return out;
    }

    /* device layout -> logical CSR over owned rows (AoS per entry) */
    void gather_logical(const std::vector<double> &dev, uint32_t n_comp, double *out) const
    {
      parallel_chunks(n_owned, [&](const uint64_t i0, const uint64_t i1) {
        for (uint32_t i = (uint32_t)i0; i < (uint32_t)i1; ++i)
          for (uint32_t c = 0; c < row_len[i]; ++c) {
            const uint64_t p = pos(i, c);
            for (uint32_t d = 0; d < n_comp; ++d)
              out[(logical_ptr[i] + c) * n_comp + d] = dev[comp_pos(p, n_comp, d)];
          }
      });
    }
  };
} // namespace ryujin_hip
