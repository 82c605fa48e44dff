// CPU check of the chain codes of the tile map (ryujin_amd/csrc/host_layout.hpp, TileDesc::chain / chain_loads): for
// every tile of a synthetic mesh and every lane the mask does NOT name, the relation the sweeps rely on must hold --
//   kChainPrevColumn: cols(l, c) == cols(l + 1, c - 1)     kChainOwnNext / kChainOwnPrev: cols(l, c) == row(l) +- 1
// -- and the end-lane-only flag must say what the mask says. Prints the counts the GPU tests and bench.py report.
// usage: host_layout_chains dim nx ny nz   (test infrastructure; built by tests/test_host_layout_chains.py)
#include <cstdio>
#include <cstdlib>

#include "host_layout.hpp"
#include "ryujin_synth.h"

using namespace ryujin_hip;

int main(int argc, char **argv)
{
  if (argc < 5)
    return 2;
  ryujin_synth_spec spec{};
  spec.dim = std::atoi(argv[1]);
  for (int d = 0; d < 3; ++d) {
    spec.n_cells[d] = d < spec.dim ? (uint32_t)std::atoi(argv[2 + d]) : 1u;
    spec.lower[d] = 0.;
    spec.upper[d] = 1.;
  }
  for (int f = 0; f < 6; ++f)
    spec.bc[f] = RYUJIN_BC_SLIP;
  spec.cut_kind = RYUJIN_CUT_NONE;
  spec.n_ranks = 1;
  ryujin_synth *synth = ryujin_synth_build(&spec);
  if (!synth) {
    std::fprintf(stderr, "%s\n", ryujin_synth_last_error());
    return 3;
  }
  SellLayout L;
  L.build(*ryujin_synth_offline(synth));
  L.build_tiles();
  uint64_t bad = 0, chained = 0, entries = 0, end_lane = 0, off_diagonal = 0;
  for (uint32_t s = 0; s < L.n_slices; ++s) {
    const uint32_t width = L.slice_off[s + 1] - L.slice_off[s];
    off_diagonal += width > 0 ? width - 1 : 0;
    for (uint32_t c = 0; c < width; ++c) {
      const uint64_t t = (uint64_t)L.slice_off[s] + c;
      const uint32_t kind = L.tiles[t].chain & kChainKindMask;
      const uint64_t loads = L.chain_loads[t];
      if (kind == kChainNone) {
        bad += loads != ~0ull || (L.tiles[t].chain & kChainEndLaneOnly) != 0u;
        continue;
      }
      ++chained;
      bad += c == 0 || (kind == kChainPrevColumn && c < 2) || (uint64_t)(s + 1) * kWave > L.n_owned;
      const uint64_t end_mask = kind == kChainOwnPrev ? 1ull : 1ull << 63;
      const bool flag = (L.tiles[t].chain & kChainEndLaneOnly) != 0u;
      bad += flag != (loads == end_mask);
      end_lane += flag;
      bad += (loads & end_mask) == 0ull; /* the lane at the end of the wave has no neighbour to take from */
      for (uint32_t l = 0; l < kWave; ++l) {
        if ((loads >> l) & 1ull)
          continue;
        ++entries;
        const uint64_t row = (uint64_t)s * kWave + l;
        const uint32_t col = L.cols[t * kWave + l];
        if (kind == kChainPrevColumn)
          bad += l + 1 >= kWave || col != L.cols[(t - 1) * kWave + l + 1];
        else if (kind == kChainOwnNext)
          bad += l + 1 >= kWave || col != row + 1;
        else
          bad += l == 0 || (uint64_t)col + 1 != row;
      }
    }
  }
  bad += chained != L.n_chained_tiles || entries != L.n_chained_entries || end_lane != L.n_end_lane_tiles;
  std::printf("{\"n_tiles\": %llu, \"off_diagonal_tiles\": %llu, \"n_regular_tiles\": %llu, \"n_chained_tiles\": %llu, "
              "\"n_chained_entries\": %llu, \"n_end_lane_tiles\": %llu, \"violations\": %llu}\n",
              (unsigned long long)L.slice_off[L.n_slices], (unsigned long long)off_diagonal,
              (unsigned long long)L.n_regular_tiles, (unsigned long long)chained, (unsigned long long)entries,
              (unsigned long long)end_lane, (unsigned long long)bad);
  ryujin_synth_free(synth);
  return bad == 0 ? 0 : 1;
}
