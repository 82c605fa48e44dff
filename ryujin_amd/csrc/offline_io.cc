// offline_io.cc -- reader/writer for OfflineData dumps (include/ryujin_offline_io.h).
// Host-only; part of libryujin_synth.so next to the synthetic generator.

#include "ryujin_offline_io.h"

#include "host_layout.hpp"

#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
  thread_local std::string g_error;

  constexpr char kMagic[8] = {'R', 'Y', 'J', 'O', 'F', 'F', 'L', '1'};
  constexpr uint32_t kVersion = 1;

  struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void feed(const void *p, size_t n)
    {
      /* 8 bytes at a time (all sections are 8-byte aligned and padded): word-wise FNV-1a */
      const unsigned char *b = static_cast<const unsigned char *>(p);
      size_t i = 0;
      for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 1099511628211ull;
      }
      for (; i < n; ++i)
        h = (h ^ b[i]) * 1099511628211ull;
    }
  };

  struct Writer {
    FILE *f;
    Fnv sum;
    void raw(const void *p, size_t n)
    {
      if (n && std::fwrite(p, 1, n, f) != n)
        throw std::runtime_error("short write");
      sum.feed(p, n);
    }
    template <typename T>
    void scalar(T v)
    {
      raw(&v, sizeof(T));
    }
    void section(const void *p, uint64_t n_bytes)
    {
      if (n_bytes && !p)
        throw std::invalid_argument("NULL array with non-zero extent");
      scalar<uint64_t>(n_bytes);
      raw(p, n_bytes);
      static const char zeros[8] = {0};
      raw(zeros, (8 - n_bytes % 8) % 8);
    }
  };

  struct Reader {
    FILE *f;
    Fnv sum;
    void raw(void *p, size_t n)
    {
      if (n && std::fread(p, 1, n, f) != n)
        throw std::runtime_error("truncated file");
      sum.feed(p, n);
    }
    template <typename T>
    T scalar()
    {
      T v;
      raw(&v, sizeof(T));
      return v;
    }
    template <typename T>
    void section(std::vector<T> &out, uint64_t n_expected, const char *name)
    {
      const uint64_t n_bytes = scalar<uint64_t>();
      if (n_bytes != n_expected * sizeof(T))
        throw std::runtime_error(std::string("section '") + name + "': " + std::to_string(n_bytes) +
                                 " bytes, expected " + std::to_string(n_expected * sizeof(T)));
      out.resize(n_expected);
      raw(out.data(), n_bytes);
      char pad[8];
      raw(pad, (8 - n_bytes % 8) % 8);
    }
  };
} // namespace

struct ryujin_offline_file {
  ryujin_hip_offline view{};
  int dim = 0, n_init_prec = 0;
  uint64_t nnz = 0;
  bool has_positions = false, has_b_positions = false;
  std::vector<uint64_t> row_starts;
  std::vector<uint32_t> columns, b_i, p_i, p_col, p_j, send_off, send_idx, recv_off, row_send_off,
      row_send_row, row_send_col;
  std::vector<int32_t> nbr_rank;
  std::vector<uint8_t> b_id;
  std::vector<double> cij, mij, mi, mi_inv, b_normal, initial_precomputed, positions, b_positions,
      incidence, mass_matrix_inverse;

  void validate() const
  {
    const ryujin_hip_offline &o = view;
    auto fail = [](const std::string &m) { throw std::runtime_error("invalid OfflineData dump: " + m); };
    if (!(o.n_export <= o.n_internal && o.n_internal <= o.n_owned && o.n_owned <= o.n_relevant))
      fail("index ranges are not nested");
    const uint32_t sl = o.simd_length ? o.simd_length : 1;
    if (sl > 1 && o.n_internal % sl != 0)
      fail("n_internal is not a multiple of simd_length");
    ryujin_hip::RefView ref(o);
    /* monotone row_starts over the meaningful index sets, total == nnz */
    const uint32_t n_groups = o.n_internal / sl;
    for (uint32_t g = 0; g < n_groups; ++g)
      if (row_starts[g + 1] < row_starts[g] || (row_starts[g + 1] - row_starts[g]) % sl != 0)
        fail("row_starts of SIMD group " + std::to_string(g));
    if (o.n_internal < o.n_relevant && n_groups > 0 && row_starts[o.n_internal] != row_starts[n_groups])
      fail("row_starts[n_internal] is not re-based onto the end of the SIMD region");
    for (uint32_t i = o.n_internal; i < o.n_relevant; ++i)
      if (row_starts[i + 1] < row_starts[i])
        fail("row_starts of row " + std::to_string(i));
    const uint64_t end = o.n_relevant > o.n_internal ? row_starts[o.n_relevant] : row_starts[n_groups];
    if (end != nnz)
      fail("nnz does not match row_starts");
    for (uint32_t i = 0; i < o.n_relevant; ++i) {
      const uint32_t len = ref.row_length(i);
      if (len == 0)
        fail("empty row " + std::to_string(i));
      if (columns[ref.scalar_pos(i, 0)] != i)
        fail("row " + std::to_string(i) + " does not start with its diagonal");
      for (uint32_t c = 1; c < len; ++c) {
        const uint32_t j = columns[ref.scalar_pos(i, c)];
        if (j >= o.n_relevant)
          fail("column index out of range in row " + std::to_string(i));
        if (c > 1 && j <= columns[ref.scalar_pos(i, c - 1)])
          fail("columns of row " + std::to_string(i) + " do not ascend");
      }
    }
    for (uint32_t i = 0; i < o.n_relevant; ++i)
      if (!(mi[i] > 0.) || !(mi_inv[i] > 0.))
        fail("non-positive lumped mass at " + std::to_string(i));
    for (uint32_t b = 0; b < o.n_bdry; ++b) {
      if (b_i[b] >= o.n_owned)
        fail("boundary_map index out of the owned range");
      if (b_id[b] > RYUJIN_BC_DIRICHLET_MOMENTUM)
        fail("unknown boundary id");
    }
    for (uint32_t q = 0; q < o.n_pairs; ++q)
      if (p_i[q] >= o.n_owned || p_j[q] >= o.n_relevant || p_col[q] >= ref.row_length(p_i[q]) ||
          columns[ref.scalar_pos(p_i[q], p_col[q])] != p_j[q])
        fail("coupling boundary pair " + std::to_string(q));
    if (o.n_nbr > 0) {
      if (send_off[0] != 0 || recv_off[0] != o.n_owned || recv_off[o.n_nbr] != o.n_relevant ||
          row_send_off[0] != 0)
        fail("exchange offsets");
      for (int p = 0; p < o.n_nbr; ++p)
        if (send_off[p + 1] < send_off[p] || recv_off[p + 1] < recv_off[p] ||
            row_send_off[p + 1] < row_send_off[p] || nbr_rank[p] < 0 ||
            (p > 0 && nbr_rank[p] <= nbr_rank[p - 1]))
          fail("exchange lists of neighbour " + std::to_string(p));
      for (uint32_t i : send_idx)
        if (i >= o.n_owned)
          fail("send index out of the owned range");
      for (size_t e = 0; e < row_send_row.size(); ++e)
        if (row_send_row[e] >= o.n_owned || row_send_col[e] >= ref.row_length(row_send_row[e]))
          fail("matrix send list entry " + std::to_string(e));
    } else if (o.n_owned != o.n_relevant) {
      fail("ghost rows without neighbours");
    }
  }

  void bind()
  {
    ryujin_hip_offline &o = view;
    o.row_starts = row_starts.data();
    o.columns = columns.data();
    o.cij = cij.data();
    o.mij = mij.data();
    o.mi = mi.data();
    o.mi_inv = mi_inv.data();
    o.b_i = b_i.data();
    o.b_normal = b_normal.data();
    o.b_id = b_id.data();
    o.p_i = p_i.data();
    o.p_col = p_col.data();
    o.p_j = p_j.data();
    o.initial_precomputed = n_init_prec ? initial_precomputed.data() : nullptr;
    o.nbr_rank = nbr_rank.data();
    o.send_off = o.n_nbr ? send_off.data() : nullptr;
    o.send_idx = send_idx.data();
    o.recv_off = o.n_nbr ? recv_off.data() : nullptr;
    o.row_send_off = o.n_nbr ? row_send_off.data() : nullptr;
    o.row_send_row = row_send_row.data();
    o.row_send_col = row_send_col.data();
    o.incidence = o.discontinuous_ansatz ? incidence.data() : nullptr;
    o.mass_matrix_inverse = o.discontinuous_ansatz ? mass_matrix_inverse.data() : nullptr;
  }
};

extern "C" {

const char *ryujin_offline_io_last_error(void)
{
  return g_error.c_str();
}

int ryujin_offline_write(const char *path, const ryujin_hip_offline *o, int dim, int n_init_prec,
                         const double *positions, const double *b_positions)
{
  if (!path || !o || dim < 1 || dim > 3 || n_init_prec < 0 || !o->row_starts) {
    g_error = "ryujin_offline_write: bad argument";
    return RYUJIN_ERR_ARG;
  }
  FILE *f = std::fopen(path, "wb");
  if (!f) {
    g_error = std::string("ryujin_offline_write: cannot open ") + path;
    return RYUJIN_ERR_ARG;
  }
  try {
    Writer w{f, {}};
    const uint32_t sl = o->simd_length ? o->simd_length : 1;
    const uint64_t nnz =
        o->n_relevant > o->n_internal ? o->row_starts[o->n_relevant] : o->row_starts[o->n_internal / sl];
    const uint32_t n_nbr = (uint32_t)o->n_nbr;
    if (n_init_prec && !o->initial_precomputed)
      throw std::invalid_argument("initial_precomputed is NULL");
    w.raw(kMagic, 8);
    w.scalar<uint32_t>(kVersion);
    w.scalar<uint32_t>((uint32_t)dim);
    w.scalar<uint32_t>((uint32_t)n_init_prec);
    const bool dg = o->discontinuous_ansatz != 0;
    if (dg && (!o->incidence || !o->mass_matrix_inverse))
      throw std::invalid_argument("discontinuous ansatz without incidence / inverse mass matrix");
    w.scalar<uint32_t>((positions ? 1u : 0u) | (b_positions ? 2u : 0u) | (dg ? 4u : 0u));
    for (uint32_t v : {o->n_export, o->n_internal, o->n_owned, o->n_relevant, sl, o->n_bdry, o->n_pairs, n_nbr})
      w.scalar<uint32_t>(v);
    w.scalar<uint64_t>(nnz);
    w.scalar<double>(o->measure_of_omega);
    const uint64_t n = o->n_relevant;
    w.section(o->row_starts, (n + 1) * 8);
    w.section(o->columns, nnz * 4);
    w.section(o->cij, nnz * dim * 8);
    w.section(o->mij, nnz * 8);
    w.section(o->mi, n * 8);
    w.section(o->mi_inv, n * 8);
    w.section(o->b_i, (uint64_t)o->n_bdry * 4);
    w.section(o->b_normal, (uint64_t)o->n_bdry * dim * 8);
    w.section(o->b_id, (uint64_t)o->n_bdry);
    w.section(o->p_i, (uint64_t)o->n_pairs * 4);
    w.section(o->p_col, (uint64_t)o->n_pairs * 4);
    w.section(o->p_j, (uint64_t)o->n_pairs * 4);
    w.section(o->initial_precomputed, n * n_init_prec * 8);
    const uint64_t n_send = n_nbr ? o->send_off[n_nbr] : 0;
    const uint64_t n_row_send = n_nbr ? o->row_send_off[n_nbr] : 0;
    w.section(o->nbr_rank, (uint64_t)n_nbr * 4);
    w.section(o->send_off, n_nbr ? (uint64_t)(n_nbr + 1) * 4 : 0);
    w.section(o->send_idx, n_send * 4);
    w.section(o->recv_off, n_nbr ? (uint64_t)(n_nbr + 1) * 4 : 0);
    w.section(o->row_send_off, n_nbr ? (uint64_t)(n_nbr + 1) * 4 : 0);
    w.section(o->row_send_row, n_row_send * 4);
    w.section(o->row_send_col, n_row_send * 4);
    w.section(positions, positions ? n * dim * 8 : 0);
    w.section(b_positions, b_positions ? (uint64_t)o->n_bdry * dim * 8 : 0);
    if (dg) { /* flag bit 2: two more sections */
      w.section(o->incidence, nnz * 8);
      w.section(o->mass_matrix_inverse, nnz * 8);
    }
    const uint64_t h = w.sum.h;
    if (std::fwrite(&h, 8, 1, f) != 1)
      throw std::runtime_error("short write");
    if (std::fclose(f) != 0) {
      f = nullptr;
      throw std::runtime_error("close failed");
    }
    return RYUJIN_OK;
  } catch (const std::exception &e) {
    if (f)
      std::fclose(f);
    std::remove(path);
    g_error = std::string("ryujin_offline_write: ") + e.what();
    return RYUJIN_ERR_ARG;
  }
}

ryujin_offline_file *ryujin_offline_read(const char *path)
{
  FILE *f = path ? std::fopen(path, "rb") : nullptr;
  if (!f) {
    g_error = std::string("ryujin_offline_read: cannot open ") + (path ? path : "(null)");
    return nullptr;
  }
  try {
    auto file = std::make_unique<ryujin_offline_file>();
    Reader r{f, {}};
    char magic[8];
    r.raw(magic, 8);
    if (std::memcmp(magic, kMagic, 8) != 0)
      throw std::runtime_error("not an OfflineData dump (bad magic)");
    if (r.scalar<uint32_t>() != kVersion)
      throw std::runtime_error("unsupported version");
    const uint32_t dim = r.scalar<uint32_t>();
    const uint32_t nip = r.scalar<uint32_t>();
    const uint32_t flags = r.scalar<uint32_t>();
    if (dim < 1 || dim > 3 || nip > 16 || flags > 7)
      throw std::runtime_error("bad header");
    ryujin_hip_offline &o = file->view;
    o.n_export = r.scalar<uint32_t>();
    o.n_internal = r.scalar<uint32_t>();
    o.n_owned = r.scalar<uint32_t>();
    o.n_relevant = r.scalar<uint32_t>();
    o.simd_length = r.scalar<uint32_t>();
    o.n_bdry = r.scalar<uint32_t>();
    o.n_pairs = r.scalar<uint32_t>();
    const uint32_t n_nbr = r.scalar<uint32_t>();
    if (n_nbr > (1u << 20) || o.simd_length == 0)
      throw std::runtime_error("bad header");
    o.n_nbr = (int)n_nbr;
    const uint64_t nnz = r.scalar<uint64_t>();
    o.measure_of_omega = r.scalar<double>();
    file->dim = (int)dim;
    file->n_init_prec = (int)nip;
    file->nnz = nnz;
    /* the file length bounds every section before anything is allocated */
    const long here = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const uint64_t file_bytes = (uint64_t)std::ftell(f);
    std::fseek(f, here, SEEK_SET);
    if (nnz > file_bytes / 4 || (uint64_t)o.n_relevant > file_bytes / 8)
      throw std::runtime_error("header counts exceed the file size");
    const uint64_t n = o.n_relevant;
    r.section(file->row_starts, n + 1, "row_starts");
    r.section(file->columns, nnz, "columns");
    r.section(file->cij, nnz * dim, "cij");
    r.section(file->mij, nnz, "mij");
    r.section(file->mi, n, "mi");
    r.section(file->mi_inv, n, "mi_inv");
    r.section(file->b_i, o.n_bdry, "b_i");
    r.section(file->b_normal, (uint64_t)o.n_bdry * dim, "b_normal");
    r.section(file->b_id, o.n_bdry, "b_id");
    r.section(file->p_i, o.n_pairs, "p_i");
    r.section(file->p_col, o.n_pairs, "p_col");
    r.section(file->p_j, o.n_pairs, "p_j");
    r.section(file->initial_precomputed, n * nip, "initial_precomputed");
    r.section(file->nbr_rank, n_nbr, "nbr_rank");
    r.section(file->send_off, n_nbr ? n_nbr + 1 : 0, "send_off");
    r.section(file->send_idx, n_nbr ? file->send_off[n_nbr] : 0, "send_idx");
    r.section(file->recv_off, n_nbr ? n_nbr + 1 : 0, "recv_off");
    r.section(file->row_send_off, n_nbr ? n_nbr + 1 : 0, "row_send_off");
    r.section(file->row_send_row, n_nbr ? file->row_send_off[n_nbr] : 0, "row_send_row");
    r.section(file->row_send_col, n_nbr ? file->row_send_off[n_nbr] : 0, "row_send_col");
    file->has_positions = flags & 1u;
    file->has_b_positions = flags & 2u;
    r.section(file->positions, file->has_positions ? n * dim : 0, "positions");
    r.section(file->b_positions, file->has_b_positions ? (uint64_t)o.n_bdry * dim : 0, "b_positions");
    o.discontinuous_ansatz = (flags & 4u) ? 1 : 0;
    if (o.discontinuous_ansatz) {
      r.section(file->incidence, nnz, "incidence");
      r.section(file->mass_matrix_inverse, nnz, "mass_matrix_inverse");
    }
    const uint64_t expected = r.sum.h;
    uint64_t stored = 0;
    if (std::fread(&stored, 8, 1, f) != 1)
      throw std::runtime_error("truncated file (checksum missing)");
    if (stored != expected)
      throw std::runtime_error("checksum mismatch");
    char extra;
    if (std::fread(&extra, 1, 1, f) == 1)
      throw std::runtime_error("trailing bytes after the checksum");
    std::fclose(f);
    f = nullptr;
    file->bind();
    file->validate();
    return file.release();
  } catch (const std::exception &e) {
    if (f)
      std::fclose(f);
    g_error = std::string("ryujin_offline_read: ") + e.what();
    return nullptr;
  }
}

void ryujin_offline_file_free(ryujin_offline_file *f)
{
  delete f;
}

const ryujin_hip_offline *ryujin_offline_file_view(const ryujin_offline_file *f)
{
  return &f->view;
}
int ryujin_offline_file_dim(const ryujin_offline_file *f)
{
  return f->dim;
}
int ryujin_offline_file_n_initial_precomputed(const ryujin_offline_file *f)
{
  return f->n_init_prec;
}
uint64_t ryujin_offline_file_nnz(const ryujin_offline_file *f)
{
  return f->nnz;
}
const double *ryujin_offline_file_positions(const ryujin_offline_file *f)
{
  return f->has_positions ? f->positions.data() : nullptr;
}
const double *ryujin_offline_file_b_positions(const ryujin_offline_file *f)
{
  return f->has_b_positions ? f->b_positions.data() : nullptr;
}

} // extern "C"
