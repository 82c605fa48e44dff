#pragma once
#include <deal.II/base/index_set.h>
#include <deal.II/base/subscriptor.h>
#include <algorithm>
#include <vector>
namespace dealii
{
  class SparsityPatternBase : public Subscriptor
  {
  public:
    using size_type = types::global_dof_index;
  };

  namespace mock_detail
  {
    /* one iterator type for both patterns: a position in a flat column array */
    struct SparsityIterator {
      using size_type = types::global_dof_index;
      struct Accessor {
        const size_type *columns;
        std::size_t position;
        size_type row_;
        size_type row() const { return row_; }
        size_type column() const { return columns[position]; }
        size_type index() const { return position; }
        std::size_t global_index() const { return position; }
      };
      Accessor a;
      const Accessor *operator->() const { return &a; }
      const Accessor &operator*() const { return a; }
      SparsityIterator &operator++() { ++a.position; return *this; }
      SparsityIterator operator++(int) { SparsityIterator r(*this); ++a.position; return r; }
      bool operator!=(const SparsityIterator &o) const { return a.position != o.a.position; }
      bool operator==(const SparsityIterator &o) const { return a.position == o.a.position; }
      int operator-(const SparsityIterator &o) const { return int(a.position) - int(o.a.position); }
    };
  }

  /* rows of ascending column indices */
  class DynamicSparsityPattern : public SparsityPatternBase
  {
  public:
    DynamicSparsityPattern() = default;
    DynamicSparsityPattern(const size_type m, const size_type n, const IndexSet & = IndexSet()) { reinit(m, n); }
    explicit DynamicSparsityPattern(const IndexSet &s) { reinit(s.size(), s.size()); }
    void reinit(const size_type m, const size_type n, const IndexSet & = IndexSet())
    {
      rows_.assign(m, {});
      n_cols_ = n;
      flat_valid_ = false;
    }
    void add(const size_type i, const size_type j)
    {
      auto &row = rows_[i];
      const auto it = std::lower_bound(row.begin(), row.end(), j);
      if (it == row.end() || *it != j)
        row.insert(it, j);
      flat_valid_ = false;
    }
    template <typename It> void add_entries(const size_type i, It begin, It end, const bool = false)
    {
      for (It it = begin; it != end; ++it)
        add(i, *it);
    }
    bool exists(const size_type i, const size_type j) const { return std::binary_search(rows_[i].begin(), rows_[i].end(), j); }
    size_type n_rows() const { return rows_.size(); }
    size_type n_cols() const { return n_cols_; }
    size_type row_length(const size_type i) const { return rows_[i].size(); }
    size_type column_number(const size_type i, const size_type k) const { return rows_[i][k]; }
    size_type max_entries_per_row() const { size_type m = 0; for (const auto &r : rows_) m = std::max<size_type>(m, r.size()); return m; }
    size_type n_nonzero_elements() const { size_type n = 0; for (const auto &r : rows_) n += r.size(); return n; }
    void symmetrize();
    void compress() {}
    const IndexSet &row_index_set() const;
    using iterator = mock_detail::SparsityIterator;
    using const_iterator = iterator;
    iterator begin(const size_type i) const { flatten(); return {{flat_.data(), start_[i], i}}; }
    iterator end(const size_type i) const { flatten(); return {{flat_.data(), start_[i + 1], i}}; }
    iterator begin() const { return begin(0); }
    iterator end() const { flatten(); return {{flat_.data(), start_.back(), n_rows()}}; }
  private:
    void flatten() const
    {
      if (flat_valid_)
        return;
      start_.assign(rows_.size() + 1, 0);
      flat_.clear();
      for (std::size_t i = 0; i < rows_.size(); ++i) {
        flat_.insert(flat_.end(), rows_[i].begin(), rows_[i].end());
        start_[i + 1] = flat_.size();
      }
      flat_valid_ = true;
    }
    std::vector<std::vector<size_type>> rows_;
    size_type n_cols_ = 0;
    mutable std::vector<size_type> flat_;
    mutable std::vector<std::size_t> start_;
    mutable bool flat_valid_ = false;
  };

  /* compressed rows; square patterns store the diagonal entry FIRST and the rest ascending (deal.II's convention,
   * relied on at hyperbolic_module.template.h:394-396) */
  class SparsityPattern : public SparsityPatternBase
  {
  public:
    SparsityPattern() = default;
    void copy_from(const DynamicSparsityPattern &dsp)
    {
      const size_type m = dsp.n_rows();
      n_cols_ = dsp.n_cols();
      start_.assign(m + 1, 0);
      columns_.clear();
      for (size_type i = 0; i < m; ++i) {
        const bool square = m == n_cols_;
        if (square)
          columns_.push_back(i); /* deal.II adds the diagonal of a square pattern if it is missing */
        for (size_type k = 0; k < dsp.row_length(i); ++k)
          if (!square || dsp.column_number(i, k) != i)
            columns_.push_back(dsp.column_number(i, k));
        start_[i + 1] = columns_.size();
      }
    }
    void reinit(const size_type, const size_type, const unsigned int);
    void compress() {}
    size_type n_rows() const { return start_.size() - 1; }
    size_type n_cols() const { return n_cols_; }
    std::size_t n_nonzero_elements() const { return columns_.size(); }
    unsigned int row_length(const size_type i) const { return start_[i + 1] - start_[i]; }
    size_type column_number(const size_type i, const unsigned int k) const { return columns_[start_[i] + k]; }
    /* global index of entry (i, j) */
    std::size_t operator()(const size_type i, const size_type j) const
    {
      for (std::size_t e = start_[i]; e < start_[i + 1]; ++e)
        if (columns_[e] == j)
          return e;
      return numbers::invalid_unsigned_int;
    }
    using iterator = mock_detail::SparsityIterator;
    using const_iterator = iterator;
    iterator begin(const size_type i) const { return {{columns_.data(), start_[i], i}}; }
    iterator end(const size_type i) const { return {{columns_.data(), start_[i + 1], i}}; }
    iterator begin() const { return begin(0); }
    iterator end() const { return {{columns_.data(), start_.back(), n_rows()}}; }
  private:
    std::vector<size_type> columns_;
    std::vector<std::size_t> start_ = {0};
    size_type n_cols_ = 0;
  };
}
