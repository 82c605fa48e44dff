#pragma once
#include <deal.II/base/index_set.h>
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  class SparsityPatternBase : public Subscriptor
  {
  public:
    using size_type = types::global_dof_index;
  };
  class DynamicSparsityPattern : public SparsityPatternBase
  {
  public:
    DynamicSparsityPattern();
    DynamicSparsityPattern(const size_type, const size_type, const IndexSet & = IndexSet());
    explicit DynamicSparsityPattern(const IndexSet &);
    void reinit(const size_type, const size_type, const IndexSet & = IndexSet());
    void add(const size_type, const size_type);
    template <typename It> void add_entries(const size_type, It, It, const bool = false);
    bool exists(const size_type, const size_type) const;
    size_type n_rows() const;
    size_type n_cols() const;
    size_type row_length(const size_type) const;
    size_type column_number(const size_type, const size_type) const;
    size_type max_entries_per_row() const;
    size_type n_nonzero_elements() const;
    void symmetrize();
    void compress();
    const IndexSet &row_index_set() const;
    struct iterator {
      struct Accessor { size_type row() const; size_type column() const; size_type index() const; };
      const Accessor *operator->() const;
      const Accessor &operator*() const;
      iterator &operator++();
      bool operator!=(const iterator &) const;
      bool operator==(const iterator &) const;
      int operator-(const iterator &) const;
    };
    using const_iterator = iterator;
    iterator begin() const;
    iterator end() const;
    iterator begin(const size_type) const;
    iterator end(const size_type) const;
  };
  class SparsityPattern : public SparsityPatternBase
  {
  public:
    SparsityPattern();
    void copy_from(const DynamicSparsityPattern &);
    void reinit(const size_type, const size_type, const unsigned int);
    void compress();
    size_type n_rows() const;
    size_type n_cols() const;
    size_type n_nonzero_elements() const;
    unsigned int row_length(const size_type) const;
    size_type column_number(const size_type, const unsigned int) const;
    using iterator = DynamicSparsityPattern::iterator;
    using const_iterator = iterator;
    iterator begin() const;
    iterator end() const;
    iterator begin(const size_type) const;
    iterator end(const size_type) const;
  };
}
