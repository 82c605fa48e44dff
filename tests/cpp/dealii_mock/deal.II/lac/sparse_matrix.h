#pragma once
#include <deal.II/lac/dynamic_sparsity_pattern.h>
#include <deal.II/lac/vector.h>
namespace dealii
{
  template <typename Number>
  class SparseMatrix : public Subscriptor
  {
  public:
    using size_type = types::global_dof_index;
    using value_type = Number;
    SparseMatrix();
    explicit SparseMatrix(const SparsityPattern &);
    void reinit(const SparsityPattern &);
    SparseMatrix &operator=(const double);
    SparseMatrix &operator*=(const Number);
    size_type m() const;
    size_type n() const;
    void set(const size_type, const size_type, const Number);
    void add(const size_type, const size_type, const Number);
    template <typename N2> void add(const Number, const SparseMatrix<N2> &);
    Number operator()(const size_type, const size_type) const;
    Number el(const size_type, const size_type) const;
    Number diag_element(const size_type) const;
    template <class Out, class In> void vmult(Out &, const In &) const;
    template <class Out, class In> void Tvmult(Out &, const In &) const;
    template <class Out, class In> void vmult_add(Out &, const In &) const;
    const SparsityPattern &get_sparsity_pattern() const;
    template <typename M> void copy_from(const M &);
    struct const_iterator {
      struct Accessor { size_type row() const; size_type column() const; Number value() const; size_type global_index() const; };
      const Accessor *operator->() const;
      const_iterator &operator++();
      bool operator!=(const const_iterator &) const;
    };
    using iterator = const_iterator;
    const_iterator begin() const;
    const_iterator end() const;
    const_iterator begin(const size_type) const;
    const_iterator end(const size_type) const;
  };
}
