#pragma once
#include <deal.II/base/index_set.h>
#include <deal.II/base/subscriptor.h>
#include <utility>
#include <vector>
namespace dealii
{
  template <typename Number = double>
  class AffineConstraints : public Subscriptor
  {
  public:
    using size_type = types::global_dof_index;
    AffineConstraints() = default;
    explicit AffineConstraints(const IndexSet &);
    void reinit(const IndexSet &);
    void clear();
    void close();
    void add_line(const size_type);
    void add_entry(const size_type, const size_type, const Number);
    void set_inhomogeneity(const size_type, const Number);
    bool is_constrained(const size_type) const;
    bool can_store_line(const size_type) const;
    size_type n_constraints() const;
    const std::vector<std::pair<size_type, Number>> *get_constraint_entries(const size_type) const;
    template <class V> void distribute(V &) const;
    template <class V> void set_zero(V &) const;
    void merge(const AffineConstraints &);
    template <typename M, typename... A> void distribute_local_to_global(const M &, A &&...) const;
    template <typename S> void add_entries_local_to_global(const std::vector<size_type> &, S &, const bool = true) const;
    struct ConstraintLine { size_type index; std::vector<std::pair<size_type, Number>> entries; Number inhomogeneity; };
    const std::vector<ConstraintLine> &get_lines() const;
  };
}
