#pragma once
#include <deal.II/base/config.h>
#include <vector>
namespace dealii
{
  class IndexSet
  {
  public:
    using size_type = types::global_dof_index;
    IndexSet() = default;
    explicit IndexSet(size_type) {}
    size_type size() const;
    size_type n_elements() const;
    bool is_element(size_type) const;
    void add_index(size_type);
    void add_range(size_type, size_type);
    template <typename It> void add_indices(It, It);
    void compress() const;
    void set_size(size_type);
    size_type nth_index_in_set(size_type) const;
    size_type index_within_set(size_type) const;
    void subtract_set(const IndexSet &);
    struct ElementIterator {
      size_type operator*() const;
      ElementIterator &operator++();
      bool operator!=(const ElementIterator &) const;
    };
    ElementIterator begin() const;
    ElementIterator end() const;
    std::vector<size_type> get_index_vector() const;
  };
  IndexSet complete_index_set(IndexSet::size_type);
}
