#pragma once
#include <deal.II/base/config.h>
#include <algorithm>
#include <utility>
#include <vector>
namespace dealii
{
  /* sorted, disjoint half-open ranges; enough for the partitioners of one MPI rank (tests/cpp/time_integrator_run.cc) */
  class IndexSet
  {
  public:
    using size_type = types::global_dof_index;
    IndexSet() = default;
    explicit IndexSet(size_type n) : size_(n) {}
    size_type size() const { return size_; }
    void set_size(size_type n) { size_ = n; }
    size_type n_elements() const
    {
      size_type n = 0;
      for (const auto &r : ranges_)
        n += r.second - r.first;
      return n;
    }
    bool is_element(size_type i) const
    {
      for (const auto &r : ranges_)
        if (i >= r.first && i < r.second)
          return true;
      return false;
    }
    void add_index(size_type i) { add_range(i, i + 1); }
    void add_range(size_type begin, size_type end)
    {
      if (begin >= end)
        return;
      ranges_.emplace_back(begin, end);
      std::sort(ranges_.begin(), ranges_.end());
      std::vector<std::pair<size_type, size_type>> merged;
      for (const auto &r : ranges_) {
        if (!merged.empty() && r.first <= merged.back().second)
          merged.back().second = std::max(merged.back().second, r.second);
        else
          merged.push_back(r);
      }
      ranges_.swap(merged);
    }
    template <typename It> void add_indices(It begin, It end)
    {
      for (It it = begin; it != end; ++it)
        add_index(*it);
    }
    void compress() const {}
    size_type nth_index_in_set(size_type n) const
    {
      for (const auto &r : ranges_) {
        if (n < r.second - r.first)
          return r.first + n;
        n -= r.second - r.first;
      }
      return numbers::invalid_unsigned_int;
    }
    size_type index_within_set(size_type i) const
    {
      size_type n = 0;
      for (const auto &r : ranges_) {
        if (i >= r.first && i < r.second)
          return n + (i - r.first);
        n += r.second - r.first;
      }
      return numbers::invalid_unsigned_int;
    }
    void subtract_set(const IndexSet &);
    struct ElementIterator {
      const IndexSet *set = nullptr;
      std::size_t range = 0;
      size_type index = 0;
      size_type operator*() const { return index; }
      ElementIterator &operator++()
      {
        if (++index >= set->ranges_[range].second) {
          ++range;
          index = range < set->ranges_.size() ? set->ranges_[range].first : 0;
        }
        return *this;
      }
      bool operator!=(const ElementIterator &o) const { return range != o.range || index != o.index; }
    };
    ElementIterator begin() const { return {this, 0, ranges_.empty() ? 0 : ranges_[0].first}; }
    ElementIterator end() const { return {this, ranges_.size(), 0}; }
    /* intervals (multicomponent_vector.cc walks them) */
    struct IntervalAccessor {
      const IndexSet *set;
      std::size_t range;
      ElementIterator begin() const { return {set, range, set->ranges_[range].first}; }
      size_type last() const { return set->ranges_[range].second - 1; }
      size_type n_elements() const { return set->ranges_[range].second - set->ranges_[range].first; }
    };
    struct IntervalIterator {
      IntervalAccessor a;
      const IntervalAccessor *operator->() const { return &a; }
      const IntervalAccessor &operator*() const { return a; }
      IntervalIterator &operator++() { ++a.range; return *this; }
      bool operator!=(const IntervalIterator &o) const { return a.range != o.a.range; }
      bool operator==(const IntervalIterator &o) const { return a.range == o.a.range; }
    };
    IntervalIterator begin_intervals() const { return {{this, 0}}; }
    IntervalIterator end_intervals() const { return {{this, ranges_.size()}}; }
    std::vector<size_type> get_index_vector() const
    {
      std::vector<size_type> v;
      for (const auto &r : ranges_)
        for (size_type i = r.first; i < r.second; ++i)
          v.push_back(i);
      return v;
    }
  private:
    size_type size_ = 0;
    std::vector<std::pair<size_type, size_type>> ranges_;
  };
  inline IndexSet complete_index_set(IndexSet::size_type n)
  {
    IndexSet s(n);
    s.add_range(0, n);
    return s;
  }
}
