#pragma once
#include <deal.II/base/config.h>
#include <vector>
namespace dealii
{
  /* pointer iterators, as deal.II's (sparse_matrix_simd.template.h compares end() with a raw pointer) */
  template <typename T>
  class AlignedVector
  {
  public:
    using value_type = T;
    using iterator = T *;
    using const_iterator = const T *;
    using size_type = std::size_t;
    AlignedVector() = default;
    explicit AlignedVector(const size_type n, const T &init = T()) : v(n, init) {}
    void resize_fast(const size_type n) { v.resize(n); }
    void resize(const size_type n) { v.resize(n); }
    void resize(const size_type n, const T &init) { v.resize(n, init); }
    void reserve(const size_type n) { v.reserve(n); }
    void clear() { v.clear(); }
    void push_back(const T &t) { v.push_back(t); }
    void fill(const T &t) { v.assign(v.size(), t); }
    void swap(AlignedVector &o) { v.swap(o.v); }
    bool empty() const { return v.empty(); }
    size_type size() const { return v.size(); }
    T *data() { return v.data(); }
    const T *data() const { return v.data(); }
    T &operator[](const size_type i) { return v[i]; }
    const T &operator[](const size_type i) const { return v[i]; }
    T &back() { return v.back(); }
    const T &back() const { return v.back(); }
    iterator begin() { return v.data(); }
    iterator end() { return v.data() + v.size(); }
    const_iterator begin() const { return v.data(); }
    const_iterator end() const { return v.data() + v.size(); }
    size_type memory_consumption() const { return v.size() * sizeof(T); }
  private:
    std::vector<T> v;
  };
}
