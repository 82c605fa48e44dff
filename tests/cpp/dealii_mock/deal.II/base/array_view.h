#pragma once
#include <deal.II/base/config.h>
#include <vector>
namespace dealii
{
  template <typename ElementType>
  class ArrayView
  {
  public:
    using value_type = ElementType;
    using iterator = ElementType *;
    ArrayView() : p(nullptr), n(0) {}
    ArrayView(ElementType *starting_element, const std::size_t n_elements) : p(starting_element), n(n_elements) {}
    template <typename O> ArrayView(const ArrayView<O> &o) : p(o.data()), n(o.size()) {}
    template <typename O> ArrayView(std::vector<O> &v) : p(v.data()), n(v.size()) {}
    template <typename O> ArrayView(const std::vector<O> &v) : p(v.data()), n(v.size()) {}
    std::size_t size() const { return n; }
    ElementType *data() const { return p; }
    iterator begin() const { return p; }
    iterator end() const { return p + n; }
    ElementType &operator[](const std::size_t i) const { return p[i]; }
  private:
    ElementType *p;
    std::size_t n;
  };
  template <typename T> inline ArrayView<T> make_array_view(std::vector<T> &v) { return ArrayView<T>(v.data(), v.size()); }
  template <typename T> inline ArrayView<const T> make_array_view(const std::vector<T> &v) { return ArrayView<const T>(v.data(), v.size()); }
  template <typename T> inline ArrayView<T> make_array_view(T *p, std::size_t n) { return ArrayView<T>(p, n); }
}
