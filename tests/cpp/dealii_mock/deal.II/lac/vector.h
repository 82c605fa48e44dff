#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/subscriptor.h>
#include <vector>
namespace dealii
{
  template <typename Number>
  class Vector : public Subscriptor
  {
  public:
    using value_type = Number;
    using size_type = std::size_t;
    Vector() = default;
    explicit Vector(const size_type n) : v(n) {}
    void reinit(const size_type n, const bool = false) { v.assign(n, Number()); }
    size_type size() const { return v.size(); }
    Number &operator()(const size_type i) { return v[i]; }
    Number operator()(const size_type i) const { return v[i]; }
    Number &operator[](const size_type i) { return v[i]; }
    Number operator[](const size_type i) const { return v[i]; }
    Vector &operator=(const Number s) { v.assign(v.size(), s); return *this; }
    Number *begin() { return v.data(); }
    Number *end() { return v.data() + v.size(); }
    const Number *begin() const { return v.data(); }
    const Number *end() const { return v.data() + v.size(); }
    Number l2_norm() const;
    Number linfty_norm() const;
  private:
    std::vector<Number> v;
  };
}
