#pragma once
#include <deal.II/base/tensor.h>
namespace dealii
{
  template <int dim, typename Number = double>
  class Point : public Tensor<1, dim, Number>
  {
  public:
    Point() = default;
    explicit Point(const Tensor<1, dim, Number> &t) : Tensor<1, dim, Number>(t) {}
    explicit Point(const Number x) { (*this)[0] = x; }
    Point(const Number x, const Number y) { (*this)[0] = x; (*this)[dim > 1 ? 1 : 0] = y; }
    Point(const Number x, const Number y, const Number z) { (*this)[0] = x; (*this)[dim > 1 ? 1 : 0] = y; (*this)[dim > 2 ? 2 : 0] = z; }
    Number operator()(const unsigned int i) const { return (*this)[i]; }
    Number &operator()(const unsigned int i) { return (*this)[i]; }
    Point operator+(const Tensor<1, dim, Number> &t) const { Point p(*this); p += t; return p; }
    Tensor<1, dim, Number> operator-(const Point &o) const { Tensor<1, dim, Number> t(*this); t -= o; return t; }
    Point operator-(const Tensor<1, dim, Number> &t) const { Point p(*this); p -= t; return p; }
    Point operator-() const { Point p; for (int i = 0; i < dim; ++i) p[i] = -(*this)[i]; return p; }
    template <typename O> Point operator*(const O f) const { Point p(*this); p *= f; return p; }
    template <typename O> Point operator/(const O f) const { Point p(*this); p /= f; return p; }
    Number distance(const Point &o) const { return ((*this) - o).norm(); }
    Number square() const { return this->norm_square(); }
  };
  template <int dim, typename Number, typename O>
  inline Point<dim, Number> operator*(const O f, const Point<dim, Number> &p) { return p * f; }
}
