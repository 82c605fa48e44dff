#pragma once
#include <deal.II/base/point.h>
#include <deal.II/base/subscriptor.h>
#include <vector>
namespace dealii
{
  template <int dim>
  class Quadrature : public Subscriptor
  {
  public:
    explicit Quadrature(const unsigned int n_quadrature_points = 0);
    unsigned int size() const;
    const Point<dim> &point(const unsigned int) const;
    const std::vector<Point<dim>> &get_points() const;
    double weight(const unsigned int) const;
  };
  template <int dim> class QGauss : public Quadrature<dim> { public: explicit QGauss(const unsigned int n); };
  template <int dim> class QGaussSimplex : public Quadrature<dim> { public: explicit QGaussSimplex(const unsigned int n); };
}
