#pragma once
#include <deal.II/base/point.h>
#include <deal.II/base/subscriptor.h>
#include <string>
#include <vector>
namespace dealii
{
  template <int dim, int spacedim = dim>
  class FiniteElement : public Subscriptor
  {
  public:
    virtual ~FiniteElement() = default;
    unsigned int n_dofs_per_cell() const;
    unsigned int dofs_per_cell;
    unsigned int n_components() const;
    unsigned int degree;
    unsigned int tensor_degree() const;
    virtual std::string get_name() const;
    const std::vector<Point<dim>> &get_unit_support_points() const;
    bool has_support_points() const;
  };
}
