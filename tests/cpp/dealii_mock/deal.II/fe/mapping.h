#pragma once
#include <deal.II/base/point.h>
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  template <int dim, int spacedim = dim>
  class Mapping : public Subscriptor
  {
  public:
    virtual ~Mapping() = default;
  };
}
