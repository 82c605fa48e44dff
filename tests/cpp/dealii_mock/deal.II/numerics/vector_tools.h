#pragma once
#include <deal.II/base/function.h>
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/fe/mapping.h>
namespace dealii
{
  namespace VectorTools
  {
    template <int dim, int spacedim, typename V>
    void interpolate(const Mapping<dim, spacedim> &, const DoFHandler<dim, spacedim> &, const Function<spacedim, typename V::value_type> &, V &);
    template <int dim, int spacedim, typename V>
    void interpolate(const DoFHandler<dim, spacedim> &, const Function<spacedim, typename V::value_type> &, V &);
  }
}
