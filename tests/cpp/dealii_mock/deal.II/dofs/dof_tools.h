#pragma once
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/fe/mapping.h>
#include <map>
namespace dealii
{
  namespace DoFTools
  {
    template <int dim, int spacedim>
    std::map<types::global_dof_index, Point<spacedim>> map_dofs_to_support_points(const Mapping<dim, spacedim> &, const DoFHandler<dim, spacedim> &);
    template <int dim, int spacedim>
    void map_dofs_to_support_points(const Mapping<dim, spacedim> &, const DoFHandler<dim, spacedim> &, std::map<types::global_dof_index, Point<spacedim>> &);
    template <int dim, int spacedim> IndexSet extract_locally_relevant_dofs(const DoFHandler<dim, spacedim> &);
    template <int dim, int spacedim> void extract_locally_relevant_dofs(const DoFHandler<dim, spacedim> &, IndexSet &);
  }
}
