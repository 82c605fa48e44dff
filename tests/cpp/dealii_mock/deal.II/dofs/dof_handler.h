#pragma once
#include <deal.II/base/index_set.h>
#include <deal.II/fe/fe.h>
#include <deal.II/grid/tria.h>
#include <vector>
namespace dealii
{
  template <int dim, int spacedim = dim>
  class DoFHandler : public Subscriptor
  {
  public:
    DoFHandler();
    explicit DoFHandler(const Triangulation<dim, spacedim> &);
    void reinit(const Triangulation<dim, spacedim> &);
    void distribute_dofs(const FiniteElement<dim, spacedim> &);
    void distribute_mg_dofs();
    void clear();
    types::global_dof_index n_dofs() const;
    types::global_dof_index n_locally_owned_dofs() const;
    const IndexSet &locally_owned_dofs() const;
    const FiniteElement<dim, spacedim> &get_fe(const unsigned int = 0) const;
    const Triangulation<dim, spacedim> &get_triangulation() const;
    MPI_Comm get_communicator() const;
    void renumber_dofs(const std::vector<types::global_dof_index> &);
    struct cell_accessor : Triangulation<dim, spacedim>::cell_accessor {
      void get_dof_indices(std::vector<types::global_dof_index> &) const;
      void get_active_or_mg_dof_indices(std::vector<types::global_dof_index> &) const;
    };
    struct active_cell_iterator {
      cell_accessor *operator->() const;
      cell_accessor &operator*() const;
      active_cell_iterator &operator++();
      bool operator!=(const active_cell_iterator &) const;
      bool operator==(const active_cell_iterator &) const;
    };
    using cell_iterator = active_cell_iterator;
    struct IteratorRange { active_cell_iterator begin() const; active_cell_iterator end() const; };
    IteratorRange active_cell_iterators() const;
    active_cell_iterator begin_active(unsigned int = 0) const;
    active_cell_iterator end() const;
  };
}
