#pragma once
#include <deal.II/base/mpi.h>
#include <deal.II/base/point.h>
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  template <int dim, int spacedim = dim>
  class Triangulation : public Subscriptor
  {
  public:
    enum MeshSmoothing { none = 0, limit_level_difference_at_vertices = 1, eliminate_unrefined_islands = 2, maximum_smoothing = 0xffff };
    virtual ~Triangulation() = default;
    unsigned int n_levels() const;
    unsigned int n_global_levels() const;
    unsigned int n_active_cells() const;
    types::global_cell_index n_global_active_cells() const;
    void refine_global(const unsigned int times = 1);
    void clear();
    virtual MPI_Comm get_communicator() const;
    struct cell_accessor;
    struct active_cell_iterator {
      cell_accessor *operator->() const;
      cell_accessor &operator*() const;
      active_cell_iterator &operator++();
      bool operator!=(const active_cell_iterator &) const;
      bool operator==(const active_cell_iterator &) const;
    };
    using cell_iterator = active_cell_iterator;
    struct cell_accessor {
      bool is_locally_owned() const;
      bool is_ghost() const;
      bool is_artificial() const;
      Point<spacedim> center() const;
      double diameter() const;
      double measure() const;
      Point<spacedim> &vertex(const unsigned int) const;
      unsigned int n_vertices() const;
      unsigned int n_faces() const;
      void set_refine_flag() const;
      void set_coarsen_flag() const;
      void set_material_id(types::material_id) const;
      void set_manifold_id(types::manifold_id) const;
      void set_all_manifold_ids(types::manifold_id) const;
      types::material_id material_id() const;
      unsigned int level() const;
      unsigned int index() const;
      active_cell_iterator neighbor(unsigned int) const;
      bool at_boundary() const;
      bool at_boundary(unsigned int) const;
    };
    struct IteratorRange { active_cell_iterator begin() const; active_cell_iterator end() const; };
    IteratorRange active_cell_iterators() const;
    IteratorRange cell_iterators() const;
    active_cell_iterator begin_active(unsigned int = 0) const;
    active_cell_iterator end() const;
  };
}
