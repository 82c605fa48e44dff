#pragma once
#include <deal.II/grid/tria.h>
namespace dealii
{
  namespace parallel
  {
    template <int dim, int spacedim = dim>
    class TriangulationBase : public dealii::Triangulation<dim, spacedim>
    {
    public:
      MPI_Comm get_communicator() const override;
      types::subdomain_id locally_owned_subdomain() const;
    };
    template <int dim, int spacedim = dim>
    class DistributedTriangulationBase : public TriangulationBase<dim, spacedim> {};
    namespace distributed
    {
      template <int dim, int spacedim = dim>
      class Triangulation : public DistributedTriangulationBase<dim, spacedim>
      {
      public:
        enum Settings { default_setting = 0, mesh_reconstruction_after_repartitioning = 1, construct_multigrid_hierarchy = 2, no_automatic_repartitioning = 4 };
        explicit Triangulation(const MPI_Comm, const typename dealii::Triangulation<dim, spacedim>::MeshSmoothing = dealii::Triangulation<dim, spacedim>::none, const Settings = default_setting);
        void execute_coarsening_and_refinement();
        void repartition();
        void save(const std::string &) const;
        void load(const std::string &);
      };
    }
  }
}
