#pragma once
#include <deal.II/distributed/tria.h>
namespace dealii
{
  namespace parallel
  {
    namespace shared
    {
      template <int dim, int spacedim = dim>
      class Triangulation : public TriangulationBase<dim, spacedim>
      {
      public:
        enum Settings { partition_auto = 0, partition_metis = 1, partition_zorder = 2, partition_zoltan = 3, partition_custom_signal = 4, construct_multigrid_hierarchy = 8 };
        explicit Triangulation(const MPI_Comm, const typename dealii::Triangulation<dim, spacedim>::MeshSmoothing = dealii::Triangulation<dim, spacedim>::none, const bool allow_artificial_cells = false, const Settings = partition_auto);
        void execute_coarsening_and_refinement();
      };
    }
  }
}
