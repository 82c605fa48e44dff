#pragma once
#include <deal.II/base/index_set.h>
#include <deal.II/base/mpi.h>
#include <utility>
#include <vector>
namespace dealii
{
  namespace Utilities
  {
    namespace MPI
    {
      class Partitioner
      {
      public:
        Partitioner();
        Partitioner(const IndexSet &locally_owned, const IndexSet &ghost, const MPI_Comm communicator);
        Partitioner(const IndexSet &locally_owned, const MPI_Comm communicator);
        types::global_dof_index size() const;
        unsigned int locally_owned_size() const;
        unsigned int local_size() const;
        const IndexSet &locally_owned_range() const;
        std::pair<types::global_dof_index, types::global_dof_index> local_range() const;
        bool in_local_range(types::global_dof_index) const;
        unsigned int global_to_local(types::global_dof_index) const;
        types::global_dof_index local_to_global(unsigned int) const;
        bool is_ghost_entry(types::global_dof_index) const;
        const IndexSet &ghost_indices() const;
        unsigned int n_ghost_indices() const;
        const std::vector<std::pair<unsigned int, unsigned int>> &ghost_targets() const;
        const std::vector<std::pair<unsigned int, unsigned int>> &import_indices() const;
        unsigned int n_import_indices() const;
        const std::vector<std::pair<unsigned int, unsigned int>> &import_targets() const;
        unsigned int this_mpi_process() const;
        unsigned int n_mpi_processes() const;
        const MPI_Comm &get_mpi_communicator() const;
        void set_ghost_indices(const IndexSet &, const IndexSet & = IndexSet());
        void set_owned_indices(const IndexSet &);
        bool is_compatible(const Partitioner &) const;
      };
    }
  }
}
