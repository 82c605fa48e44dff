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
      /* ONE MPI rank: owned range + (normally empty) ghost set, no communication pattern */
      class Partitioner
      {
      public:
        Partitioner() = default;
        Partitioner(const IndexSet &locally_owned, const IndexSet &ghost, const MPI_Comm communicator)
            : owned_(locally_owned), ghost_(ghost), comm_(communicator) {}
        Partitioner(const IndexSet &locally_owned, const MPI_Comm communicator)
            : owned_(locally_owned), ghost_(locally_owned.size()), comm_(communicator) {}
        types::global_dof_index size() const { return owned_.size(); }
        unsigned int locally_owned_size() const { return owned_.n_elements(); }
        unsigned int local_size() const { return locally_owned_size(); }
        const IndexSet &locally_owned_range() const { return owned_; }
        std::pair<types::global_dof_index, types::global_dof_index> local_range() const
        {
          const auto first = owned_.n_elements() ? owned_.nth_index_in_set(0) : 0;
          return {first, first + owned_.n_elements()};
        }
        bool in_local_range(types::global_dof_index i) const { return owned_.is_element(i); }
        unsigned int global_to_local(types::global_dof_index i) const
        {
          if (owned_.is_element(i))
            return owned_.index_within_set(i);
          return locally_owned_size() + ghost_.index_within_set(i);
        }
        types::global_dof_index local_to_global(unsigned int i) const
        {
          const unsigned int n = locally_owned_size();
          return i < n ? owned_.nth_index_in_set(i) : ghost_.nth_index_in_set(i - n);
        }
        bool is_ghost_entry(types::global_dof_index i) const { return ghost_.is_element(i); }
        const IndexSet &ghost_indices() const { return ghost_; }
        unsigned int n_ghost_indices() const { return ghost_.n_elements(); }
        const std::vector<std::pair<unsigned int, unsigned int>> &ghost_targets() const { return none_; }
        const std::vector<std::pair<unsigned int, unsigned int>> &import_indices() const { return none_; }
        unsigned int n_import_indices() const { return 0; }
        const std::vector<std::pair<unsigned int, unsigned int>> &import_targets() const { return none_; }
        unsigned int this_mpi_process() const { return 0; }
        unsigned int n_mpi_processes() const { return 1; }
        const MPI_Comm &get_mpi_communicator() const { return comm_; }
        void set_ghost_indices(const IndexSet &g, const IndexSet & = IndexSet()) { ghost_ = g; }
        void set_owned_indices(const IndexSet &o) { owned_ = o; }
        bool is_compatible(const Partitioner &o) const
        {
          return locally_owned_size() == o.locally_owned_size() && n_ghost_indices() == o.n_ghost_indices();
        }
      private:
        IndexSet owned_, ghost_;
        MPI_Comm comm_ = MPI_COMM_SELF;
        std::vector<std::pair<unsigned int, unsigned int>> none_;
      };
    }
  }
}
