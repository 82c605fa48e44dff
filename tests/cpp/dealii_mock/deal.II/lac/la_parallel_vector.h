#pragma once
#include <deal.II/base/partitioner.h>
#include <deal.II/base/subscriptor.h>
#include <deal.II/lac/vector_operation.h>
#include <memory>
namespace dealii
{
  namespace MemorySpace { struct Host {}; }
  namespace LinearAlgebra
  {
    namespace distributed
    {
      template <typename Number, typename MemorySpaceType = MemorySpace::Host>
      class Vector : public Subscriptor
      {
      public:
        using value_type = Number;
        using size_type = types::global_dof_index;
        using real_type = Number;
        using iterator = Number *;
        using const_iterator = const Number *;
        Vector();
        Vector(const Vector &);
        explicit Vector(const std::shared_ptr<const Utilities::MPI::Partitioner> &);
        void reinit(const std::shared_ptr<const Utilities::MPI::Partitioner> &, const MPI_Comm & = MPI_COMM_SELF);
        void reinit(const Vector &, const bool omit_zeroing_entries = false);
        void reinit(const IndexSet &, const IndexSet &, const MPI_Comm);
        Vector &operator=(const Vector &);
        Vector &operator=(const Number);
        Vector &operator+=(const Vector &);
        Vector &operator-=(const Vector &);
        Vector &operator*=(const Number);
        Vector &operator/=(const Number);
        void swap(Vector &);
        void sadd(const Number, const Number, const Vector &);
        void add(const Number, const Vector &);
        void add(const Number, const Vector &, const Number, const Vector &);
        void equ(const Number, const Vector &);
        void scale(const Vector &);
        Number operator*(const Vector &) const;
        Number l1_norm() const;
        Number l2_norm() const;
        Number linfty_norm() const;
        Number mean_value() const;
        size_type size() const;
        size_type locally_owned_size() const;
        size_type local_size() const;
        IndexSet locally_owned_elements() const;
        Number &local_element(const size_type);
        Number local_element(const size_type) const;
        Number &operator()(const size_type);
        Number operator()(const size_type) const;
        Number &operator[](const size_type);
        Number operator[](const size_type) const;
        iterator begin();
        const_iterator begin() const;
        iterator end();
        const_iterator end() const;
        Number *get_values() const;
        void compress(VectorOperation::values);
        void update_ghost_values() const;
        void update_ghost_values_start(const unsigned int communication_channel = 0) const;
        void update_ghost_values_finish() const;
        void compress_start(const unsigned int communication_channel = 0, VectorOperation::values = VectorOperation::add);
        void compress_finish(VectorOperation::values);
        void zero_out_ghost_values() const;
        void zero_out_ghosts() const;
        bool has_ghost_elements() const;
        const std::shared_ptr<const Utilities::MPI::Partitioner> &get_partitioner() const;
        const MPI_Comm &get_mpi_communicator() const;
        bool partitioners_are_compatible(const Utilities::MPI::Partitioner &) const;
        std::size_t memory_consumption() const;
        template <class Archive> void serialize(Archive &, const unsigned int) {}
        template <class Archive> void save(Archive &, const unsigned int) const {}
        template <class Archive> void load(Archive &, const unsigned int) {}
      };
    }
  }
}
