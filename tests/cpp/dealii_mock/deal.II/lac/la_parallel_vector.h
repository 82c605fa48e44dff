#pragma once
#include <deal.II/base/partitioner.h>
#include <deal.II/base/subscriptor.h>
#include <deal.II/lac/vector_operation.h>
#include <algorithm>
#include <cmath>
#include <memory>
namespace dealii
{
  namespace MemorySpace { struct Host {}; }
  namespace LinearAlgebra
  {
    namespace distributed
    {
      /* owned + ghost entries in one heap array, as deal.II's; one MPI rank: the ghost exchanges are no-ops.
       * swap() exchanges the storage (what ryujin's TimeIntegrator relies on, time_integrator.template.h:296). */
      template <typename Number, typename MemorySpaceType = MemorySpace::Host>
      class Vector : public Subscriptor
      {
      public:
        using value_type = Number;
        using size_type = types::global_dof_index;
        using real_type = Number;
        using iterator = Number *;
        using const_iterator = const Number *;
        Vector() = default;
        Vector(const Vector &o) : Subscriptor() { *this = o; }
        Vector(Vector &&o) noexcept { swap(o); }
        explicit Vector(const std::shared_ptr<const Utilities::MPI::Partitioner> &p) { reinit(p); }
        void reinit(const std::shared_ptr<const Utilities::MPI::Partitioner> &p, const MPI_Comm & = MPI_COMM_SELF)
        {
          partitioner_ = p;
          n_ = p->locally_owned_size() + p->n_ghost_indices();
          values_.reset(new Number[n_ ? n_ : 1]);
          std::fill(values_.get(), values_.get() + n_, Number());
        }
        void reinit(const Vector &o, const bool = false)
        {
          if (o.partitioner_)
            reinit(o.partitioner_);
        }
        void reinit(const IndexSet &, const IndexSet &, const MPI_Comm);
        Vector &operator=(const Vector &o)
        {
          if (this != &o && o.partitioner_) {
            if (n_ != o.n_ || !values_)
              reinit(o.partitioner_);
            std::copy(o.values_.get(), o.values_.get() + n_, values_.get());
          }
          return *this;
        }
        Vector &operator=(Vector &&o) noexcept { swap(o); return *this; }
        Vector &operator=(const Number s) { std::fill(values_.get(), values_.get() + n_, s); return *this; }
        Vector &operator+=(const Vector &o) { for (size_type i = 0; i < owned(); ++i) values_[i] += o.values_[i]; return *this; }
        Vector &operator-=(const Vector &o) { for (size_type i = 0; i < owned(); ++i) values_[i] -= o.values_[i]; return *this; }
        Vector &operator*=(const Number s) { for (size_type i = 0; i < owned(); ++i) values_[i] *= s; return *this; }
        Vector &operator/=(const Number s) { for (size_type i = 0; i < owned(); ++i) values_[i] /= s; return *this; }
        void swap(Vector &o)
        {
          std::swap(values_, o.values_);
          std::swap(n_, o.n_);
          std::swap(partitioner_, o.partitioner_);
        }
        /* *this = s * (*this) + a * V on the locally owned range (deal.II: ghosts are not touched) */
        void sadd(const Number s, const Number a, const Vector &V)
        {
          for (size_type i = 0; i < owned(); ++i)
            values_[i] = s * values_[i] + a * V.values_[i];
        }
        void add(const Number a, const Vector &V) { for (size_type i = 0; i < owned(); ++i) values_[i] += a * V.values_[i]; }
        void add(const Number a, const Vector &V, const Number b, const Vector &W)
        {
          for (size_type i = 0; i < owned(); ++i)
            values_[i] += a * V.values_[i] + b * W.values_[i];
        }
        void equ(const Number a, const Vector &V) { for (size_type i = 0; i < owned(); ++i) values_[i] = a * V.values_[i]; }
        void scale(const Vector &V) { for (size_type i = 0; i < owned(); ++i) values_[i] *= V.values_[i]; }
        Number operator*(const Vector &V) const { Number s = 0; for (size_type i = 0; i < owned(); ++i) s += values_[i] * V.values_[i]; return s; }
        Number l1_norm() const { Number s = 0; for (size_type i = 0; i < owned(); ++i) s += std::abs(values_[i]); return s; }
        Number l2_norm() const { return std::sqrt((*this) * (*this)); }
        Number linfty_norm() const { Number s = 0; for (size_type i = 0; i < owned(); ++i) s = std::max(s, std::abs(values_[i])); return s; }
        Number mean_value() const { Number s = 0; for (size_type i = 0; i < owned(); ++i) s += values_[i]; return owned() ? s / owned() : s; }
        size_type size() const { return partitioner_ ? partitioner_->size() : 0; }
        size_type locally_owned_size() const { return owned(); }
        size_type local_size() const { return owned(); }
        IndexSet locally_owned_elements() const { return partitioner_->locally_owned_range(); }
        Number &local_element(const size_type i) { return values_[i]; }
        Number local_element(const size_type i) const { return values_[i]; }
        Number &operator()(const size_type i) { return values_[partitioner_->global_to_local(i)]; }
        Number operator()(const size_type i) const { return values_[partitioner_->global_to_local(i)]; }
        Number &operator[](const size_type i) { return (*this)(i); }
        Number operator[](const size_type i) const { return (*this)(i); }
        iterator begin() { return values_.get(); }
        const_iterator begin() const { return values_.get(); }
        iterator end() { return values_.get() + owned(); }
        const_iterator end() const { return values_.get() + owned(); }
        Number *get_values() const { return values_.get(); }
        void compress(VectorOperation::values) {}
        void update_ghost_values() const {}
        void update_ghost_values_start(const unsigned int = 0) const {}
        void update_ghost_values_finish() const {}
        void compress_start(const unsigned int = 0, VectorOperation::values = VectorOperation::add) {}
        void compress_finish(VectorOperation::values) {}
        void zero_out_ghost_values() const {}
        void zero_out_ghosts() const {}
        bool has_ghost_elements() const { return partitioner_ && partitioner_->n_ghost_indices() != 0; }
        const std::shared_ptr<const Utilities::MPI::Partitioner> &get_partitioner() const { return partitioner_; }
        const MPI_Comm &get_mpi_communicator() const { return partitioner_->get_mpi_communicator(); }
        bool partitioners_are_compatible(const Utilities::MPI::Partitioner &p) const { return partitioner_->is_compatible(p); }
        std::size_t memory_consumption() const { return n_ * sizeof(Number); }
        template <class Archive> void serialize(Archive &, const unsigned int) {}
        template <class Archive> void save(Archive &, const unsigned int) const {}
        template <class Archive> void load(Archive &, const unsigned int) {}
      private:
        size_type owned() const { return partitioner_ ? partitioner_->locally_owned_size() : 0; }
        std::unique_ptr<Number[]> values_;
        size_type n_ = 0;
        std::shared_ptr<const Utilities::MPI::Partitioner> partitioner_;
      };
    }
  }
}
