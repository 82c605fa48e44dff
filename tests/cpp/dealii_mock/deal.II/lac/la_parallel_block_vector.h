#pragma once
#include <deal.II/lac/la_parallel_vector.h>
namespace dealii
{
  namespace LinearAlgebra
  {
    namespace distributed
    {
      template <typename Number>
      class BlockVector : public Subscriptor
      {
      public:
        using BlockType = Vector<Number>;
        using value_type = Number;
        using size_type = types::global_dof_index;
        BlockVector();
        explicit BlockVector(const unsigned int n_blocks);
        void reinit(const unsigned int n_blocks, const size_type = 0, const bool = false);
        void reinit(const BlockVector &, const bool = false);
        BlockType &block(const unsigned int);
        const BlockType &block(const unsigned int) const;
        unsigned int n_blocks() const;
        void collect_sizes();
        void update_ghost_values() const;
        void zero_out_ghost_values() const;
        void compress(VectorOperation::values);
        BlockVector &operator=(const Number);
        BlockVector &operator=(const BlockVector &);
        void swap(BlockVector &);
        void sadd(const Number, const Number, const BlockVector &);
        Number l2_norm() const;
        Number linfty_norm() const;
      };
    }
  }
}
