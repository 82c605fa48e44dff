#pragma once
#include <deal.II/lac/la_parallel_vector.h>
#include <vector>
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
        BlockVector() = default;
        explicit BlockVector(const unsigned int n_blocks) : blocks_(n_blocks) {}
        void reinit(const unsigned int n_blocks, const size_type = 0, const bool = false) { blocks_.resize(n_blocks); }
        void reinit(const BlockVector &o, const bool = false) { blocks_.resize(o.blocks_.size()); }
        BlockType &block(const unsigned int i) { return blocks_[i]; }
        const BlockType &block(const unsigned int i) const { return blocks_[i]; }
        unsigned int n_blocks() const { return blocks_.size(); }
        void collect_sizes() {}
        void update_ghost_values() const {}
        void zero_out_ghost_values() const {}
        void compress(VectorOperation::values) {}
        BlockVector &operator=(const Number s) { for (auto &b : blocks_) b = s; return *this; }
        BlockVector &operator=(const BlockVector &o) { blocks_ = o.blocks_; return *this; }
        void swap(BlockVector &o) { blocks_.swap(o.blocks_); }
        void sadd(const Number s, const Number a, const BlockVector &o) { for (std::size_t i = 0; i < blocks_.size(); ++i) blocks_[i].sadd(s, a, o.blocks_[i]); }
        Number l2_norm() const { Number s = 0; for (const auto &b : blocks_) s += b * b; return std::sqrt(s); }
        Number linfty_norm() const { Number s = 0; for (const auto &b : blocks_) s = std::max(s, b.linfty_norm()); return s; }
      private:
        std::vector<BlockType> blocks_;
      };
    }
  }
}
