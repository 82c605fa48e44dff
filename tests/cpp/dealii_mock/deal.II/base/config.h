// deal.II stand-in for TYPE CHECKS of the drop-in boundary only (tests/test_binding_compile.py): declarations with
// the names and signatures ryujin's headers use, trivial or no behaviour. NOT an oracle, NOT a reference build:
// nothing compiled against these headers produces a number any test compares.
#pragma once
#define DEAL_II_NAMESPACE_OPEN namespace dealii {
#define DEAL_II_NAMESPACE_CLOSE }
#define DEAL_II_ALWAYS_INLINE __attribute__((always_inline))
#define DEAL_II_OPENMP_SIMD_PRAGMA
#define DEAL_II_VERSION_MAJOR 9
#define DEAL_II_VERSION_MINOR 5
#define DEAL_II_VERSION_GTE(a, b, c) ((9 * 10000 + 5 * 100 + 0) >= ((a) * 10000 + (b) * 100 + (c)))
#define DEAL_II_COMPILER_VECTORIZATION_LEVEL 0
#define DEAL_II_WITH_MPI
#define DEAL_II_DEPRECATED
#include <cstddef>
namespace dealii
{
  namespace numbers
  {
    static const unsigned int invalid_unsigned_int = static_cast<unsigned int>(-1);
    static const double PI = 3.14159265358979323846;
  }
  namespace types
  {
    using global_dof_index = unsigned int;
    using boundary_id = unsigned int;
    using manifold_id = unsigned int;
    using material_id = unsigned int;
    using subdomain_id = unsigned int;
    using global_cell_index = unsigned long;
  }
}
