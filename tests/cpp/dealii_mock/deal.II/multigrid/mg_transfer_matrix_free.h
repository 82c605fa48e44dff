#pragma once
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  template <int dim, typename Number>
  class MGTransferMatrixFree : public Subscriptor
  {
  public:
    MGTransferMatrixFree();
  };
  template <typename T> class MGLevelObject
  {
  public:
    MGLevelObject(unsigned int = 0, unsigned int = 0);
    T &operator[](unsigned int);
    const T &operator[](unsigned int) const;
    void resize(unsigned int, unsigned int);
    unsigned int min_level() const;
    unsigned int max_level() const;
  };
}
