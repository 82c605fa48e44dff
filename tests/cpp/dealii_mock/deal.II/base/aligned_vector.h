#pragma once
#include <deal.II/base/config.h>
#include <vector>
namespace dealii
{
  template <typename T>
  class AlignedVector : public std::vector<T>
  {
  public:
    using std::vector<T>::vector;
    void resize_fast(std::size_t n) { this->resize(n); }
    std::size_t memory_consumption() const { return this->size() * sizeof(T); }
  };
}
