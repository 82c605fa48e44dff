#pragma once
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  template <class MeshType>
  class InterGridMap : public Subscriptor
  {
  public:
    void make_mapping(const MeshType &, const MeshType &);
  };
}
