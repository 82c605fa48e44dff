#pragma once
#include <deal.II/base/config.h>
namespace dealii
{
  class Subscriptor
  {
  public:
    virtual ~Subscriptor() = default;
  };
}
