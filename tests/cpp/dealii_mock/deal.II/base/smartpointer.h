#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/subscriptor.h>
namespace dealii
{
  template <typename T, typename P = void>
  class SmartPointer
  {
  public:
    SmartPointer() : p(nullptr) {}
    SmartPointer(T *q) : p(q) {}
    SmartPointer(T *q, const char *) : p(q) {}
    SmartPointer &operator=(T *q) { p = q; return *this; }
    operator T *() const { return p; }
    T &operator*() const { return *p; }
    T *operator->() const { return p; }
    T *get() const { return p; }
  private:
    T *p;
  };
}
