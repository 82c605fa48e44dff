#pragma once
#include <deal.II/base/config.h>
#include <exception>
#include <string>
namespace dealii
{
  class ExceptionBase : public std::exception
  {
  public:
    const char *what() const noexcept override { return message.empty() ? "dealii mock exception" : message.c_str(); }
    std::string message;
  };
  struct ExcMessage : ExceptionBase { explicit ExcMessage(const std::string &m) { message = m; } };
  struct ExcInternalError : ExceptionBase {};
  struct ExcNotInitialized : ExceptionBase {};
  struct ExcNotImplemented : ExceptionBase {};
  struct ExcIO : ExceptionBase {};
  struct ExcFileNotOpen : ExceptionBase { explicit ExcFileNotOpen(const std::string &) {} };
  struct ExcDimensionMismatch : ExceptionBase { ExcDimensionMismatch(std::size_t, std::size_t) {} };
  template <typename T> struct ExcIndexRangeType : ExceptionBase { ExcIndexRangeType(T, T, T) {} };
  namespace StandardExceptions
  {
    using dealii::ExcMessage;
    using dealii::ExcInternalError;
    using dealii::ExcNotImplemented;
    using dealii::ExcNotInitialized;
  }
}
#define Assert(cond, exc) do { if (false) { if (!(cond)) throw exc; } } while (false)
#define AssertNothrow(cond, exc) do { if (false) { (void)(cond); } } while (false)
#define AssertThrow(cond, exc) do { if (!(cond)) throw exc; } while (false)
#define AssertDimension(a, b) Assert((a) == (b), dealii::ExcDimensionMismatch((a), (b)))
#define AssertIndexRange(i, n) Assert((i) < (n), dealii::ExcInternalError())
#define AssertThrowMPI(ierr) do { (void)(ierr); } while (false)
#define AssertIsFinite(x) do { (void)(x); } while (false)
