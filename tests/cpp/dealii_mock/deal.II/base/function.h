#pragma once
#include <deal.II/base/exceptions.h>
#include <deal.II/base/point.h>
#include <deal.II/base/subscriptor.h>
#include <deal.II/lac/vector.h>
#include <map>
#include <string>
#include <vector>
namespace dealii
{
  template <int dim, typename RangeNumberType = double>
  class Function : public Subscriptor
  {
  public:
    static constexpr unsigned int dimension = dim;
    const unsigned int n_components;
    explicit Function(const unsigned int n_components = 1, const RangeNumberType initial_time = 0.0)
        : n_components(n_components), time(initial_time) {}
    virtual ~Function() = default;
    virtual RangeNumberType value(const Point<dim> &, const unsigned int = 0) const { return RangeNumberType(); }
    virtual void vector_value(const Point<dim> &, Vector<RangeNumberType> &) const {}
    virtual void value_list(const std::vector<Point<dim>> &, std::vector<RangeNumberType> &, const unsigned int = 0) const {}
    virtual Tensor<1, dim, RangeNumberType> gradient(const Point<dim> &, const unsigned int = 0) const { return {}; }
    RangeNumberType get_time() const { return time; }
    virtual void set_time(const RangeNumberType t) { time = t; }
  private:
    RangeNumberType time;
  };
  namespace Functions
  {
    template <int dim, typename RangeNumberType = double>
    class ZeroFunction : public Function<dim, RangeNumberType>
    {
    public:
      explicit ZeroFunction(const unsigned int n_components = 1) : Function<dim, RangeNumberType>(n_components) {}
    };
    template <int dim, typename RangeNumberType = double>
    class ConstantFunction : public Function<dim, RangeNumberType>
    {
    public:
      explicit ConstantFunction(const RangeNumberType, const unsigned int n_components = 1) : Function<dim, RangeNumberType>(n_components) {}
    };
  }
  template <int dim>
  class FunctionParser : public Function<dim, double>
  {
  public:
    explicit FunctionParser(const unsigned int n_components = 1, const double initial_time = 0.0, const double h = 1e-8);
    FunctionParser(const std::string &expression, const std::string &constants = "", const std::string &variable_names = default_variable_names() + ",t", const double h = 1e-8);
    using ConstMap = std::map<std::string, double>;
    void initialize(const std::string &vars, const std::vector<std::string> &expressions, const ConstMap &constants, const bool time_dependent = false);
    void initialize(const std::string &vars, const std::string &expression, const ConstMap &constants, const bool time_dependent = false);
    static std::string default_variable_names();
    double value(const Point<dim> &, const unsigned int = 0) const override;
  };
}
