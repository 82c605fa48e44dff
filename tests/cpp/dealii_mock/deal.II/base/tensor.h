#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/exceptions.h>
#include <cmath>
#include <type_traits>
namespace dealii
{
  template <int rank, int dim, typename Number = double>
  class Tensor;

  template <int dim, typename Number>
  class Tensor<0, dim, Number>
  {
  public:
    using value_type = Number;
    static constexpr unsigned int dimension = dim;
    static constexpr unsigned int rank = 0;
    Tensor() : v() {}
    Tensor(const Number &x) : v(x) {}
    operator Number &() { return v; }
    operator const Number &() const { return v; }
    Number v;
  };

  template <int rank_, int dim, typename Number>
  class Tensor
  {
  public:
    using value_type = typename Tensor<rank_ - 1, dim, Number>::value_type;
    using entry_type = std::conditional_t<rank_ == 1, Number, Tensor<rank_ - 1, dim, Number>>;
    static constexpr unsigned int dimension = dim;
    static constexpr unsigned int rank = rank_;
    static constexpr unsigned int n_independent_components = 1;
    Tensor() : values() {}
    template <typename Other>
    Tensor(const Tensor<rank_, dim, Other> &o)
    {
      for (int i = 0; i < dim; ++i)
        values[i] = o[i];
    }
    entry_type &operator[](const unsigned int i) { return values[i]; }
    const entry_type &operator[](const unsigned int i) const { return values[i]; }
    template <typename Other>
    Tensor &operator=(const Tensor<rank_, dim, Other> &o)
    {
      for (int i = 0; i < dim; ++i)
        values[i] = o[i];
      return *this;
    }
    Tensor &operator=(const Number &d)
    {
      for (int i = 0; i < dim; ++i)
        values[i] = d;
      return *this;
    }
    template <typename Other>
    Tensor &operator+=(const Tensor<rank_, dim, Other> &o)
    {
      for (int i = 0; i < dim; ++i)
        values[i] += o[i];
      return *this;
    }
    template <typename Other>
    Tensor &operator-=(const Tensor<rank_, dim, Other> &o)
    {
      for (int i = 0; i < dim; ++i)
        values[i] -= o[i];
      return *this;
    }
    template <typename Other>
    Tensor &operator*=(const Other &f)
    {
      for (int i = 0; i < dim; ++i)
        values[i] *= f;
      return *this;
    }
    template <typename Other>
    Tensor &operator/=(const Other &f)
    {
      for (int i = 0; i < dim; ++i)
        values[i] /= f;
      return *this;
    }
    Tensor operator-() const
    {
      Tensor t;
      for (int i = 0; i < dim; ++i)
        t.values[i] = -values[i];
      return t;
    }
    bool operator==(const Tensor &o) const
    {
      for (int i = 0; i < dim; ++i)
        if (!(values[i] == o.values[i]))
          return false;
      return true;
    }
    bool operator!=(const Tensor &o) const { return !(*this == o); }
    Number norm() const
    {
      using std::sqrt;
      return sqrt(norm_square());
    }
    Number norm_square() const
    {
      Number s = Number();
      if constexpr (rank_ == 1) {
        for (int i = 0; i < dim; ++i)
          s += values[i] * values[i];
      } else {
        for (int i = 0; i < dim; ++i)
          s += values[i].norm_square();
      }
      return s;
    }
    void clear()
    {
      for (int i = 0; i < dim; ++i)
        values[i] = entry_type();
    }
    Number *begin_raw() { return reinterpret_cast<Number *>(values); }
    const Number *begin_raw() const { return reinterpret_cast<const Number *>(values); }
    template <class Archive> void serialize(Archive &, const unsigned int) {}

  private:
    entry_type values[dim > 0 ? dim : 1];
  };

  template <int rank, int dim, typename N, typename O>
  inline Tensor<rank, dim, decltype(N() + O())> operator+(const Tensor<rank, dim, N> &a, const Tensor<rank, dim, O> &b)
  {
    Tensor<rank, dim, decltype(N() + O())> t(a);
    t += b;
    return t;
  }
  template <int rank, int dim, typename N, typename O>
  inline Tensor<rank, dim, decltype(N() - O())> operator-(const Tensor<rank, dim, N> &a, const Tensor<rank, dim, O> &b)
  {
    Tensor<rank, dim, decltype(N() - O())> t(a);
    t -= b;
    return t;
  }
  template <int rank, int dim, typename N, typename O,
            typename = std::enable_if_t<std::is_arithmetic<O>::value || std::is_same<O, N>::value>>
  inline Tensor<rank, dim, N> operator*(const Tensor<rank, dim, N> &a, const O &f)
  {
    Tensor<rank, dim, N> t(a);
    t *= f;
    return t;
  }
  template <int rank, int dim, typename N, typename O,
            typename = std::enable_if_t<std::is_arithmetic<O>::value || std::is_same<O, N>::value>>
  inline Tensor<rank, dim, N> operator*(const O &f, const Tensor<rank, dim, N> &a)
  {
    return a * f;
  }
  template <int rank, int dim, typename N, typename O,
            typename = std::enable_if_t<std::is_arithmetic<O>::value || std::is_same<O, N>::value>>
  inline Tensor<rank, dim, N> operator/(const Tensor<rank, dim, N> &a, const O &f)
  {
    Tensor<rank, dim, N> t(a);
    t /= f;
    return t;
  }
  /* single contraction */
  template <int dim, typename N, typename O>
  inline decltype(N() * O()) operator*(const Tensor<1, dim, N> &a, const Tensor<1, dim, O> &b)
  {
    decltype(N() * O()) s = decltype(N() * O())();
    for (int i = 0; i < dim; ++i)
      s += a[i] * b[i];
    return s;
  }
  template <int dim, typename N, typename O>
  inline Tensor<1, dim, decltype(N() * O())> operator*(const Tensor<2, dim, N> &a, const Tensor<1, dim, O> &b)
  {
    Tensor<1, dim, decltype(N() * O())> r;
    for (int i = 0; i < dim; ++i)
      r[i] = a[i] * b;
    return r;
  }
  template <int rank_1, int rank_2, int dim, typename N, typename O>
  inline Tensor<rank_1 + rank_2 - 2, dim, decltype(N() * O())> contract(const Tensor<rank_1, dim, N> &, const Tensor<rank_2, dim, O> &);
  template <int dim, typename N, typename O>
  inline Tensor<2, dim, decltype(N() * O())> outer_product(const Tensor<1, dim, N> &a, const Tensor<1, dim, O> &b)
  {
    Tensor<2, dim, decltype(N() * O())> r;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j)
        r[i][j] = a[i] * b[j];
    return r;
  }
  template <int dim, typename N, typename O>
  inline decltype(N() * O()) scalar_product(const Tensor<2, dim, N> &a, const Tensor<2, dim, O> &b)
  {
    decltype(N() * O()) s = decltype(N() * O())();
    for (int i = 0; i < dim; ++i)
      s += a[i] * b[i];
    return s;
  }
  template <int dim, typename N>
  inline N trace(const Tensor<2, dim, N> &a)
  {
    N s = N();
    for (int i = 0; i < dim; ++i)
      s += a[i][i];
    return s;
  }
  template <int dim, typename N> Tensor<2, dim, N> transpose(const Tensor<2, dim, N> &);
  template <int dim, typename N> Tensor<1, dim, N> cross_product_2d(const Tensor<1, dim, N> &);
  template <int dim, typename N> Tensor<1, dim, N> cross_product_3d(const Tensor<1, dim, N> &, const Tensor<1, dim, N> &);
}
