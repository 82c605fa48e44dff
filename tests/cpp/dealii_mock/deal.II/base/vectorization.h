#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/exceptions.h>
#include <algorithm>
#include <array>
#include <cmath>
namespace dealii
{
  /* the scalar "SIMD" type: one lane (DEAL_II_COMPILER_VECTORIZATION_LEVEL 0) */
  template <typename Number, std::size_t width = 1>
  class VectorizedArray
  {
  public:
    using value_type = Number;
    static constexpr std::size_t size() { return width; }
    VectorizedArray() = default;
    VectorizedArray(const Number s) { data.fill(s); }
    VectorizedArray &operator=(const Number s) { data.fill(s); return *this; }
    Number &operator[](const unsigned int i) { return data[i]; }
    const Number &operator[](const unsigned int i) const { return data[i]; }
    VectorizedArray &operator+=(const VectorizedArray &o) { for (std::size_t i = 0; i < width; ++i) data[i] += o.data[i]; return *this; }
    VectorizedArray &operator-=(const VectorizedArray &o) { for (std::size_t i = 0; i < width; ++i) data[i] -= o.data[i]; return *this; }
    VectorizedArray &operator*=(const VectorizedArray &o) { for (std::size_t i = 0; i < width; ++i) data[i] *= o.data[i]; return *this; }
    VectorizedArray &operator/=(const VectorizedArray &o) { for (std::size_t i = 0; i < width; ++i) data[i] /= o.data[i]; return *this; }
    void load(const Number *p) { for (std::size_t i = 0; i < width; ++i) data[i] = p[i]; }
    void store(Number *p) const { for (std::size_t i = 0; i < width; ++i) p[i] = data[i]; }
    void streaming_store(Number *p) const { store(p); }
    void gather(const Number *base, const unsigned int *offsets) { for (std::size_t i = 0; i < width; ++i) data[i] = base[offsets[i]]; }
    void scatter(const unsigned int *offsets, Number *base) const { for (std::size_t i = 0; i < width; ++i) base[offsets[i]] = data[i]; }
    std::array<Number, width> data;
  };
#define RYUJIN_MOCK_VA_BINOP(op)                                                                              \
  template <typename N, std::size_t w>                                                                        \
  inline VectorizedArray<N, w> operator op(const VectorizedArray<N, w> &a, const VectorizedArray<N, w> &b)    \
  { VectorizedArray<N, w> r(a); r op## = b; return r; }                                                       \
  template <typename N, std::size_t w>                                                                        \
  inline VectorizedArray<N, w> operator op(const N &a, const VectorizedArray<N, w> &b)                        \
  { VectorizedArray<N, w> r(a); r op## = b; return r; }                                                       \
  template <typename N, std::size_t w>                                                                        \
  inline VectorizedArray<N, w> operator op(const VectorizedArray<N, w> &a, const N &b)                        \
  { VectorizedArray<N, w> r(a); r op## = VectorizedArray<N, w>(b); return r; }
  RYUJIN_MOCK_VA_BINOP(+)
  RYUJIN_MOCK_VA_BINOP(-)
  RYUJIN_MOCK_VA_BINOP(*)
  RYUJIN_MOCK_VA_BINOP(/)
#undef RYUJIN_MOCK_VA_BINOP
  template <typename N, std::size_t w> inline VectorizedArray<N, w> operator-(const VectorizedArray<N, w> &a) { return N(0) - a; }
  template <typename N, std::size_t w> inline VectorizedArray<N, w> operator+(const VectorizedArray<N, w> &a) { return a; }
  template <typename N, std::size_t w> inline bool operator==(const VectorizedArray<N, w> &a, const VectorizedArray<N, w> &b) { return a.data == b.data; }
  template <typename N, std::size_t w> inline bool operator!=(const VectorizedArray<N, w> &a, const VectorizedArray<N, w> &b) { return !(a == b); }

  enum class SIMDComparison : int { equal, not_equal, less_than, less_than_or_equal, greater_than, greater_than_or_equal };
  template <SIMDComparison predicate, typename Number>
  inline Number compare_and_apply_mask(const Number &left, const Number &right, const Number &true_value, const Number &false_value)
  {
    bool mask;
    switch (predicate) {
    case SIMDComparison::equal: mask = (left == right); break;
    case SIMDComparison::not_equal: mask = (left != right); break;
    case SIMDComparison::less_than: mask = (left < right); break;
    case SIMDComparison::less_than_or_equal: mask = (left <= right); break;
    case SIMDComparison::greater_than: mask = (left > right); break;
    default: mask = (left >= right); break;
    }
    return mask ? true_value : false_value;
  }
  template <SIMDComparison predicate, typename N, std::size_t w>
  inline VectorizedArray<N, w> compare_and_apply_mask(const VectorizedArray<N, w> &l, const VectorizedArray<N, w> &r,
                                                      const VectorizedArray<N, w> &t, const VectorizedArray<N, w> &f)
  {
    VectorizedArray<N, w> out;
    for (std::size_t i = 0; i < w; ++i)
      out[i] = compare_and_apply_mask<predicate, N>(l[i], r[i], t[i], f[i]);
    return out;
  }
  template <typename N, std::size_t w>
  void vectorized_load_and_transpose(const unsigned int n_entries, const N *in, const unsigned int *offsets, VectorizedArray<N, w> *out);
  template <typename N, std::size_t w>
  void vectorized_transpose_and_store(const bool add_into, const unsigned int n_entries, const VectorizedArray<N, w> *in,
                                      const unsigned int *offsets, N *out);
  template <typename N, std::size_t w>
  void vectorized_load_and_transpose(const unsigned int n_entries, const std::array<N *, w> &in, VectorizedArray<N, w> *out);
  template <typename N, std::size_t w>
  void vectorized_transpose_and_store(const bool add_into, const unsigned int n_entries, const VectorizedArray<N, w> *in,
                                      std::array<N *, w> &out);
  namespace internal
  {
    template <typename T> struct VectorizedArrayTrait { using value_type = T; static constexpr std::size_t width() { return 1; } };
    template <typename T, std::size_t w> struct VectorizedArrayTrait<VectorizedArray<T, w>> { using value_type = T; static constexpr std::size_t width() { return w; } };
  }
}
namespace std
{
#define RYUJIN_MOCK_VA_FN(fn)                                                                     \
  template <typename N, size_t w> inline dealii::VectorizedArray<N, w> fn(const dealii::VectorizedArray<N, w> &x) \
  { dealii::VectorizedArray<N, w> r; for (size_t i = 0; i < w; ++i) r[i] = std::fn(x[i]); return r; }
  RYUJIN_MOCK_VA_FN(sqrt) RYUJIN_MOCK_VA_FN(abs) RYUJIN_MOCK_VA_FN(exp) RYUJIN_MOCK_VA_FN(log) RYUJIN_MOCK_VA_FN(sin)
  RYUJIN_MOCK_VA_FN(cos) RYUJIN_MOCK_VA_FN(tan) RYUJIN_MOCK_VA_FN(floor) RYUJIN_MOCK_VA_FN(ceil)
#undef RYUJIN_MOCK_VA_FN
  template <typename N, size_t w> inline dealii::VectorizedArray<N, w> max(const dealii::VectorizedArray<N, w> &a, const dealii::VectorizedArray<N, w> &b)
  { dealii::VectorizedArray<N, w> r; for (size_t i = 0; i < w; ++i) r[i] = std::max(a[i], b[i]); return r; }
  template <typename N, size_t w> inline dealii::VectorizedArray<N, w> min(const dealii::VectorizedArray<N, w> &a, const dealii::VectorizedArray<N, w> &b)
  { dealii::VectorizedArray<N, w> r; for (size_t i = 0; i < w; ++i) r[i] = std::min(a[i], b[i]); return r; }
  template <typename N, size_t w> inline dealii::VectorizedArray<N, w> pow(const dealii::VectorizedArray<N, w> &a, const N b)
  { dealii::VectorizedArray<N, w> r; for (size_t i = 0; i < w; ++i) r[i] = std::pow(a[i], b); return r; }
  template <typename N, size_t w> inline dealii::VectorizedArray<N, w> pow(const dealii::VectorizedArray<N, w> &a, const dealii::VectorizedArray<N, w> &b)
  { dealii::VectorizedArray<N, w> r; for (size_t i = 0; i < w; ++i) r[i] = std::pow(a[i], b[i]); return r; }
}
