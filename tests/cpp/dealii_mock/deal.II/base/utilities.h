#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/exceptions.h>
#include <string>
#include <vector>
namespace dealii
{
  namespace Utilities
  {
    template <int N, typename T>
    inline T fixed_power(const T t)
    {
      T r = T(1.);
      for (int i = 0; i < (N < 0 ? -N : N); ++i)
        r *= t;
      return N < 0 ? T(1.) / r : r;
    }
    std::string int_to_string(unsigned int value, unsigned int digits = numbers::invalid_unsigned_int);
    template <typename T> std::string to_string(const T, unsigned int digits = numbers::invalid_unsigned_int);
    std::string trim(const std::string &);
    std::vector<std::string> split_string_list(const std::string &, const std::string &delimiter = ",");
    std::vector<std::string> split_string_list(const std::string &, char delimiter);
    double string_to_double(const std::string &);
    int string_to_int(const std::string &);
    template <typename T> std::size_t pack(const T &, std::vector<char> &, bool = true);
    template <typename T> std::vector<char> pack(const T &, bool = true);
    template <typename T> T unpack(const std::vector<char> &, bool = true);
    namespace System
    {
      std::string get_hostname();
      std::string get_time();
      std::string get_date();
      struct MemoryStats { unsigned long VmPeak, VmSize, VmHWM, VmRSS; };
      void get_memory_stats(MemoryStats &);
    }
  }
}
