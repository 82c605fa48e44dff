#pragma once
#include <deal.II/base/mpi.h>
#include <string>
namespace dealii
{
  class Timer
  {
  public:
    Timer();
    Timer(const MPI_Comm, const bool sync_lap_times = false);
    void start();
    double stop();
    void reset();
    void restart();
    double wall_time() const;
    double cpu_time() const;
    double last_wall_time() const;
    double last_cpu_time() const;
  };
  class TimerOutput
  {
  public:
    class Scope
    {
    public:
      Scope(TimerOutput &, const std::string &);
    };
  };
}
