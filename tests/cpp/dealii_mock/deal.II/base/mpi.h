#pragma once
#include <deal.II/base/config.h>
#include <deal.II/base/exceptions.h>
#include <vector>
// a stand-in for <mpi.h>: the handful of names ryujin's headers and the adapter use
using MPI_Comm = int;
using MPI_Request = int;
using MPI_Datatype = int;
using MPI_Op = int;
struct MPI_Status { int MPI_SOURCE, MPI_TAG, MPI_ERROR; };
#define MPI_COMM_WORLD 0
#define MPI_COMM_SELF 1
#define MPI_COMM_NULL (-1)
#define MPI_REQUEST_NULL (-1)
#define MPI_STATUSES_IGNORE (static_cast<MPI_Status *>(nullptr))
#define MPI_STATUS_IGNORE (static_cast<MPI_Status *>(nullptr))
#define MPI_SUCCESS 0
#define MPI_BYTE 1
#define MPI_CHAR 2
#define MPI_INT 3
#define MPI_UNSIGNED 4
#define MPI_DOUBLE 5
#define MPI_FLOAT 6
#define MPI_UNSIGNED_LONG 7
#define MPI_UNSIGNED_LONG_LONG 8
#define MPI_MIN 1
#define MPI_MAX 2
#define MPI_SUM 3
#define MPI_LOR 4
#define MPI_IN_PLACE (static_cast<void *>(nullptr))
inline int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm) { return MPI_SUCCESS; } /* one rank */
int MPI_Barrier(MPI_Comm);
int MPI_Comm_rank(MPI_Comm, int *);
int MPI_Comm_size(MPI_Comm, int *);
int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);
int MPI_Isend(const void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Request *);
int MPI_Irecv(void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Request *);
int MPI_Send(const void *, int, MPI_Datatype, int, int, MPI_Comm);
int MPI_Recv(void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Status *);
int MPI_Waitall(int, MPI_Request *, MPI_Status *);
int MPI_Wait(MPI_Request *, MPI_Status *);
int MPI_Allgather(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
namespace dealii
{
  namespace Utilities
  {
    namespace MPI
    {
      inline unsigned int this_mpi_process(const MPI_Comm) { return 0; } /* one rank */
      inline unsigned int n_mpi_processes(const MPI_Comm) { return 1; }
      template <typename T> T min(const T &, const MPI_Comm);
      template <typename T> T max(const T &, const MPI_Comm);
      template <typename T> T sum(const T &, const MPI_Comm);
      template <typename T> T logical_or(const T &, const MPI_Comm);
      template <typename T> T broadcast(const MPI_Comm, const T &, unsigned int root = 0);
      template <typename T> std::vector<T> all_gather(const MPI_Comm, const T &);
      template <typename T> std::vector<T> gather(const MPI_Comm, const T &, unsigned int root = 0);
      struct MinMaxAvg { double sum, min, max, avg; unsigned int min_index, max_index; };
      MinMaxAvg min_max_avg(double, const MPI_Comm);
      namespace internal
      {
        namespace Tags
        {
          enum enumeration : unsigned int { partitioner_export_start = 200, partitioner_export_end = 400 };
        }
      }
      class MPI_InitFinalize
      {
      public:
        MPI_InitFinalize(int &, char **&, unsigned int = numbers::invalid_unsigned_int);
      };
    }
  }
}
