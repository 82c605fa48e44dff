#pragma once
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/fe/mapping.h>
#include <string>
#include <vector>
namespace dealii
{
  namespace DataOutBase
  {
    enum class CompressionLevel { no_compression, best_speed, best_compression, default_compression, plain_text };
    struct VtkFlags {
      VtkFlags(const double time = 0., const unsigned int cycle = 0, const bool print_date_and_time = true,
               const CompressionLevel = CompressionLevel::best_speed, const bool write_higher_order_cells = false);
      CompressionLevel compression_level;
      bool write_higher_order_cells;
      double time;
      unsigned int cycle;
    };
  }
  namespace DataComponentInterpretation { enum DataComponentInterpretation { component_is_scalar, component_is_part_of_vector, component_is_part_of_tensor }; }
  template <int dim, int spacedim = dim>
  class DataOut
  {
  public:
    enum CurvedCellRegion { no_curved_cells, curved_boundary, curved_inner_cells };
    enum DataVectorType { type_dof_data, type_cell_data, type_automatic };
    template <class V> void add_data_vector(const V &, const std::string &, const DataVectorType);
    void attach_dof_handler(const DoFHandler<dim, spacedim> &);
    template <class V> void add_data_vector(const V &, const std::string &);
    template <class V> void add_data_vector(const V &, const std::vector<std::string> &);
    template <class V, class P> void add_data_vector(const DoFHandler<dim, spacedim> &, const V &, const P &);
    void build_patches(const unsigned int = 0);
    void build_patches(const Mapping<dim, spacedim> &, const unsigned int = 0, const CurvedCellRegion = curved_boundary);
    template <typename F> void set_cell_selection(const F &) {}
    void set_flags(const DataOutBase::VtkFlags &);
    void write_vtu_with_pvtu_record(const std::string &, const std::string &, const unsigned int, const MPI_Comm, const unsigned int = 4, const unsigned int = 0) const;
    void write_vtu_in_parallel(const std::string &, const MPI_Comm) const;
    void clear();
  };
}
