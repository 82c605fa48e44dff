#pragma once
#include <deal.II/lac/sparse_matrix.h>
