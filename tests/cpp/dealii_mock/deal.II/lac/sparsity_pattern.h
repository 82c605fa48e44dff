#pragma once
#include <deal.II/lac/dynamic_sparsity_pattern.h>
