#pragma once
#include <deal.II/base/function.h>
